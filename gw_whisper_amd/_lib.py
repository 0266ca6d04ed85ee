"""ctypes binding of libgww.so (the C ABI declared in include/gww.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C gw_whisper_amd/csrc``.  There is NO fallback: if the shared object is
missing or a call fails, the product path raises.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GWW_LIB") or os.path.join(_HERE, "libgww.so")   # GWW_LIB: tuning builds only

PREC_BF16 = 0
PREC_F32 = 1

EPI_BIAS, EPI_GELU, EPI_RESID = 0, 1, 2


class EncCfg(C.Structure):
    _fields_ = [("d_model", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
                ("ffn", C.c_int), ("n_mels", C.c_int), ("t_in", C.c_int)]


class EncGlobals(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "pos", "ln_w", "ln_b")]


class EncLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("ln1_w", "ln1_b", "q_w", "q_b", "k_w", "v_w", "v_b", "o_w", "o_b",
                 "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class DoraTarget(C.Structure):
    _fields_ = [("layer", C.c_int), ("proj", C.c_int), ("r", C.c_int), ("scaling", C.c_float),
                ("A", C.c_void_p), ("B", C.c_void_p), ("mag", C.c_void_p), ("nrm", C.c_void_p),
                ("dA", C.c_void_p), ("dB", C.c_void_p), ("dm", C.c_void_p)]


class DoraMergeItem(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("m", C.c_void_p), ("w_eff", C.c_void_p),
                ("norm_out", C.c_void_p), ("scaling", C.c_float), ("d_out", C.c_int), ("d_in", C.c_int), ("r", C.c_int)]


# name -> (restype, argtypes); every symbol include/gww.h declares
SIGNATURES = {
    "gww_version": (C.c_int, []),
    "gww_last_error": (C.c_char_p, []),
    "gww_frontend_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gww_frontend_destroy": (None, [C.c_void_p]),
    "gww_logmel_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "gww_logmel_host_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p]),
    "gww_encoder_create": (C.c_int, [C.POINTER(EncCfg), C.POINTER(C.c_void_p)]),
    "gww_encoder_destroy": (None, [C.c_void_p]),
    "gww_encoder_set_weights": (C.c_int, [C.c_void_p, C.POINTER(EncGlobals), C.POINTER(EncLayer),
                                          C.c_int, C.c_void_p]),
    "gww_encoder_update_weights": (C.c_int, [C.c_void_p, C.POINTER(EncGlobals), C.POINTER(EncLayer), C.c_int,
                                             C.POINTER(C.c_uint), C.c_void_p]),
    "gww_encoder_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "gww_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "gww_encoder_set_split": (C.c_int, [C.c_void_p, C.c_int]),
    "gww_encoder_trace_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gww_encoder_trace_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "gww_encoder_trace_classes": (C.c_int, []),
    "gww_encoder_trace_class_name": (C.c_char_p, [C.c_int]),
    "gww_train_saved_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "gww_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "gww_encoder_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "gww_encoder_train_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.POINTER(DoraTarget), C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p]),
    "gww_attention_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attention_bwd_log2q_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attention_log2q_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attention_lse_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_long, C.c_int, C.c_void_p]),
    "gww_gelu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "gww_dora_grads": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_float,
                                 C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "gww_dora_grads_multi": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                       C.POINTER(C.c_long), C.POINTER(C.c_void_p), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_long, C.c_int, C.c_void_p]),
    "gww_dora_merge_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gww_conv1_gelu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "gww_dora_merge_batch_f32": (C.c_int, [C.POINTER(DoraMergeItem), C.c_int, C.c_void_p]),
    "gww_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long,
                                C.c_int, C.c_void_p]),
    "gww_gemm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_gemm_bf16_v4_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_gemm_astat_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_ln_fold_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gww_gemm_fulln_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p]),
    "gww_mlp_fused_bf16": (C.c_int, [C.c_void_p] * 8 + [C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "gww_mlp_pack_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attn_out_mlp_fused_bf16": (C.c_int, [C.c_void_p] * 9 + [C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_void_p]),
    "gww_attn_out_mlp_final_bf16": (C.c_int, [C.c_void_p] * 11 + [C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "gww_mlp_pack_op_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_lnqkv_fused_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                       C.c_void_p]),
    "gww_qscan_energy_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                       C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]),
    "gww_qscan_interp_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gww_qadapter_tail_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "gww_qadapter_cnn_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gww_qadapter_cnn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gww_qadapter_cnn_backward_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gww_qadapter_cnn_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 9),
    "gww_qadapter_cnn_pack_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gww_qadapter_cnn_forward_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gww_welch_power_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "gww_column_median_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "gww_fir_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p]),
    "gww_cluster_triggers_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_double, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p]),
    "gww_gemm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                               C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attention_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gww_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
}

ABI_VERSION = 107   # include/gww.h GWW_VERSION this binding was written against

_lib = None


class GwwError(RuntimeError):
    pass


def lib():
    """Load libgww.so once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GwwError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gw_whisper_amd/csrc`.  gw_whisper_amd has no CPU / PyTorch fallback.")
    # libgww.so needs libamdhip64.so.7.  PyTorch-ROCm ships its own copy of that runtime;
    # two HIP runtimes in one process do not share devices, streams or allocations, so make
    # sure torch's copy is the one already loaded when libgww.so is resolved.
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # a stale build with the same symbol names but older argument lists would be called with the wrong ABI
    if handle.gww_version() < ABI_VERSION:
        raise GwwError(f"{LIB_PATH} reports ABI {handle.gww_version()}, this package needs {ABI_VERSION}: rebuild it "
                       "(`make -C gw_whisper_amd/csrc`)")
    _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gww_last_error()
        raise GwwError(f"{what or 'libgww call'} failed ({rc}): {msg.decode() if msg else '?'}")
