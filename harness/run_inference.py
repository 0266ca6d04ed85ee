#!/usr/bin/env python3
"""Counterpart of the reference's ``MLGWSC-1/inference.py`` command line (``main``, ``get_triggers``,
``build_model``: lines 407-679) on ``gw_whisper_amd``: strain segments in -> clustered triggers out, with the
windows sliced on the GPU, the Q-transform adapter, the DoRA-adapted Whisper encoder and the thresholding all on the
device, and the window range of every segment sharded over ranks when launched with ``torchrun``.

Same positional arguments and flags as the reference (``inputfile outputfile --white --softmax --lora-weights
--dense-weights --adapter-weights -t --step-size --cluster-threshold --device --force --verbose
--debug-triggers-file``), same output datasets (``time``, ``stat``, ``var``, ``all_vals``).  Differences, all forced by
the offline box:

  * without ``--white`` every segment is whitened on the device first (``gw_whisper_amd/whiten.py``, the counterpart of
    the reference's PyCBC whitening ``inference.py:56-137``; parity unpinned, PyCBC is absent -- DESIGN.md section 6);
  * files are HDF5 with the reference's layout (``/<detector>/<segment key>`` datasets with ``start_time`` and
    ``delta_t`` attributes) when ``h5py`` is importable, otherwise ``.npz`` with ``<detector>/<key>`` arrays plus
    ``<key>/start_time`` and ``<key>/delta_t`` scalars; the output follows the input's kind;
  * ``--synthetic SECONDS`` makes a two-detector white-noise segment instead of reading ``inputfile``;
  * the weights are optional (seeded random parameters without them -- there is no network for checkpoints);
    ``--encoder`` / ``--encoder-weights`` choose the base encoder the adapters were trained on (the reference loads
    ``openai/whisper-tiny`` from the hub).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time as t

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DETECTORS = ["H1", "L1"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="GW-Whisper sliding-window search (MI355X).")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--force", action="store_true", help="Overwrite existing output file.")
    p.add_argument("inputfile", type=str, help="Input HDF5 / .npz (ignored with --synthetic).")
    p.add_argument("outputfile", type=str, help="Output HDF5 / .npz (must not exist unless --force).")
    p.add_argument("--white", action="store_true", help="Input is already whitened (otherwise every segment is whitened on the device first).")
    p.add_argument("--softmax", action="store_true", help="Use Softmax outputs (default is USR logits).")
    p.add_argument("--coinc-window", type=float, default=0.1, help="(Reserved) coincidence window; not used.")
    p.add_argument("--lora-weights", type=str, default=None, help="peft adapter directory.")
    p.add_argument("--dense-weights", type=str, default=None, help="Dense head weights (.pth).")
    p.add_argument("--adapter-weights", type=str, default=None, help="Q-Adapter weights (.pt).")
    p.add_argument("-t", "--trigger-threshold", type=float, default=-0.5)
    p.add_argument("--step-size", type=float, default=0.1)
    p.add_argument("--cluster-threshold", type=float, default=0.35)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--debug-triggers-file", type=str, default=None)
    p.add_argument("--num-workers", type=int, default=0, help="Accepted for compatibility; slicing runs on the GPU.")
    p.add_argument("--synthetic", type=float, default=None, help="Seconds of synthetic whitened noise instead of inputfile.")
    p.add_argument("--encoder", type=str, default="tiny")
    p.add_argument("--encoder-weights", type=str, default=None, help="HF WhisperEncoder state_dict (.pth / .safetensors).")
    p.add_argument("--batch-size", type=int, default=256)
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------- files
def _have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


def read_segments(path: str):
    """{key: (strain [D, N] float32, start_time, delta_t)} in the reference's layout."""
    segs = {}
    if path.endswith(".npz"):
        z = np.load(path)
        keys = sorted({k.split("/", 1)[1] for k in z.files if k.split("/", 1)[0] == DETECTORS[0]})
        for key in keys:
            segs[key] = (np.stack([z[f"{d}/{key}"] for d in DETECTORS]).astype(np.float32),
                         float(z[f"{key}/start_time"]), float(z[f"{key}/delta_t"]))
        return segs
    import h5py
    with h5py.File(path, "r") as f:
        for key in f[DETECTORS[0]].keys():
            dss = [f[d][key] for d in DETECTORS]
            st = dss[0].attrs["start_time"]
            assert all(ds.attrs["start_time"] == st for ds in dss)
            segs[key] = (np.stack([ds[()] for ds in dss]).astype(np.float32), float(st), float(dss[0].attrs["delta_t"]))
    return segs


def write_result(path: str, arrays: dict):
    if path.endswith(".npz") or not _have_h5py():
        np.savez(path if path.endswith(".npz") else path + ".npz", **arrays)
        return
    import h5py
    with h5py.File(path, "w") as f:
        for k, v in arrays.items():
            f.create_dataset(k, data=v)


# ---------------------------------------------------------------------------------------------- model
def build_model(args, device):
    """``build_model`` / ``build_encoder_with_lora`` of the reference (inference.py:407-432)."""
    from gw_whisper_amd import inference as inf
    from gw_whisper_amd import synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.peft import LoraConfig, PeftModel, get_peft_model
    from gw_whisper_amd.qscan import QTransformAdapter
    torch.manual_seed(args.seed)
    if args.encoder_weights:
        if args.encoder_weights.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(args.encoder_weights)
        else:
            sd = torch.load(args.encoder_weights, map_location="cpu")
        encoder = WhisperEncoder(WhisperConfig.named(args.encoder))
        encoder.load_state_dict(sd)
    else:
        encoder = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict(args.encoder, seed=args.seed),
                                                       WhisperConfig.named(args.encoder), precision="bf16")
    if args.lora_weights:
        encoder = PeftModel.from_pretrained(encoder, args.lora_weights)
    else:
        targets = [n for n, _ in encoder.named_modules() if n.endswith(("q_proj", "k_proj", "v_proj"))]
        encoder = get_peft_model(encoder, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets))
    # the inference.py variant of the adapter (512 x 512 Q-scan, channels 16 / 32 / 64; inference.py:303-337) unless the
    # checkpoint was saved from the train.py variant (train.py:78-133): the variant is read off its tensor shapes
    if args.adapter_weights:
        adapter = QTransformAdapter.from_state_dict(torch.load(args.adapter_weights, map_location="cpu"),
                                                    n_detectors=len(DETECTORS))
    else:
        adapter = QTransformAdapter.inference_variant(n_detectors=len(DETECTORS))
    model = inf.GWWhisperClassifier(whisper_encoder=encoder, n_detectors=len(DETECTORS), adapter=adapter)
    if args.dense_weights:
        model.classifier.load_state_dict(torch.load(args.dense_weights, map_location="cpu"))
    if not args.softmax:
        inf.remove_softmax_from_classifier(model)
    return model.to(device).eval()


# ---------------------------------------------------------------------------------------------- search
def gather_shards(trig, vals, world):
    """Every rank's (triggers, per-batch scores) of ONE segment on every rank, in window order: the shards are contiguous,
    batch-aligned window ranges in rank order (``inference.shard_windows``), so concatenating them in rank order restores
    the single-GPU run's lists.  One object collective per segment, no tensor collective on the data path."""
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, (trig, vals))
    return [x for p in parts for x in p[0]], [v for p in parts for v in p[1]]


def get_triggers(args, device, rank, world):
    """``get_triggers`` of the reference (inference.py:492-590): every segment, longest first; each rank evaluates its
    batch-aligned shard of the segment's windows and rank 0 receives all triggers."""
    from gw_whisper_amd import inference as inf
    network = build_model(args, device)
    if args.synthetic is not None:
        rng = np.random.default_rng(args.seed)
        n = int(round(args.synthetic * 2048))
        segs = {"synthetic": (rng.standard_normal((len(DETECTORS), n)).astype(np.float32), 1.0e9, 1.0 / 2048)}
    else:
        segs = read_segments(args.inputfile)
    triggers, all_vals, clusters = {}, [], {}
    for key in sorted(segs, key=lambda k: segs[k][0].shape[1], reverse=True):
        strain, start, dt = segs[key]
        # --white: the file already holds whitened strain; otherwise every segment is whitened on the device first
        # (inference.py:218-246 -> gw_whisper_amd/whiten.py, parity unpinned: PyCBC is absent)
        slicer = inf.DeviceSegmentSlicer(strain, start_time=start, delta_t=dt, step_size=args.step_size, key=key,
                                         device=device, white=args.white)
        w0, w1 = inf.shard_windows(len(slicer), rank, world, args.batch_size)
        logging.info("rank %d: segment %s, windows [%d, %d) of %d", rank, key, w0, w1, len(slicer))
        if world == 1:
            # one GPU holds every score of the segment: threshold AND cluster on the device (gww_cluster_triggers_f64)
            trig, vals, clusters[key] = inf.evaluate_slices(slicer, network, trigger_threshold=args.trigger_threshold,
                                                            batch_size=args.batch_size, window_range=(w0, w1),
                                                            cluster_threshold=args.cluster_threshold)
        else:
            import torch
            import torch.distributed as dist
            trig, vals = inf.evaluate_slices(slicer, network, trigger_threshold=args.trigger_threshold,
                                             batch_size=args.batch_size, window_range=(w0, w1))
            trig, vals = gather_shards(trig, vals, world)
            if rank == 0:                                     # the gathered scores go back to rank 0's GPU to be clustered
                full = torch.from_numpy(np.concatenate(vals).astype(np.float32)).to(device) if vals else torch.empty(0, device=device)
                clusters[key] = inf.cluster_triggers_device(slicer.times(0, len(slicer)), full, args.trigger_threshold,
                                                            args.cluster_threshold)
        triggers[key] = trig
        all_vals.extend(vals)
    keys = sorted(triggers)
    clustered = tuple(np.concatenate([clusters[k][j] for k in keys]) if keys and all(k in clusters for k in keys) else None
                      for j in range(3))
    return dict(sorted(triggers.items(), key=lambda x: x[0])), all_vals, clustered


def main(argv=None) -> int:
    start = t.time()
    args = parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else (logging.INFO if args.verbose else logging.WARN),
                        format="%(levelname)s | %(asctime)s: %(message)s", datefmt="%d.%m.%Y %H:%M:%S")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.path.isfile(args.outputfile) and not args.force:
        raise RuntimeError("Output file exists. Use --force to overwrite.")
    if args.debug_triggers_file is not None and os.path.isfile(args.debug_triggers_file) and not args.force:
        raise RuntimeError("Triggers file exists. Use --force to overwrite.")
    if not torch.cuda.is_available():
        raise SystemExit("run_inference: no GPU -- gw_whisper_amd has no CPU fallback")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if args.device == "cuda" else torch.device(args.device)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from gw_whisper_amd import inference as inf
    triggers, all_vals, clustered = get_triggers(args, device, rank, world)
    if rank == 0:
        logging.info("Total slices above threshold %.3f: %d", args.trigger_threshold, sum(len(v) for v in triggers.values()))
        if args.debug_triggers_file is not None:
            write_result(args.debug_triggers_file, {k: np.array(v, dtype=np.float32) for k, v in triggers.items()})
        # clusters come from the device kernel (per segment, in key order: what get_clusters' loop over the dict does); the
        # host restatement of the reference stays as the cross-check
        time_arr, stat_arr, var_arr = clustered if clustered[0] is not None else inf.get_clusters(triggers, args.cluster_threshold)
        flat = np.concatenate(all_vals).astype("float32") if len(all_vals) else np.array([], dtype="float32")
        write_result(args.outputfile, {"time": time_arr, "stat": stat_arr, "var": var_arr, "all_vals": flat})
        print(f"Total execution time: {t.time() - start:.2f} seconds")
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
