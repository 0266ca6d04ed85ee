#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of `bench.py` (two separate rocprofv3 --pmc runs) -> profiles/rNN_pmc_traffic.{json,md}.

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <profiles/rNN_pmc_traffic>

Per kernel (mean over the whole-batch dispatches of the timed region): FETCH_SIZE raw (KB units x 1024), doubled as
MI355X_MICROARCH.md prescribes for 16-B-per-lane streaming reads on gfx950 (an upper bound where a kernel also issues
narrower reads), WRITE_SIZE.  The JSON carries the hash of the kernel sources it was measured on (`bench.csrc_hash`):
bench.py reports `roofline.traffic` from it only while that hash matches the tree."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel-name substring (in order: first match wins) -> bench.py kernel class
CLASSES = [
    ("k_mlp_fused<1", "mlp_fused+next_ln_qkv"), ("k_mlp_fused<0", "mlp_fused(ln+fc1+gelu+fc2)"),
    ("k_mlp_fused<3", "mlp_fused+final_layernorm"),
    ("k_mlp_fused<2", "ln+qkv_proj"),
    ("k_attention_", "attention"), ("k_gemm_bf16_v4<3", "conv2_gelu_pos"), ("k_gemm_fulln<3", "conv2_gelu_pos"), ("k_conv1_mel", "conv1_gelu"), ("k_gemm_astat<4", "conv1_gelu"),
    ("k_gemm_astat<0, 1", "ln+qkv_proj"), ("k_gemm_astat<0, 0", "out_proj"), ("k_mel_to_tokens", "mel_to_tokens"),
    ("k_layernorm<", "final_layernorm"),
]


def load(d, counter):
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    assert files, "no counter_collection.csv under " + d
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            acc[name].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return acc


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out, rows = {}, []
    for sub, cls in CLASSES:
        for name in fetch:
            if sub not in name or cls in out:
                continue
            # whole-batch launches only: the largest grid of this kernel (the pooled B-row launches are tiny)
            gmax = max(g for _, g in fetch[name])
            f = [v for v, g in fetch[name] if g == gmax]
            w = [v for v, g in write.get(name, []) if g == gmax]
            if not f or not w:
                continue
            fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
            out[cls] = {"kernel": name.split("(")[0], "dispatches": len(f), "fetch_size_raw_bytes": fb,
                        "fetch_bytes_corrected_x2": 2 * fb, "write_bytes": wb}
            rows.append((cls, name.split("(")[0], fb, wb))
    doc = {"config": "whisper-tiny, B=256, bf16, one stream (whole-batch launches)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py), KB units x1024",
           "csrc_hash": bench.csrc_hash(), "kernels": out}
    json.dump(doc, open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as f:
        f.write(f"# HBM-side traffic per launch (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; separate passes of `bench.py`)\n\n"
                f"kernel sources hash `{doc['csrc_hash']}`.  FETCH_SIZE is reported raw and doubled (MI355X_MICROARCH.md: on gfx950 it "
                f"counts 16-B/lane streaming reads at half their bytes).\n\n"
                "| kernel class | kernel | FETCH_SIZE raw MB | x2 MB | WRITE_SIZE MB |\n|---|---|---|---|---|\n")
        for cls, name, fb, wb in rows:
            f.write(f"| {cls} | `{name}` | {fb / 1e6:.0f} | {2 * fb / 1e6:.0f} | {wb / 1e6:.0f} |\n")
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main()
