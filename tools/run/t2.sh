echo "vg18 (default): $(timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep mlp_fused)"
for v in 6 10 14 24; do echo "vg$v: $(GWW_LIB=gw_whisper_amd/csrc/build/libgww_v$v.so timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep mlp_fused)"; done
