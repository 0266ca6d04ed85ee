set -o pipefail
cd gw_whisper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DGWW_ATT_NOLOAD -c attention.hip -o /tmp/att_noload.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libgww_noload.so build/elementwise.o build/logmel.o build/gemm_bf16.o build/gemm_astat.o build/gemm_fulln.o build/gemm_f32.o /tmp/att_noload.o build/attention_bwd.o build/train_ops.o build/dora_grads.o build/mlp_fused.o build/qscan.o build/encoder.o || exit 1
cd ../..
GWW_LIB=/tmp/libgww_noload.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --no-pooled --steps 5 --warmup 2 > gpurun_out/bench4.json 2> gpurun_out/bench4.err; python tools/show_bench.py gpurun_out/bench4.json
