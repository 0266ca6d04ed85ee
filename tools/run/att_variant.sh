# usage: att_variant.sh "<extra hipcc defines>"  -- bench with attention.hip rebuilt with those defines
set -o pipefail
cd gw_whisper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast $1 -c attention.hip -o /tmp/att_var.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libgww_var.so build/elementwise.o build/logmel.o build/gemm_bf16.o build/gemm_astat.o build/gemm_fulln.o build/gemm_f32.o /tmp/att_var.o build/attention_bwd.o build/train_ops.o build/dora_grads.o build/mlp_fused.o build/qscan.o build/encoder.o || exit 1
cd ../..
GWW_LIB=/tmp/libgww_var.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --no-pooled --steps 8 --warmup 3 > gpurun_out/bench_var.json 2> gpurun_out/bench_var.err; python tools/show_bench.py gpurun_out/bench_var.json
