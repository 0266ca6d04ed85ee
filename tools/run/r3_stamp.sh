#!/bin/bash
# round 3: phase stamps of k_mlp_fused<1,true> at B = 20 (one lock-step round) and B = 256 (steady state), modes 2 and 3
mkdir -p gpurun_out
out=gpurun_out/r3_stamp.txt
: > $out
for b in 20 256; do
  for mode in 2 3; do
    GWW_STAMP_MODE=$mode GWW_STAMP_QKV=1 GWW_STAMP_OP=1 python tools/stamp_mlp.py $b 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
cat $out
