#!/bin/bash
# Main-loop ticks of diagnostic builds of the fused MLP kernel (light stamps: the main loop itself carries none).
# usage: tools/run/mlp_bisect.sh OUT  "B DEFS" ["B DEFS" ...]     e.g.  "256 -DGWW_MF_EXP=2 -DGWW_MF_SCHED=1"
out=$1; shift
export GWW_STAMP_MODE=2
for e in "$@"; do
  b=${e%% *}; d=${e#* }
  GWW_EXTRA_DEFS="$d" python tools/stamp_mlp.py $b 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
