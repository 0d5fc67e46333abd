#!/bin/bash
# Same-box A/B of everything round 5 dispatched: default tree against the round-4 dispatch (every round-5 switch off). -> gpurun_out/ab_round5.txt
OFF="AGRL_HIP_CONV1X1_DUO=0 AGRL_CONV3X3_HALF=0 AGRL_DISTMAT_TILE_N=256 AGRL_HIP_FUSE_DS_STRIDED=0 AGRL_STEM_SPLIT_LDS=0 AGRL_STEM_XCD_MAP=0 AGRL_CONV3X3_FAT_PB=2 AGRL_HIP_CONV1X1_DUO_C1=0 AGRL_HIP_CONV3X3_PACKED_L2=0"
O=gpurun_out/ab_round5.txt; mkdir -p gpurun_out; : > $O
run() {
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --no-config5 --no-config4 --no-modes --sustain-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', d['ms_per_step'], d['ms_per_step_blocks'], 'layer-4 pointwise ms', d['roofline_pointwise_layer4']['ms_per_step'], 'frac', d['roofline_pointwise_layer4']['frac'], 'dominant frac', d['roofline']['frac'])" >> $O
}
for i in 1 2 3; do
  LABEL="round-5 dispatch on " run AGRL_DUMMY=1
  LABEL="round-5 dispatch off" run $OFF
done
cat $O
