#!/bin/bash
# A/B of the plane-PAIR layout of the conforming mode (fp16x3): AGRL_HIP_SPLIT16_PAIRS=1 / 0, alternating, same box -> gpurun_out/ab_pairs.txt
mkdir -p gpurun_out
: > gpurun_out/ab_pairs.txt
for i in 1 2; do
  for v in 1 0; do
    AGRL_HIP_SPLIT16_PAIRS=$v python3 bench.py --precision fp16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --no-config5 --no-config4 --no-modes --sustain-seconds 0 --no-host-issue > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "PAIRS=$v FAILED" >> gpurun_out/ab_pairs.txt; tail -5 gpurun_out/ab_tmp.err >> gpurun_out/ab_pairs.txt; continue; }
    python3 - "PAIRS=$v" >> gpurun_out/ab_pairs.txt <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('ms_per_step_blocks'))
PY
  done
done
cat gpurun_out/ab_pairs.txt
