#!/bin/bash
# Same-box A/B of the bench step under an environment switch: tools/ab_step.sh VAR ON OFF [pairs]  ->  gpurun_out/ab_VAR.txt
# (alternating runs of bench.py's timed block only; prints ms_per_step, the three extra blocks and the HBM-bound-trunk sum)
set -u
VAR=$1; ON=$2; OFF=$3; PAIRS=${4:-3}
OUT=gpurun_out/ab_${VAR}.txt
mkdir -p gpurun_out
: > "$OUT"
for i in $(seq 1 "$PAIRS"); do
  for v in "$ON" "$OFF"; do
    if [ "$v" = unset ]; then E="-u $VAR"; else E="$VAR=$v"; fi   # "unset" = run without the variable
    env $E python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy --no-config5 --no-config4 --no-modes --sustain-seconds 0 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$VAR=$v FAILED" >> "$OUT"; tail -5 gpurun_out/ab_tmp.err >> "$OUT"; continue; }
    python3 - "$VAR=$v" >> "$OUT" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
t = d.get('roofline_hbm_bound_trunk', {})
print(sys.argv[1], d['ms_per_step'], d.get('ms_per_step_blocks'), 'hbm-bound trunk ms', t.get('ms_per_step'), 'frac', t.get('frac'))
PY
  done
done
cat "$OUT"
