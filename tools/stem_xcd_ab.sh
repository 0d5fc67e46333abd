#!/bin/bash
# The 16-bit stem with every frame's tiles on one XCD (default) against launch-order tiles (AGRL_STEM_XCD_MAP=0): parity tests, HIP-event
# times (tools/stem_bench.py), HBM fetch bytes per launch (rocprofv3 --pmc FETCH_SIZE, its own pass) and the bench step. -> gpurun_out/stem_xcd_ab.txt
export TMPDIR=/tmp
O=gpurun_out/stem_xcd_ab.txt
mkdir -p gpurun_out
{
python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k stem 2>&1 | tail -2
for v in 1 0 1 0; do echo "AGRL_STEM_XCD_MAP=$v"; AGRL_STEM_XCD_MAP=$v python3 tools/stem_bench.py 30 2>&1 | grep stem; done
for v in 1 0; do
  export AGRL_STEM_XCD_MAP=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/stem_fetch_$v -o f --output-format csv -- python3 tools/stem_bench.py 10 > /dev/null 2> gpurun_out/stem_fetch_$v.err
  python3 - gpurun_out/stem_fetch_$v $v <<'PY'
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*counter_collection.csv'))[0]
tot, n = 0.0, 0
for r in csv.DictReader(open(f)):
    if 'stem_mfma_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        tot += float(r['Counter_Value']); n += 1
print("AGRL_STEM_XCD_MAP=%s: FETCH_SIZE %.1f MB per launch over %d launches (KiB units x 2, the gfx950 correction of tools/pmc_traffic.py)" % (sys.argv[2], tot / n * 1024 * 2 / 1e6, n))
PY
done
unset AGRL_STEM_XCD_MAP
} > $O 2>&1
bash tools/ab_step.sh AGRL_STEM_XCD_MAP 1 0 3 >> $O 2>&1
cat $O
