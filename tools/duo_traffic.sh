#!/bin/bash
# HBM fetch / write bytes per launch of conv1x1_duo_kernel, shape by shape (rocprofv3 --pmc, one counter per pass) -> gpurun_out/duo_traffic.txt
export TMPDIR=/tmp
O=gpurun_out/duo_traffic.txt; : > $O
for c in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/duo_tr_${c}_$ctr; rm -rf $d
    rocprofv3 --kernel-trace --pmc $ctr -d $d -o t --output-format csv -- python3 tools/duo_traffic.py $c > gpurun_out/duo_tr_tmp.txt 2> $d.err
    python3 - $d $ctr $c >> $O <<'PY'
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*/*counter_collection.csv'))[0]
agg = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if r['Counter_Name'] == sys.argv[2] and ('conv1x1_duo' in k or 'igemm_wide' in k):
        a = agg.setdefault(k.split('(')[0][-60:], [0.0, 0]); a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in agg.items():
    print("%-9s %-10s %-50s %8.1f MB per launch (n=%d)" % (sys.argv[3], sys.argv[2], k, v / n * 1024 * (2 if sys.argv[2] == 'FETCH_SIZE' else 1) / 1e6, n))
PY
  done
  cat gpurun_out/duo_tr_tmp.txt >> $O
done
cat $O
