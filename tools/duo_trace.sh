#!/bin/bash
# kernel-trace durations (rocprofv3) of conv1x1_duo_kernel per start delay against igemm_wide_kernel on the layer-4 conv3 + residual shape
#   tools/duo_trace.sh "0 6000 12000" [form]
export TMPDIR=/tmp
form=${2:-res}
for st in $1; do
  out=gpurun_out/prof_duo_$st; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 tools/conv1x1_duo_bench.py 10 256 $st $form > $out/stdout.txt 2> $out/stderr.txt
  python3 - "$(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1)" $st <<'PY'
import re, sqlite3, sys
seq = {}
for name, dur in sqlite3.connect(sys.argv[1]).execute("select name, end - start from kernels order by start"):
    m = re.search("conv1x1_duo_kernel|igemm_wide_kernel", name)
    if m:
        seq.setdefault(m.group(0), []).append(dur / 1e3)
for k, v in seq.items():
    if k == "igemm_wide_kernel":   # (the conv1 of before() is conv1x1_fat_kernel; igemm_wide_kernel here = the residual form only)
        pass
    v = sorted(v[-10:])
    print("stagger %-6s %-22s median %6.1f  min %6.1f  max %6.1f us (last 10 of %d launches)" % (sys.argv[2], k, v[len(v) // 2], v[0], v[-1], len(seq[k])))
PY
done
