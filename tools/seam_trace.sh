#!/bin/bash
# kernel-trace durations (rocprofv3) of tools/seam_bench.py for a list of libraries: tools/seam_trace.sh <lib> [<lib> ...]
export TMPDIR=/tmp
for L in "$@"; do
  out=gpurun_out/prof_seamtrace; rm -rf $out; mkdir -p $out
  AGRL_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 tools/seam_bench.py 10 > $out/stdout.txt 2> $out/stderr.txt
  python3 tools/rocprof_stats.py $(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1) $out/stats.csv "seam" > /dev/null
  echo "== $L"; grep -E "bottleneck_seam|igemm_wide" $out/stats.csv | awk -F, '{printf "%-50s calls %s avg %.1f us min %.1f\n", $1, $2, $4/1000, $6/1000}'
done
