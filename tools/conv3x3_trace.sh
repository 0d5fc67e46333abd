#!/bin/bash
# kernel-trace durations (rocprofv3) of tools/conv3x3_bench.py for a list of libraries: tools/conv3x3_trace.sh <lib> [<lib> ...]
export TMPDIR=/tmp
for L in "$@"; do
  out=gpurun_out/prof_c3trace; rm -rf $out; mkdir -p $out
  AGRL_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 tools/conv3x3_bench.py 10 > $out/stdout.txt 2> $out/stderr.txt
  python3 tools/rocprof_stats.py $(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1) $out/stats.csv "c3" > /dev/null
  echo "== $L"; grep -E "conv3x3_fat_kernel|conv3x3_wide" $out/stats.csv | awk -F, '{printf "%-60s calls %s avg %.1f us min %.1f\n", substr($1,1,60), $2, $4/1000, $6/1000}'
done
