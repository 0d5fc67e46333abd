#!/bin/bash
# kernel-trace durations (rocprofv3) of one of the tools/*_bench.py scripts, per library:
#   tools/kernel_trace.sh tools/conv1x1_bench.py "conv1x1_fat|igemm_wide" [<lib> ...]     (no lib = the default library)
# HIP-event timing of a 30-100 us kernel carries ~10 us of launch overhead; these are the kernels' own durations.
export TMPDIR=/tmp
script=$1; pat=$2; shift 2
[ $# -eq 0 ] && set -- ""
for L in "$@"; do
  out=gpurun_out/prof_ktrace; rm -rf $out; mkdir -p $out
  if [ -n "$L" ]; then export AGRL_HIP_LIB=$L; else unset AGRL_HIP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $script 10 > $out/stdout.txt 2> $out/stderr.txt
  python3 tools/rocprof_stats.py $(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1) $out/stats.csv "ktrace" > /dev/null
  echo "== ${L:-default library}"; cat $out/stdout.txt | grep -v amdgpu.ids
  python3 - "$(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1)" "$pat" <<'PY'
# the bench scripts call each arm 1 + 3 + 10 times per shape, shape after shape: per kernel name, consecutive groups of 14 dispatches
import re, sqlite3, sys
seq = {}
for name, dur in sqlite3.connect(sys.argv[1]).execute("select name, end - start from kernels order by start"):
    m = re.search(sys.argv[2], name)
    if m and "pack_kernel" not in name:
        seq.setdefault(m.group(0), []).append(dur / 1e3)
for k, v in seq.items():
    print("  %-24s" % k + "  ".join("%6.1f (min %5.1f)" % (sorted(v[i:i + 14])[len(v[i:i + 14]) // 2], min(v[i:i + 14])) for i in range(0, len(v), 14)), "us per shape")
PY
done
