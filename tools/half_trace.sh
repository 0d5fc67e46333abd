export TMPDIR=/tmp
for v in 1 0; do
  out=gpurun_out/prof_half_$v; rm -rf $out; mkdir -p $out
  AGRL_CONV3X3_HALF=$v AGRL_CONV3X3_HALF_STAGGER=0 timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 tools/conv3x3_bench.py 20 > $out/stdout.txt 2> $out/stderr.txt
  python3 - "$(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1)" $v <<'PY'
import re, sqlite3, sys
seq = {}
for name, dur in sqlite3.connect(sys.argv[1]).execute("select name, end - start from kernels order by start"):
    m = re.search("conv3x3_half_kernel|conv3x3_fat_kernel<[12]>|conv3x3_wide_kernel<[^>]*>", name)
    if m:
        seq.setdefault(m.group(0), []).append(dur / 1e3)
for k, v in seq.items():
    half = len(v) // 2
    a, b = sorted(v[:half]), sorted(v[half:])
    print("HALF=%s %-28s first shape median %6.1f (n=%d)  second shape median %6.1f (n=%d)" % (sys.argv[2], k, a[len(a)//2] if a else 0, len(a), b[len(b)//2] if b else 0, len(b)))
PY
done
