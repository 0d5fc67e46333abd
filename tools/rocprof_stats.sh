#!/bin/bash
# Kernel-trace summary of any script: tools/rocprof_stats.sh <tag> <script.py> [args]  -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 "$@" > $out/stdout.txt 2> $out/stderr.txt
python3 tools/rocprof_stats.py $(ls $out/trace/*results.db $out/trace/*/*results.db 2>/dev/null | head -1) gpurun_out/${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 $*"
head -${TOPN:-14} gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
tail -4 $out/stdout.txt
