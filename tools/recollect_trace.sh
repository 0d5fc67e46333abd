export TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
CMD="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-config5 --no-config4 --no-modes --sustain-seconds 0"
rm -rf $O/trace
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $CMD > $O/bench_under_rocprof.json 2> $O/trace.err
python3 tools/rocprof_stats.py $(ls $O/trace/*results.db $O/trace/*/*results.db 2>/dev/null | head -1) $O/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 $CMD (1x MI355X, fp16, B=32 S=8; 13 steps + 3 profiled steps + the 256-tracklet GraphLayer and 8x-gallery distance-matrix measurements); torch:* / copyBuffer rows are one-off weight packing, input generation and the yardstick's buffers, not part of a step (tools/step_ops.py)"
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
