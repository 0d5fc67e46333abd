#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): kernel-trace summary + the two PMC traffic passes of
# the SAME bench command, then the bench line itself. Outputs under gpurun_out/prof_<round>/; copy the summaries into profiles/.
export TMPDIR=/tmp
R=${1:-r06}
O=gpurun_out/prof_$R
mkdir -p $O
CMD="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-config5 --no-config4 --no-modes --no-host-issue --sustain-seconds 0"
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $CMD > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $CMD > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $CMD > /dev/null 2> $O/write.err
DB=$(ls $O/trace/*results.db $O/trace/*/*results.db 2>/dev/null | head -1)
NSTEP=$(python3 -c "import sqlite3,sys; print(sqlite3.connect(sys.argv[1]).execute(\"select count(*) from kernels where name like '%stem_mfma_kernel%'\").fetchone()[0])" $DB)
python3 tools/rocprof_stats.py $DB $O/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 $CMD (1x MI355X, fp16, B=32 S=8; $NSTEP forward steps in all = launches of stem_mfma_kernel: warm-up, the timed block, three more timed blocks, the profiled steps; + the 256-tracklet GraphLayer and 8x-gallery distance-matrix measurements); torch:* / copyBuffer rows are one-off weight packing, input generation and the yardstick's buffers, not part of a step (tools/step_ops.py)"
python3 tools/pmc_traffic.py $(ls $O/fetch/*counter_collection.csv $O/fetch/*/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/write/*counter_collection.csv $O/write/*/*counter_collection.csv 2>/dev/null | head -1) auto fp16 $O/traffic.json > $O/traffic.txt 2>&1
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
ls -la $O
