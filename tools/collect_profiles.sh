#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): kernel-trace summary + the two PMC traffic passes of
# the SAME bench command, then the bench line itself. Outputs under gpurun_out/; copy the summaries into profiles/.
export TMPDIR=/tmp
R=${1:-r01}
mkdir -p gpurun_out/prof_$R
CMD="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R/trace -o trace -- python3 $CMD > gpurun_out/prof_$R/bench_trace.json 2> gpurun_out/prof_$R/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_$R/fetch -o fetch --output-format csv -- python3 $CMD > /dev/null 2> gpurun_out/prof_$R/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_$R/write -o write --output-format csv -- python3 $CMD > /dev/null 2> gpurun_out/prof_$R/write.err
python3 tools/rocprof_stats.py $(ls gpurun_out/prof_$R/trace/*results.db gpurun_out/prof_$R/trace/*/*results.db 2>/dev/null | head -1) gpurun_out/prof_$R/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 $CMD (1x MI355X, bf16, B=32 S=8; 13 steps + 3 profiled steps); torch:* rows are one-off weight packing / input generation"
python3 tools/pmc_traffic.py $(ls gpurun_out/prof_$R/fetch/*counter_collection.csv gpurun_out/prof_$R/fetch/*/*counter_collection.csv 2>/dev/null | head -1) $(ls gpurun_out/prof_$R/write/*counter_collection.csv gpurun_out/prof_$R/write/*/*counter_collection.csv 2>/dev/null | head -1) 16 bf16 > gpurun_out/prof_$R/traffic.txt 2>&1
cp profiles/traffic_r01.json gpurun_out/prof_$R/traffic.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_$R/bench_n1.json 2> gpurun_out/prof_$R/bench_n1.err
ls -la gpurun_out/prof_$R
