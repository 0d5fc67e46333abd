#!/bin/bash
# round 6, second GPU call: the split-fp16 plane kernels, the model in the conforming mode (planes behind layer 3's first block), the
# full split, and the mode timings
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "split_fp16 or split16 or subnormal" 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "fp32_matches_oracle or benchmarked_size" 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_gpu_fullsplit.py -q -x -s -k "split_fp16" 2>&1 | tail -30
} > gpurun_out/r06_run2_tests.log 2>&1
timeout 1500 python bench.py --no-cpu-baseline --no-config4 --no-config5 --no-accuracy --no-host-issue > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err
AGRL_HIP_SPLIT16_PLANES=0 timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-config4 --no-config5 --no-accuracy --no-host-issue --no-modes --steps 5 --warmup 2 --sustain-seconds 0 > gpurun_out/r06_bench_fp16x3_inloop.json 2> gpurun_out/r06_bench_fp16x3_inloop.err
timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-config4 --no-config5 --no-accuracy --no-host-issue --no-modes --steps 5 --warmup 2 --sustain-seconds 0 > gpurun_out/r06_bench_fp16x3_planes.json 2> gpurun_out/r06_bench_fp16x3_planes.err
tail -c 2500 gpurun_out/r06_run2_tests.log
