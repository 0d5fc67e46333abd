#!/bin/bash
# round 6, first GPU call: the new tests (split-fp16 conv + model mode + full split, 192-column distance tile, two-rank native train
# step / rank-0-only evaluate) and a bench line with the host-issue fields
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "split_fp16 or tile_192" 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "fp32_matches_oracle or benchmarked_size" 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_gpu_dist.py -q -x -s 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_gpu_fullsplit.py -q -x -s -k "split_fp16" 2>&1 | tail -30
} > gpurun_out/r06_run1_tests.log 2>&1
timeout 1500 python bench.py --no-cpu-baseline --no-config4 --no-config5 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
tail -c 1500 gpurun_out/r06_run1_tests.log
