#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "dual_split16 or split_fp16" 2>&1 | grep -v "^$" | tail -20
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "fp32_matches_oracle or benchmarked_size or gsta" 2>&1 | grep -v "^$" | tail -30
timeout 1200 python -m pytest tests/test_gpu_fullsplit.py -q -x -s -k "split_fp16" 2>&1 | grep -v "^$" | tail -12
} > gpurun_out/r06_run5_tests.log 2>&1
timeout 600 python tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_fp16x3_c.txt
timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-config4 --no-config5 --no-accuracy --no-host-issue --no-modes --steps 5 --warmup 2 --sustain-seconds 0 > gpurun_out/r06_bench_fp16x3_planes.json 2> gpurun_out/r06_bench_fp16x3_planes.err
grep -E "passed|failed|FAILED|Error|dual split16|fp16x3 (cos|euc)" gpurun_out/r06_run5_tests.log; tail -3 gpurun_out/r06_profile_layers_fp16x3_c.txt
