#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/profile_layers.py fp16x3 > gpurun_out/r06_profile_layers_fp16x3.txt 2>&1
AGRL_HIP_SPLIT16_PLANES=0 timeout 600 python tools/profile_layers.py fp16x3 > gpurun_out/r06_profile_layers_fp16x3_inloop.txt 2>&1
timeout 600 python tools/profile_layers.py fp32 > gpurun_out/r06_profile_layers_fp32.txt 2>&1
timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-config4 --no-config5 --no-accuracy --no-host-issue --no-modes --steps 5 --warmup 2 --sustain-seconds 0 > gpurun_out/r06_bench_fp16x3_planes.json 2> gpurun_out/r06_bench_fp16x3_planes.err
tail -70 gpurun_out/r06_profile_layers_fp16x3.txt
