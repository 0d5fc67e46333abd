#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
AGRL_SPLIT16_NS=364 timeout 600 python tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_fp16x3_ns364.txt
timeout 600 python tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_fp16x3_default.txt
paste <(awk '{print $1, $2, $3, $4, $5, $6, $7, $(NF-2)}' gpurun_out/r06_profile_layers_fp16x3_default.txt) <(awk '{print $(NF-2)}' gpurun_out/r06_profile_layers_fp16x3_ns364.txt) | sed -n 1,30p
tail -n 1 gpurun_out/r06_profile_layers_fp16x3_default.txt gpurun_out/r06_profile_layers_fp16x3_ns364.txt
