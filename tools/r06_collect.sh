#!/bin/bash
# Round 6, final tree: kernel stats + FETCH / WRITE traffic of the bench command + the default bench line (tools/collect_profiles.sh),
# the per-launch profiles of the 16-bit step and of the conforming mode (tools/profile_layers.py).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 bash tools/collect_profiles.sh ${1:-r06} > gpurun_out/${1:-r06}_collect.log 2>&1
python3 tools/profile_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${1:-r06}_profile_layers.txt
python3 tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/${1:-r06}_profile_layers_fp16x3.txt
tail -3 gpurun_out/${1:-r06}_collect.log
