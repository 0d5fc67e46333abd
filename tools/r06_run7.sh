#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
AGRL_DUO_PERSIST=1 timeout 600 python tools/profile_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_persist.txt
timeout 600 python tools/profile_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_oneshot.txt
paste <(awk '{print $1, $2, $NF-0, $(NF-2)}' gpurun_out/r06_profile_layers_oneshot.txt) <(awk '{print $(NF-2)}' gpurun_out/r06_profile_layers_persist.txt) | sed -n 1,60p
