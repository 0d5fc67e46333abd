#!/bin/bash
# Round 5, final tree: kernel stats + traffic of the bench command + the default bench line (tools/collect_profiles.sh) and the
# per-launch profile (tools/profile_layers.py).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 bash tools/collect_profiles.sh ${1:-r05b} > gpurun_out/${1:-r05b}_collect.log 2>&1
python3 tools/profile_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${1:-r05b}_profile_layers.txt
tail -3 gpurun_out/${1:-r05b}_collect.log
