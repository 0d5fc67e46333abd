#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/mall_gallery_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_mall_gallery_probe.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke
AGRL_SPLIT16_NS=3 timeout 600 python tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_fp16x3_ns3.txt
timeout 600 python tools/profile_layers.py fp16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_profile_layers_fp16x3_ns2.txt
paste <(awk '{print $1, $2, $3, $4, $5, $6, $7, $(NF-2)}' gpurun_out/r06_profile_layers_fp16x3_ns2.txt) <(awk '{print $(NF-2)}' gpurun_out/r06_profile_layers_fp16x3_ns3.txt) | sed -n 1,30p
tail -1 gpurun_out/r06_profile_layers_fp16x3_ns2.txt gpurun_out/r06_profile_layers_fp16x3_ns3.txt
