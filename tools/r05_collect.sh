#!/bin/bash
# Round-5 profile collection on the GPU box (one gpurun call): the traffic-mix microbenchmark, the phase timeline and ablations of
# conv1x1_duo_kernel, PMC passes of its pooled form, the 8x-gallery distance matrix's FETCH_SIZE, then tools/collect_profiles.sh
# (kernel stats + FETCH / WRITE traffic of the bench command + the default bench line).
export TMPDIR=/tmp
mkdir -p gpurun_out
# (built here from its source: no binary in the history)
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -o /tmp/mix_stream tools/ubench/mix_stream.hip && timeout 120 /tmp/mix_stream > gpurun_out/r05_mix_stream.txt 2>&1
AGRL_HIP_LIB=$PWD/agrl.pytorch_amd/lib/libagrl_hip_duoabl64.so timeout 200 python3 tools/duo_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_duo_timeline.txt
DUO_ABLS="0 1 2 3 4 8 16 32" timeout 600 bash tools/duo_ablate_run.sh 2>&1 | grep -v "Memory access" > gpurun_out/r05_duo_ablations.txt
timeout 400 bash tools/pmc_any.sh r05_conv1x1_duo_pool tools/conv1x1_duo_bench.py 5 256 - pool1 > gpurun_out/r05_pmc_conv1x1_duo_pool.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
tail -3 gpurun_out/r05_collect.log
