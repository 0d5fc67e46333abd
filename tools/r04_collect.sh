#!/bin/bash
# Round-4 profile collection on the GPU box (one gpurun call): MFMA stream ceiling, PMC passes of the four-wave kernels,
# then tools/collect_profiles.sh (kernel stats + FETCH / WRITE traffic of the bench command + the default bench line).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/ubench/mfma_stream > gpurun_out/r04_mfma_stream.txt 2>&1
PMC_SCRIPT=tools/conv3x3_bench.py timeout 400 bash tools/pmc_conv_full.sh r04_conv3x3_fat 5 > /dev/null 2>&1
PMC_SCRIPT=tools/seam_bench.py timeout 400 bash tools/pmc_conv_full.sh r04_bottleneck_seam 5 > /dev/null 2>&1
PMC_SCRIPT=tools/conv1x1_bench.py timeout 400 bash tools/pmc_conv_full.sh r04_conv1x1_fat_2048_512 5 256 0 > /dev/null 2>&1
PMC_SCRIPT=tools/conv1x1_bench.py timeout 400 bash tools/pmc_conv_full.sh r04_conv1x1_fat_dual 5 256 2 > /dev/null 2>&1
timeout 1500 bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
tail -3 gpurun_out/r04_collect.log
