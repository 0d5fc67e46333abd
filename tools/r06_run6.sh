#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "persistent_form" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06_run6_tests.log
grep -E "passed|failed" gpurun_out/r06_run6_tests.log
if grep -q "failed" gpurun_out/r06_run6_tests.log; then tail -25 gpurun_out/r06_run6_tests.log; exit 0; fi
timeout 1500 bash tools/ab_step.sh AGRL_DUO_PERSIST 1 unset 4
cp gpurun_out/ab_AGRL_DUO_PERSIST.txt gpurun_out/r06_ab_duo_persist.txt
