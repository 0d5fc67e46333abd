#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3300 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1
tail -15 gpurun_out/r06_gpu_suite.log
