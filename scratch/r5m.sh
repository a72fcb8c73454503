#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
timeout 1500 python -m pytest tests/test_round5_gpu.py -q -m gpu > gpurun_out/r5m/tests_r5.log 2>&1
tail -25 gpurun_out/r5m/tests_r5.log | cut -c1-300
