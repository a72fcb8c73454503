#!/bin/bash
# round 5, call b: the new GPU tests (RCCL entry points, batch-128 oracle steps) under a time limit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_round5_gpu.py -q -m gpu -x --durations=10 > gpurun_out/r5b/tests.log 2>&1
tail -30 gpurun_out/r5b/tests.log
