#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r5i/gpu_tests.log 2>&1
tail -8 gpurun_out/r5i/gpu_tests.log | cut -c1-250
bash scratch/knobs_r5.sh
