#!/bin/bash
# round 5, call a: the ordered row reduction after its ticket fix -- stress test, then round 4's schedule-check bisect again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_ops_gpu.py -q -m gpu -k "round5 or reduction or uninjected or ordered" > gpurun_out/r5a/tests.log 2>&1
tail -5 gpurun_out/r5a/tests.log
run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" 2>&1 | grep -o '"max_relative_loss_difference": [0-9.e-]*\|"timed_schedule_twice": [0-9.e-]*\|disagree.*' | head -3 | cut -c1-400 | tr '\n' ' ')
  echo "$label: $out"
}
B="python bench.py --steps 2 --warmup 1 --image-size 64 --batch-per-gpu 2 --no-cpu-baseline --no-roofline"
{
for i in 1 2 3 4; do run "plain $i" $B; done
for i in 1 2 3; do run "forced-dp $i" $B --force-dp --backend nccl; done
for i in 1 2; do run "forced-dp no-dnn-stream $i" SRGAN_NO_DNN_STREAM=1 $B --force-dp --backend nccl; done
for i in 1 2; do run "forced-dp no-penalty-stream $i" SRGAN_NO_PENALTY_STREAM=1 $B --force-dp --backend nccl; done
for i in 1 2 3; do run "plain no-aux $i" SRGAN_NO_AUX_STREAM=1 $B; done
for i in 1 2; do run "forced-dp blocking exchange $i" $B --force-dp --backend nccl --no-overlap-exchange; done
} | tee gpurun_out/r5a/bisect.txt
