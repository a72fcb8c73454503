#!/bin/bash
# round 5, call c: (1) does the penalty drift between schedules survive when NO contraction splits K with atomics on the
# forward / data-gradient path?  (2) which host call sites issue the three image-sized hipMemcpy per iteration?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" 2>&1 | grep -o '"max_relative_loss_difference": [0-9.e-]*\|"timed_schedule_twice": [0-9.e-]*\|"single_stream_twice": [0-9.e-]*\|disagree.*' | head -3 | cut -c1-300 | tr '\n' ' ')
  echo "$label: $out"
}
B="python bench.py --steps 2 --warmup 1 --image-size 64 --batch-per-gpu 2 --no-cpu-baseline --no-roofline"
{
for i in 1 2 3 4 5 6 7 8; do run "no split-K anywhere $i" SRGAN_PW_SPLIT_BELOW=0 SRGAN_CONV3_SPLIT_BELOW=0 SRGAN_TILE_TARGET=1 $B; done
for i in 1 2 3 4; do run "plain $i" $B; done
} | tee gpurun_out/r5c/no_split.txt
timeout 600 python scratch/find_copies.py > gpurun_out/r5c/copies.txt 2>&1
head -80 gpurun_out/r5c/copies.txt
