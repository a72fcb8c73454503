#!/bin/bash
# rocprofv3 kernel statistics of ONE step (3 steps - 1 step, halved), single stream:  scratch/stats_one.sh <tag> <name> [bench arguments...]
tag=$1; name=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for steps in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${name}_s$steps -o t -- python3 bench.py --steps $steps --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --single-stream "$@" > $out/prof_${name}_s$steps.log 2>&1
  cp $(find $out/prof_${name}_s$steps -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${steps}step_${name}.csv
  rm -rf $out/prof_${name}_s$steps
done
python scratch/per_step_stats.py $out/kernel_stats_1step_${name}.csv $out/kernel_stats_3step_${name}.csv > $out/kernel_stats_per_step_${name}.md
