#!/bin/bash
# round 5, call h: where did the fully ordered library lose time?  per-kernel statistics ordered vs atomic, then bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5h; mkdir -p $out
timeout 600 python -m pytest tests/test_round5_gpu.py -q -m gpu -k "split or reproducible" > $out/tests_r5.log 2>&1
tail -4 $out/tests_r5.log | cut -c1-300
line() { grep '^{' | tail -1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line > $out/bench_512_ordered.json
SRGAN_ATOMIC_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line > $out/bench_512_atomic.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --image-size 224 2>/dev/null | line > $out/bench_224_ordered.json
for f in $out/bench_*.json; do
python - <<PY
import json
try:
    d = json.load(open("$f")); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), d["config"].get("host_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("max_weight_difference")))
except Exception as e:
    print("$f FAILED", e)
PY
done
for mode in ordered atomic; do
  for steps in 1 3; do
    if [ $mode = atomic ]; then export SRGAN_ATOMIC_SPLIT=1; else unset SRGAN_ATOMIC_SPLIT; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${mode}_s$steps -o t -- python3 bench.py --steps $steps --warmup 0 --no-cpu-baseline --no-roofline --single-stream > $out/prof_${mode}_s$steps.log 2>&1
    cp $(find $out/prof_${mode}_s$steps -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${steps}step_${mode}.csv
    rm -rf $out/prof_${mode}_s$steps
  done
  python scratch/per_step_stats.py $out/kernel_stats_1step_${mode}.csv $out/kernel_stats_3step_${mode}.csv > $out/kernel_stats_per_step_${mode}.md
done
unset SRGAN_ATOMIC_SPLIT
head -45 $out/kernel_stats_per_step_ordered.md | cut -c1-200
