#!/bin/bash
# round 5, call g: everything ordered (K splits, grouped weight gradients, parameter sums) -- tests, reproducibility, A / B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_round5_gpu.py -q -m gpu -k "split or reproducible" > gpurun_out/r5g/tests_r5.log 2>&1
tail -12 gpurun_out/r5g/tests_r5.log | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_round5_gpu.py > gpurun_out/r5g/gpu_tests.log 2>&1
tail -6 gpurun_out/r5g/gpu_tests.log | cut -c1-250
line() { grep '^{' | tail -1; }
for size in 512 224; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --image-size $size 2>gpurun_out/r5g/err_$size.txt | line > gpurun_out/r5g/bench_${size}_ordered.json
  SRGAN_ATOMIC_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --image-size $size 2>/dev/null | line > gpurun_out/r5g/bench_${size}_atomic.json
done
SRGAN_PKS_GROUPS=320 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --image-size 224 2>/dev/null | line > gpurun_out/r5g/bench_224_pks320.json
SRGAN_NO_PW_KSPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --image-size 224 2>/dev/null | line > gpurun_out/r5g/bench_224_no_ksplit.json
python bench.py --steps 2 --warmup 1 --image-size 64 --batch-per-gpu 2 --no-cpu-baseline --no-roofline 2>/dev/null | line > gpurun_out/r5g/bench_64.json
for f in gpurun_out/r5g/bench_*.json; do
python - <<PY
import json
try:
    d = json.load(open("$f")); r = d.get("roofline", {}); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), r.get("frac"), r.get("launches"), r.get("kernel_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("max_weight_difference"), c.get("timed_schedule_twice")))
except Exception as e:
    print("$f FAILED", e)
PY
done
tail -3 gpurun_out/r5g/err_512.txt | cut -c1-300
