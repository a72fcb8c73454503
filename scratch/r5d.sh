#!/bin/bash
# round 5, call d: the ordered split-K finish -- its tests, then bench lines with it and with round 4's atomics (A / B),
# then which aten ops are left on the hot path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
line() { grep '^{' | tail -1; }
timeout 1200 python -m pytest tests/test_round5_gpu.py -q -m gpu -x -k "split or repeatable or graph_replay" > gpurun_out/r5d/tests.log 2>&1
tail -15 gpurun_out/r5d/tests.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_step_graph_gpu.py -q -m gpu -x > gpurun_out/r5d/tests_ops.log 2>&1
tail -5 gpurun_out/r5d/tests_ops.log
for size in 512 224; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --image-size $size 2>gpurun_out/r5d/err_$size.txt | line > gpurun_out/r5d/bench_${size}_ordered.json
  SRGAN_ATOMIC_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --image-size $size 2>/dev/null | line > gpurun_out/r5d/bench_${size}_atomic.json
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph 2>gpurun_out/r5d/err_dp_graph.txt | line > gpurun_out/r5d/bench_512_dp_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph --image-size 224 2>/dev/null | line > gpurun_out/r5d/bench_224_dp_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --image-size 224 2>/dev/null | line > gpurun_out/r5d/bench_224_dp_eager.json
for f in gpurun_out/r5d/bench_*.json; do
python - <<PY
import json
try:
    d = json.load(open("$f")); r = d.get("roofline", {}); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), r.get("frac"), r.get("launches"), d["config"].get("host_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("timed_schedule_twice")))
except Exception as e:
    print("$f FAILED", e)
PY
done
timeout 600 python scratch/find_copies.py > gpurun_out/r5d/copies.txt 2>&1
head -60 gpurun_out/r5d/copies.txt
timeout 600 python scratch/count_ops.py > gpurun_out/r5d/count_ops.txt 2>&1
head -45 gpurun_out/r5d/count_ops.txt
