#!/bin/bash
# round 5, call f: the whole GPU suite on the ordered-finish library (+ the data-parallel graph on one compute stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r5f/gpu_tests.log 2>&1
tail -25 gpurun_out/r5f/gpu_tests.log | cut -c1-250
line() { grep '^{' | tail -1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph --image-size 224 2>gpurun_out/r5f/err_dp_graph.txt | line > gpurun_out/r5f/bench_224_dp_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --image-size 224 2>/dev/null | line > gpurun_out/r5f/bench_224_dp_eager.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --step-graph --image-size 224 2>/dev/null | line > gpurun_out/r5f/bench_224_graph.json
for f in gpurun_out/r5f/bench_*.json; do
python - <<PY
import json
try:
    d = json.load(open("$f")); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), d["config"].get("host_ms_per_step"), d["config"].get("launch"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("limit")))
except Exception as e:
    print("$f FAILED", e)
PY
done
tail -5 gpurun_out/r5f/err_dp_graph.txt | cut -c1-300
