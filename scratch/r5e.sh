#!/bin/bash
# round 5, call e: where does the data-parallel graph capture with side streams die?  + k4/s2 forward tests + every aten op of a step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout 600 python -X faulthandler bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph --image-size 64 --batch-per-gpu 2 > gpurun_out/r5e/dp_graph_64.txt 2>&1
echo "exit $?" >> gpurun_out/r5e/dp_graph_64.txt
tail -40 gpurun_out/r5e/dp_graph_64.txt | cut -c1-400
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_mixed_precision_gpu.py -q -m gpu -x -k "conv_passes or representable or rounding or transposed" > gpurun_out/r5e/tests_k4s2.log 2>&1
tail -8 gpurun_out/r5e/tests_k4s2.log
timeout 600 python scratch/find_copies.py --threshold 0 > gpurun_out/r5e/aten_ops.txt 2>&1
head -70 gpurun_out/r5e/aten_ops.txt
line() { grep '^{' | tail -1; }
for w in driving-fp16 age-vgg-bf16; do
  python bench.py --workload $w --steps 20 --warmup 5 2>/dev/null | line > gpurun_out/r5e/bench_$w.json
  SRGAN_NO_K4S2_FWD=1 python bench.py --workload $w --steps 20 --warmup 5 2>/dev/null | line > gpurun_out/r5e/bench_${w}_generic_fwd.json
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > gpurun_out/r5e/bench_512.json
for f in gpurun_out/r5e/bench_*.json; do
python - <<PY
import json
try:
    d = json.load(open("$f")); r = d.get("roofline", {}); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), r.get("frac"), r.get("launches"), r.get("low_precision_kernel_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("timed_schedule_twice")))
except Exception as e:
    print("$f FAILED", e)
PY
done
