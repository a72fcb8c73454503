#!/bin/bash
# fp32 parity suites + the headline twice + 224 (after a change to the fp32 kernels):  scratch/check_fp32.sh <tag>
tag=$1; mkdir -p gpurun_out/$tag
python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_determinism_gpu.py tests/test_grouped_weight_gradients_gpu.py -x -q > gpurun_out/$tag/tests.log 2>&1; tail -3 gpurun_out/$tag/tests.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/$tag/bench$i.json; done
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --image-size 224 2>/dev/null | grep "^{" | tail -1 > gpurun_out/$tag/bench224.json
python - <<PY
import json
for n in ["bench1", "bench2", "bench224"]:
    d = json.load(open("gpurun_out/$tag/%s.json" % n)); r = d["roofline"]
    print(n, round(d["value"], 2), round(d["ms_per_step"], 2), round(r["frac"], 4), round(r["kernel_ms_per_step"], 1))
PY
