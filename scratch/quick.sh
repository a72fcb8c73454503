#!/bin/bash
# one bench line + the per-kernel table of its roofline leg: scratch/quick.sh <tag> [bench arguments]
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report gpurun_out/$tag/shapes.txt "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python scratch/shapes.py gpurun_out/$tag/shapes.txt > gpurun_out/$tag/table.md 2>&1
head -${QUICK_LINES:-17} gpurun_out/$tag/table.md
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$tag/bench.json") if l.startswith("{")][-1])
r = d.get("roofline", {})
print("images/s %.2f  ms %.2f  frac %.4f  step_frac %.4f  launches %s" % (d["value"], d["ms_per_step"], r.get("frac", 0), r.get("step_frac_executed") or 0, r.get("launches")))
PY
