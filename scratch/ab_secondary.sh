#!/bin/bash
# A/B of the two 16-bit configurations and the headline:  scratch/ab_secondary.sh <tag> "<ENV=.. for B>" [steps]
tag=$1; assignment=${2:-X=0}; steps=${3:-100}
out=gpurun_out/$tag; mkdir -p $out
for w in age-vgg-bf16 driving-fp16; do
  for v in a b a b; do
    n=$(ls $out | grep -c "^${w}_$v")
    if [ $v = a ]; then python bench.py --workload $w --steps $steps --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $out/${w}_${v}$n.json
    else env $assignment python bench.py --workload $w --steps $steps --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $out/${w}_${v}$n.json; fi
  done
done
python - <<PY
import json, glob, os
for path in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(path)); r = d["roofline"]
        print(os.path.basename(path), round(d["value"], 1), round(d["ms_per_step"], 2), round(r["frac"], 4), round(r["kernel_ms_per_step"], 2), (r.get("fp32_part") or {}).get("kernel_ms_per_step"))
    except Exception as e:
        print(path, "FAILED", e)
PY
