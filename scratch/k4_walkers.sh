mkdir -p gpurun_out/r6k4
for w in 384 512 768 256; do
  for i in 1 2; do
    SRGAN_H_K4_WALKERS=$w python bench.py --workload driving-fp16 --steps 100 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r6k4/w${w}_$i.json
  done
done
python - <<PY
import json, glob, os
for path in sorted(glob.glob("gpurun_out/r6k4/*.json")):
    d = json.load(open(path)); r = d["roofline"]
    print(os.path.basename(path), round(d["value"], 1), round(d["ms_per_step"], 2), round(r["frac"], 4), round(r["kernel_ms_per_step"], 2))
PY
