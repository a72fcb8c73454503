#!/bin/bash
# the host's share of an iteration when a rank has TWO CPUs (eight ranks on the pool's 16-CPU quota), measured
# on one GPU with the process confined by taskset (not under rocprofv3): images/s and config.host_ms_per_step per line.
# scratch/host_budget.sh <tag>
tag=${1:-r06_host}
out=gpurun_out/$tag
mkdir -p $out
line() { grep '^{' | tail -1; }
run() {  # <name> <cpus or all> <bench arguments...>
  name=$1; cpus=$2; shift 2
  if [ "$cpus" = all ]; then python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary "$@" 2>/dev/null | line > $out/$name.json
  else taskset -c $cpus python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary "$@" 2>/dev/null | line > $out/$name.json; fi
}
for cpus in all 0,1; do
  c=${cpus/,/_}
  # (dp_eager = the default transport: RCCL through the C ABI; dp_torch = SRGAN_ABI_COLLECTIVES=0: torch.distributed's nccl backend)
  run crowd512_dp_eager_$c $cpus --force-dp --backend nccl
  run crowd224_dp_eager_$c $cpus --force-dp --backend nccl --image-size 224
  run driving_dp_eager_$c $cpus --force-dp --backend nccl --workload driving-fp16
  run age_dp_eager_$c $cpus --force-dp --backend nccl --workload age-vgg-bf16
  SRGAN_ABI_COLLECTIVES=0 run crowd512_dp_torch_$c $cpus --force-dp --backend nccl
  SRGAN_ABI_COLLECTIVES=0 run crowd224_dp_torch_$c $cpus --force-dp --backend nccl --image-size 224
  run crowd512_dp_graph_$c $cpus --force-dp --backend nccl --step-graph
  run crowd224_dp_graph_$c $cpus --force-dp --backend nccl --step-graph --image-size 224
  run crowd224_plain_$c $cpus --image-size 224
done
python3 - <<PY
import json, glob, os
rows = []
for path in sorted(glob.glob('$out/*.json')):
    try:
        d = json.load(open(path))
        rows.append((os.path.basename(path)[:-5], d['value'], d['ms_per_step'], d['config'].get('host_ms_per_step')))
    except Exception as error:
        rows.append((os.path.basename(path)[:-5], None, None, str(error)))
with open('$out/table.md', 'w') as handle:
    handle.write('| line | images/s | ms / step | host ms to enqueue a step |\n|---|---|---|---|\n')
    for name, value, ms, host in rows:
        handle.write(f'| {name} | {value and round(value, 2)} | {ms and round(ms, 2)} | {host} |\n')
print(open('$out/table.md').read())
PY
