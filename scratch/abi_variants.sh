#!/bin/bash
# forced world-1 data-parallel lines: the default transport (RCCL through the C ABI, control plane on gloo) against
# torch.distributed's nccl backend (SRGAN_ABI_COLLECTIVES=0), next to the plain single-device line
mkdir -p gpurun_out/$1
line() {  # name, env assignment, bench arguments...
  name=$1; assignment=$2; shift 2
  env $assignment python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline "$@" 2> gpurun_out/$OUT/$name.err | grep '^{' | tail -1 > gpurun_out/$OUT/$name.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/$OUT/$name.json")); c = d["config"]
    print("$name", round(d["value"], 2), round(d["ms_per_step"], 1), c.get("host_ms_per_step"), c.get("collective_transport"), "|", c.get("control_plane"))
except Exception as e:
    print("$name failed", e)
PY
}
OUT=$1
line plain X=0
line dp_abi X=0 --force-dp --backend nccl
line dp_torch SRGAN_ABI_COLLECTIVES=0 --force-dp --backend nccl
line dp_abi_bf16_rs X=0 --force-dp --backend nccl --grad-wire bf16 --exchange-form reduce_scatter
line dp_abi_224 X=0 --force-dp --backend nccl --image-size 224
line dp_torch_224 SRGAN_ABI_COLLECTIVES=0 --force-dp --backend nccl --image-size 224
