#!/bin/bash
# round 5: runtime environment switches that act on launch latency (no code change): kernel arguments in device memory
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_env
mkdir -p $out
run() { size=$1; label=$2; shift 2
  v=$(env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --image-size $size 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('host_ms_per_step'))" 2>/dev/null)
  echo "$size $label: $v" | tee -a $out/env.txt
}
: > $out/env.txt
run 224 "default" SRGAN_DUMMY=1
run 224 "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run 224 "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run 224 "default (again)" SRGAN_DUMMY=2
run 224 "HIP_FORCE_DEV_KERNARG=1 (again)" HIP_FORCE_DEV_KERNARG=1
run 512 "default" SRGAN_DUMMY=1
run 512 "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run 512 "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run 224 "graph, default" SRGAN_DUMMY=1 SRGAN_X=1
