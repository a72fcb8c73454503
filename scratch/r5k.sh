#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
out=gpurun_out/r5k/tangent_limit.txt
: > $out
run() { label=$1; shift
  v=$(env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2))" 2>/dev/null)
  echo "512 $label: $v" | tee -a $out
}
run "default (800 MB)" SRGAN_DUMMY=1
run "limit 1200 MB (block 2 grouped)" SRGAN_GROUPED_TANGENT_LIMIT_MB=1200
run "limit 4000 MB (blocks 2 and 3 grouped)" SRGAN_GROUPED_TANGENT_LIMIT_MB=4000
run "limit 0 (nothing grouped)" SRGAN_GROUPED_TANGENT_LIMIT_MB=0
run "default again" SRGAN_DUMMY=2
run "atomics" SRGAN_ATOMIC_SPLIT=1
run "atomics, limit 4000" SRGAN_ATOMIC_SPLIT=1 SRGAN_GROUPED_TANGENT_LIMIT_MB=4000
