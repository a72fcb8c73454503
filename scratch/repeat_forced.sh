#!/bin/bash
count=$1; shift
for i in $(seq $count); do
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --force-dp --backend nccl 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced nccl $*', round(d['value'],2), round(d['ms_per_step'],2))"
done
