#!/bin/bash
# scratch/repeat_any.sh <count> <bench arguments...>: the same bench line several times
count=$1; shift
for i in $(seq $count); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],2), round(d['ms_per_step'],2))"
done
