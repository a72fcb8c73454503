#!/bin/bash
# kernel traces of the forced world-1 data-parallel step: torch.distributed against the C ABI's collectives
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
for v in torch abi; do
  case $v in torch) export SRGAN_ABI_COLLECTIVES=0;; abi) export SRGAN_ABI_COLLECTIVES=1;; esac
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --force-dp --backend nccl > $O/$v.out 2> $O/$v.err
  f=$(find $O/trace_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v $f"; [ -n "$f" ] || continue; head -1 "$f"; grep -i "nccl\|rccl" "$f" | head; 
  python3 - <<PY
import csv,sys
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
print("total kernel ms", tot/1e6, "calls", n)
PY
  find $O/trace_$v -name '*kernel_trace.csv' -size +60M -delete
done
