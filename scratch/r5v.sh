#!/bin/bash
# round 5, call v: what a per-shape choice of the LDS-DMA 1x1 kernel's tile width could gain (all launches forced to 128 / 64 / 32 pixels, per-shape times compared)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5v
mkdir -p $out
for px in 0 128 64 32; do
  SRGAN_PW_RING_FORCE_PIXELS=$px python bench.py --steps 12 --warmup 3 --no-cpu-baseline --single-stream --shape-report $out/shape_$px.txt 2>$out/err_$px.txt | grep '^{' | tail -1 > $out/bench_$px.json
  python -c "import json; d=json.load(open('$out/bench_$px.json')); print($px, round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
done
