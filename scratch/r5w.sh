#!/bin/bash
# round 5, call w: the fused 1x1 data gradient on 64-pixel tiles where the last round of workgroups is cheaper for it -- parity, then per-shape A / B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5w
mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_steps_gpu.py tests/test_round3_gpu.py -q -m gpu -x > $out/tests.log 2>&1
grep -E "passed|failed" $out/tests.log | tail -2 | cut -c1-300
for px in 0 128 64; do
  SRGAN_PW_RING_EPILOGUE_PIXELS=$px python bench.py --steps 12 --warmup 3 --no-cpu-baseline --single-stream --shape-report $out/shape_single_$px.txt 2>$out/err_$px.txt | grep '^{' | tail -1 > $out/bench_single_$px.json
  python -c "import json; d=json.load(open('$out/bench_single_$px.json')); print('single stream', $px, round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
done
for px in 0 128 0 128; do
  SRGAN_PW_RING_EPILOGUE_PIXELS=$px python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 > $out/bench_$px.json
  python -c "import json; d=json.load(open('$out/bench_$px.json')); print('four streams', $px, round(d['value'],2), round(d['ms_per_step'],2), d['config']['schedule_check']['max_relative_loss_difference'])"
done
for px in 0 128; do
  SRGAN_PW_RING_EPILOGUE_PIXELS=$px python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --image-size 224 2>/dev/null | grep '^{' | tail -1 > $out/bench_224_$px.json
  python -c "import json; d=json.load(open('$out/bench_224_$px.json')); print('224', $px, round(d['value'],2), round(d['ms_per_step'],2))"
done
