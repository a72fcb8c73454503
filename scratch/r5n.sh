#!/bin/bash
# round 5, call n: the LDS-DMA 1x1 kernel with its K range split over workgroups (wide tiles on few-pixel launches) -- parity, then A / B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py -q -m gpu -x -k "conv_passes or fused_batch_norm or reproducible or split" > gpurun_out/r5n/tests.log 2>&1
tail -5 gpurun_out/r5n/tests.log | cut -c1-300
out=gpurun_out/r5n/ring_split.txt
: > $out
run() { size=$1; label=$2; shift 2
  v=$(env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --image-size $size 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2))" 2>/dev/null)
  echo "$size $label: $v" | tee -a $out
}
run 512 "ring K split (default)" SRGAN_DUMMY=1
run 512 "no ring K split" SRGAN_NO_PW_RING_SPLIT=1
run 512 "ring K split, no wave-split kernel (16 x 16 planes on the ring too)" SRGAN_NO_PW_KSPLIT=1
run 512 "ring K split, 768 workgroups" SRGAN_PW_RING_SPLIT_WGS=768
run 512 "ring K split, 384 workgroups" SRGAN_PW_RING_SPLIT_WGS=384
run 512 "ring K split (again)" SRGAN_DUMMY=2
run 512 "no ring K split (again)" SRGAN_NO_PW_RING_SPLIT=1
run 224 "ring K split (default)" SRGAN_DUMMY=1
run 224 "no ring K split" SRGAN_NO_PW_RING_SPLIT=1
