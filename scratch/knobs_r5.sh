#!/bin/bash
# round 5: tile / split thresholds again, now that a K split costs no zero-fill launch and no atomics (one bench line each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/knobs_r5
out=gpurun_out/knobs_r5/knobs.txt
: > $out
run() { # size, label, env...
  size=$1; label=$2; shift 2
  v=$(env "$@" python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --image-size $size 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2))" 2>/dev/null)
  echo "$size $label: $v" | tee -a $out
}
for size in 224 512; do
  run $size default SRGAN_DUMMY=1
  for v in 256 640 1024; do run $size "conv3 split below $v" SRGAN_CONV3_SPLIT_BELOW=$v; done
  for v in 768 1536; do run $size "gg tile target $v" SRGAN_TILE_TARGET=$v; done
  for v in 384 512; do run $size "ksplit groups $v" SRGAN_PKS_GROUPS=$v; done
  for v in 128 256; do run $size "ring min wgs $v" SRGAN_PW_RING_MIN_WGS=$v; done
  for v in 576 1024; do run $size "pointwise min wgs $v" SRGAN_PW_MIN_WGS=$v; done
  run $size "no wave-split 1x1 kernel" SRGAN_NO_PW_KSPLIT=1
  run $size "default again" SRGAN_DUMMY=2
done
