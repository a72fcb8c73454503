#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
out=gpurun_out/r5l/conv3_chunks.txt
: > $out
run() { size=$1; label=$2; shift 2
  v=$(env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --image-size $size 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2))" 2>/dev/null)
  echo "$size $label: $v" | tee -a $out
}
for size in 224 512; do
  run $size default SRGAN_DUMMY=1
  run $size "32-row tile: 16-channel chunks" SRGAN_CONV3_CIT32=16
  run $size "64-row tile: 8-channel chunks" SRGAN_CONV3_CIT64=8
  run $size "both" SRGAN_CONV3_CIT32=16 SRGAN_CONV3_CIT64=8
  run $size "default again" SRGAN_DUMMY=2
done
