#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
out=gpurun_out/r5j/ksplit_groups_224.txt
: > $out
run() { label=$1; shift
  v=$(env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --image-size 224 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],2), round(d['ms_per_step'],2))" 2>/dev/null)
  echo "224 $label: $v" | tee -a $out
}
run "no wave-split kernel" SRGAN_NO_PW_KSPLIT=1
for g in 26 80 100 256; do run "wave-split kernel up to $g groups" SRGAN_PKS_GROUPS=$g; done
run "no wave-split kernel (again)" SRGAN_NO_PW_KSPLIT=1
run "default (256)" SRGAN_DUMMY=1
