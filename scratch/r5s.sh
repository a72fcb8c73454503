#!/bin/bash
# round 5, call s: work-proportional shares for the register-streamed grouped kernel as well (the ragged planes of 224 x 224)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5s
mkdir -p $out
res=$out/shares.txt
: > $res
run() { size=$1; label=$2; tag=$3; shift 3
  env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --image-size $size --shape-report $out/shape_$tag.txt 2>$out/err_$tag.txt | grep '^{' | tail -1 > $out/bench_$tag.json
  v=$(python -c "import json,sys; d=json.load(open('$out/bench_$tag.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['config']['schedule_check']['max_relative_loss_difference'])" 2>/dev/null)
  python scratch/shapes.py $out/shape_$tag.txt > $out/table_$tag.md 2>&1
  w=$(grep "^| pointwise_wgrad" $out/table_$tag.md | cut -c1-90)
  echo "$size $label: $v $w" | tee -a $res
}
run 224 "equal shares on the streamed kernel (default)" a SRGAN_DUMMY=1
run 224 "shares by work on the streamed kernel" b SRGAN_PWG_WORK_SHARES=1
run 224 "equal shares (again)" c SRGAN_DUMMY=2
run 224 "shares by work (again)" d SRGAN_PWG_WORK_SHARES=1
run 224 "by work, oversubscription 2" e SRGAN_PWG_WORK_SHARES=1 SRGAN_GROUP_OVERSUB=2
run 224 "by work, oversubscription 8" f SRGAN_PWG_WORK_SHARES=1 SRGAN_GROUP_OVERSUB=8
run 512 "default" g SRGAN_DUMMY=1
