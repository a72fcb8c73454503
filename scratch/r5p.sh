#!/bin/bash
# round 5, call p: the staged 1x1 weight gradient with evenly spread column blocks, three stages on narrow tiles, mixed groups
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5p
mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_steps_gpu.py -q -m gpu -x > $out/tests.log 2>&1
grep -E "passed|failed" $out/tests.log | tail -2 | cut -c1-300
res=$out/staged_wgrad.txt
: > $res
run() { size=$1; label=$2; tag=$3; shift 3
  env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --image-size $size --shape-report $out/shape_$tag.txt 2>$out/err_$tag.txt | grep '^{' | tail -1 > $out/bench_$tag.json
  v=$(python -c "import json,sys; d=json.load(open('$out/bench_$tag.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['config']['schedule_check']['max_relative_loss_difference'])" 2>/dev/null)
  python scratch/shapes.py $out/shape_$tag.txt > $out/table_$tag.md 2>&1
  w=$(grep "^| pointwise_wgrad" $out/table_$tag.md | cut -c1-90)
  echo "$size $label: $v $w" | tee -a $res
}
run 512 "staged (default)" a SRGAN_DUMMY=1
run 512 "register-streamed (round 4/5 kernel)" b SRGAN_NO_PW_WGRAD_LDS=1
run 512 "staged from 256 input channels" c SRGAN_PWL_MIN_CI=256
run 512 "staged from 512 input channels" d SRGAN_PWL_MIN_CI=512
run 512 "staged from 768 input channels" e SRGAN_PWL_MIN_CI=768
run 512 "staged, oversubscription 6" f SRGAN_PWL_OVERSUB=6
run 512 "staged, oversubscription 3" g SRGAN_PWL_OVERSUB=3
run 512 "staged, depth 16" h SRGAN_PWL_DEPTH=16
run 224 "staged" i SRGAN_DUMMY=1
run 224 "register-streamed" j SRGAN_NO_PW_WGRAD_LDS=1
