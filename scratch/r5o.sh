#!/bin/bash
# round 5, call o: the 1x1 weight gradient staged through LDS (128 x 128 tiles per workgroup) -- parity, then A / B with per-kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5o
mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_steps_gpu.py tests/test_round3_gpu.py -q -m gpu -x > $out/tests.log 2>&1
tail -5 $out/tests.log | cut -c1-300
res=$out/staged_wgrad.txt
: > $res
run() { size=$1; label=$2; tag=$3; shift 3
  env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --image-size $size --shape-report $out/shape_$tag.txt 2>$out/err_$tag.txt | grep '^{' | tail -1 > $out/bench_$tag.json
  v=$(python -c "import json,sys; d=json.load(open('$out/bench_$tag.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['config']['schedule_check']['max_relative_loss_difference'])" 2>/dev/null)
  python scratch/shapes.py $out/shape_$tag.txt > $out/table_$tag.md 2>&1
  w=$(grep "^| pointwise_wgrad" $out/table_$tag.md | cut -c1-90)
  echo "$size $label: $v $w" | tee -a $res
}
run 512 "staged (default: 512 wgs x2, depth 8)" a SRGAN_DUMMY=1
run 512 "register-streamed (round 4/5 kernel)" b SRGAN_NO_PW_WGRAD_LDS=1
run 512 "staged, oversubscription 1" c SRGAN_PWL_OVERSUB=1
run 512 "staged, oversubscription 4" d SRGAN_PWL_OVERSUB=4
run 512 "staged, depth 16" e SRGAN_PWL_DEPTH=16
run 512 "staged, depth 4, oversubscription 4" f SRGAN_PWL_DEPTH=4 SRGAN_PWL_OVERSUB=4
run 224 "staged" g SRGAN_DUMMY=1
run 224 "register-streamed" h SRGAN_NO_PW_WGRAD_LDS=1
