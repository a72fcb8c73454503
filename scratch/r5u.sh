#!/bin/bash
# round 5, call u: the staged 1x1 weight gradient with its gy rows in (accumulation) registers, four stages of x -- parity, then A / B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5u
mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_steps_gpu.py -q -m gpu -x > $out/tests.log 2>&1
grep -E "passed|failed" $out/tests.log | tail -2 | cut -c1-300
res=$out/rows_in_registers.txt
: > $res
run() { size=$1; label=$2; tag=$3; shift 3
  env "$@" python bench.py --steps 16 --warmup 4 --no-cpu-baseline --image-size $size --shape-report $out/shape_$tag.txt 2>$out/err_$tag.txt | grep '^{' | tail -1 > $out/bench_$tag.json
  v=$(python -c "import json,sys; d=json.load(open('$out/bench_$tag.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['config']['schedule_check']['max_relative_loss_difference'])" 2>/dev/null)
  python scratch/shapes.py $out/shape_$tag.txt > $out/table_$tag.md 2>&1
  w=$(grep "^| pointwise_wgrad" $out/table_$tag.md | cut -c1-90)
  echo "$size $label: $v $w" | tee -a $res
}
run 512 "gy rows in registers, four x stages (default)" a SRGAN_DUMMY=1
run 512 "gy rows through LDS, two stages" b SRGAN_PWL_STAGED_GY=1
run 512 "registers (again)" c SRGAN_DUMMY=2
run 512 "LDS (again)" d SRGAN_PWL_STAGED_GY=1
run 512 "registers, oversubscription 3" e SRGAN_PWL_OVERSUB=3
run 224 "registers" i SRGAN_DUMMY=1
run 224 "LDS" j SRGAN_PWL_STAGED_GY=1
