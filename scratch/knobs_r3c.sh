#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --single-stream --shape-report /tmp/s.txt 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))"; python scratch/shapes.py /tmp/s.txt | grep -E "^\| (pointwise_wgrad|conv3x3_wgrad)" | cut -c1-80; }
run X=1
run SRGAN_GROUP_OVERSUB=8
run SRGAN_GROUP_OVERSUB=16
run SRGAN_GROUP_OVERSUB=2
run SRGAN_PWG_WGS=1536 SRGAN_GROUP_OVERSUB=8
