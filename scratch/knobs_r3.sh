#!/bin/bash
# one-line single-stream 512x512 bench per tuning-knob setting (scratch aid, round 3)
run() { echo -n "$* : "; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --single-stream 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))"; }
run X=1
run SRGAN_PW_MI_SHORT=1
run SRGAN_PW_NI=2
run SRGAN_PW_MIN_WGS=512
run SRGAN_PW_MIN_WGS=1024
run SRGAN_PW_MIN_WGS=1536
run SRGAN_PWG_WGS=512
run SRGAN_PWG_WGS=1024
run SRGAN_GROUP_OVERSUB=2
run SRGAN_GROUP_OVERSUB=8
run SRGAN_PWG_DEPTH=4
run X=2
