#!/bin/bash
# one-line 512x512 bench per tuning-knob setting, side streams on (scratch aid, round 3)
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))"; }
run X=1
run SRGAN_PW_MI=4
run SRGAN_PWG_WGS=1024
run SRGAN_PWG_WGS=512
run SRGAN_PW_MIN_WGS=512
run SRGAN_PW_MIN_WGS=384
run SRGAN_GROUP_OVERSUB=2
run SRGAN_WGRAD3_WGS=640
run SRGAN_PW_NI=2
run X=2
