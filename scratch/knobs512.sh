#!/bin/bash
# one-line 512x512 bench per tuning-knob setting (scratch aid)
run() { echo -n "$* : "; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))"; }
run X=1
run SRGAN_PWG_WGS=384
run SRGAN_PWG_WGS=1536
run SRGAN_GROUP_OVERSUB=2 SRGAN_PWG_WGS=1536
run SRGAN_WGRAD3_WGS=640
run SRGAN_WGRAD3_WGS=2560
run SRGAN_WGRAD3_DEPTH=8
run SRGAN_WGRAD3_DEPTH=16
run SRGAN_PWG_DEPTH=8
run X=2
