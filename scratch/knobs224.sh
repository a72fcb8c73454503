#!/bin/bash
# one-line 224x224 bench per tuning-knob setting (scratch aid)
run() { echo -n "$* : "; env "$@" python bench.py --image-size 224 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
run X=1
run SRGAN_PWG_DEPTH=4
run SRGAN_PWG_DEPTH=2
run SRGAN_PWG_WGS=1536
run SRGAN_PW_MIN_WGS=512
run SRGAN_PW_MIN_WGS=1024
run SRGAN_PW_MIN_WGS=1536
run SRGAN_WGRAD3_DEPTH=2
run SRGAN_WGRAD3_WGS=2560
run SRGAN_CONV3_SPLIT_BELOW=1024
run SRGAN_CONV3_SPLIT_BELOW=256
run SRGAN_TILE_TARGET=512
