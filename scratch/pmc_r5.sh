#!/bin/bash
# The PMC passes of scratch/measure_r5.sh alone (+ the default bench line): re-binds profiles/pmc_traffic.json to the kernel sources
# after an edit that does not change the kernels (comments).   scratch/pmc_r5.sh <tag>
tag=${1:-r05z}
out=gpurun_out/${tag}_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() {  # <name> <bench arguments...>
  name=$1; shift
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    rocprofv3 --kernel-trace --pmc ${pass#*:} --output-format csv -d $out/pmc_$name/${pass%%:*} -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --single-stream "$@" > $out/pmc_${name}_${pass%%:*}.log 2>&1
  done
}
pmc crowd512
python scratch/pmc_summarise.py ${tag}_crowd512 512 16 $out/pmc_crowd512 crowd > $out/pmc_crowd512.txt 2>&1
pmc age --workload age-vgg-bf16
python scratch/pmc_summarise.py ${tag}_age_vgg64_bf16 64 128 $out/pmc_age age-vgg-bf16 > $out/pmc_age.txt 2>&1
pmc driving --workload driving-fp16
python scratch/pmc_summarise.py ${tag}_driving_64x192_fp16 64 128 $out/pmc_driving driving-fp16 > $out/pmc_driving.txt 2>&1
cp profiles/pmc_traffic.json profiles/${tag}_*pmc_per_kernel.md $out/
rm -rf $out/pmc_crowd512 $out/pmc_age $out/pmc_driving
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.json
