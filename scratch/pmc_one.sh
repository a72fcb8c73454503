#!/bin/bash
# PMC passes of ONE training step of one workload:  scratch/pmc_one.sh <tag> <name> <image_size> <batch> <workload> [bench arguments...]
# (counters in separate passes, never combined with other trace domains; SRGAN_BENCH_NO_HOST_STEP: the process runs exactly
# one iteration, so the per-step figures of scratch/pmc_summarise.py are per step)
tag=$1; name=$2; size=$3; batch=$4; workload=$5; shift 5
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SRGAN_BENCH_NO_HOST_STEP=1
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc ${pass#*:} --output-format csv -d $out/pmc_$name/${pass%%:*} -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --single-stream --workload $workload "$@" > $out/pmc_${name}_${pass%%:*}.log 2>&1
done
python scratch/pmc_summarise.py ${tag}_$name $size $batch $out/pmc_$name $workload > $out/pmc_$name.txt 2>&1
cp profiles/${tag}_${name}_pmc_per_kernel.md $out/ 2>/dev/null
rm -rf $out/pmc_$name
