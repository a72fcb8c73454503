#!/bin/bash
# The round's final measurement set, one gpurun call:  scratch/measure_r5.sh <tag>   (outputs under gpurun_out/<tag>/)
tag=${1:-r05z}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1; }

# 1. PMC passes first (the bench lines below then carry `traffic`): single stream, one step, counters in separate passes
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

# 2. bench lines
python bench.py --steps 20 --warmup 5 --shape-report $out/shape_report.txt 2> $out/bench.err | line > $out/bench.json
python scratch/shapes.py $out/shape_report.txt > $out/per_kernel_table.md 2>&1
python bench.py --steps 20 --warmup 5 --single-stream --no-cpu-baseline 2>/dev/null | line > $out/bench_single_stream.json
python bench.py --steps 20 --warmup 5 --step-graph --no-cpu-baseline --no-roofline 2>/dev/null | line > $out/bench_graph_four_streams.json
python bench.py --steps 20 --warmup 5 --image-size 224 --shape-report $out/shape_report_224x224.txt 2>/dev/null | line > $out/bench_224x224.json
python scratch/shapes.py $out/shape_report_224x224.txt > $out/per_kernel_table_224x224.md 2>&1
python bench.py --steps 20 --warmup 5 --image-size 224 --single-stream --no-cpu-baseline 2>/dev/null | line > $out/bench_224x224_single_stream.json
python bench.py --steps 20 --warmup 5 --image-size 224 --step-graph --no-cpu-baseline --no-roofline 2>/dev/null | line > $out/bench_224x224_graph_four_streams.json
python bench.py --workload age-vgg-bf16 --steps 20 --warmup 5 --shape-report $out/shape_report_age_vgg64_bf16.txt 2>/dev/null | line > $out/bench_age_vgg64_bf16.json
python scratch/shapes.py $out/shape_report_age_vgg64_bf16.txt > $out/per_kernel_table_age_vgg64_bf16.md 2>&1
python bench.py --workload driving-fp16 --steps 20 --warmup 5 --shape-report $out/shape_report_driving_64x192_fp16.txt 2>/dev/null | line > $out/bench_driving_64x192_fp16.json
python scratch/shapes.py $out/shape_report_driving_64x192_fp16.txt > $out/per_kernel_table_driving_64x192_fp16.md 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl 2>/dev/null | line > $out/bench_forced_dp_nccl_world1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --grad-wire bf16 --exchange-form reduce_scatter 2>/dev/null | line > $out/bench_forced_dp_nccl_world1_bf16_reduce_scatter.json
python bench.py --workload driving-fp16 --steps 20 --warmup 5 --no-roofline --force-dp --backend nccl 2>/dev/null | line > $out/bench_driving_fp16_forced_dp_bf16_reduce_scatter.json
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line > $out/bench_100_steps.json
SRGAN_ATOMIC_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $out/bench_atomic_split.json
SRGAN_ATOMIC_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --image-size 224 2>/dev/null | line > $out/bench_224x224_atomic_split.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph 2>/dev/null | line > $out/bench_forced_dp_nccl_world1_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --step-graph --image-size 224 2>/dev/null | line > $out/bench_224x224_forced_dp_nccl_world1_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl --image-size 224 2>/dev/null | line > $out/bench_224x224_forced_dp_nccl_world1.json
SRGAN_ABI_COLLECTIVES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl 2>/dev/null | line > $out/bench_forced_dp_abi_collectives.json

# 3. rocprofv3 kernel statistics of ONE step (3 steps - 1 step, halved), single stream
for size in 512 224; do
  for steps in 1 3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${size}_s$steps -o t -- python3 bench.py --steps $steps --warmup 0 --no-cpu-baseline --no-roofline --single-stream --image-size $size > $out/prof_${size}_s$steps.log 2>&1
    cp $(find $out/prof_${size}_s$steps -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${steps}step_${size}.csv
    rm -rf $out/prof_${size}_s$steps
  done
  python scratch/per_step_stats.py $out/kernel_stats_1step_${size}.csv $out/kernel_stats_3step_${size}.csv > $out/kernel_stats_per_step_${size}.md
done

# 4. kernels in flight over time: the eager four-stream schedule and the same schedule captured as one HIP graph
for mode in eager graph; do
  extra=""; [ $mode = graph ] && extra="--step-graph"
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$mode -o t -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline $extra > $out/trace_$mode.log 2>&1
  python scratch/trace_concurrency.py $(find $out/trace_$mode -name "*kernel_trace.csv" | head -1) --tail 0.3 > $out/concurrency_$mode.txt 2>&1
  rm -rf $out/trace_$mode
done

# 5. the GPU test suite on the same snapshot
timeout 3000 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
tail -14 $out/gpu_tests.log
for f in bench bench_single_stream bench_graph_four_streams bench_224x224 bench_224x224_single_stream bench_224x224_graph_four_streams bench_age_vgg64_bf16 bench_driving_64x192_fp16 bench_forced_dp_nccl_world1 bench_forced_dp_nccl_world1_bf16_reduce_scatter bench_driving_fp16_forced_dp_bf16_reduce_scatter bench_100_steps bench_atomic_split bench_224x224_atomic_split bench_forced_dp_nccl_world1_graph bench_224x224_forced_dp_nccl_world1_graph bench_224x224_forced_dp_nccl_world1 bench_forced_dp_abi_collectives; do
  python - <<PY
import json
try:
    d = json.load(open("$out/$f.json")); r = d.get("roofline", {}); c = d["config"].get("schedule_check")
    print("$f", round(d["value"], 2), round(d["ms_per_step"], 2), r.get("frac"), r.get("step_frac_executed"), r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"), d["config"].get("host_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("max_weight_difference")))
except Exception as e:
    print("$f FAILED", e)
PY
done
cat $out/concurrency_eager.txt $out/concurrency_graph.txt
