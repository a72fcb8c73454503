#!/bin/bash
# The round's final measurement set, one gpurun call:  scratch/measure_r6.sh <tag>   (outputs under gpurun_out/<tag>/)
tag=${1:-r06z}
out=gpurun_out/$tag
mkdir -p $out
line() { grep '^{' | tail -1; }

# 1. PMC passes first (the bench lines below then carry `traffic`): single stream, ONE step per process, counters in separate passes
bash scratch/pmc_one.sh $tag crowd512 512 16 crowd
bash scratch/pmc_one.sh $tag age_vgg64_bf16 64 128 age-vgg-bf16
bash scratch/pmc_one.sh $tag driving_64x192_fp16 64 128 driving-fp16
cp profiles/pmc_traffic.json $out/

# 2. bench lines.  The first is the driver's own invocation (defaults; carries `secondary` and `cpu_baseline`)
python bench.py 2> $out/bench_default.err | line > $out/bench_default.json
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --shape-report $out/shape_report.txt 2> $out/bench.err | line > $out/bench.json
python scratch/shapes.py $out/shape_report.txt > $out/per_kernel_table.md 2>&1
python bench.py --steps 20 --warmup 5 --single-stream --no-cpu-baseline --no-secondary 2>/dev/null | line > $out/bench_single_stream.json
python bench.py --steps 20 --warmup 5 --step-graph --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line > $out/bench_graph_four_streams.json
SRGAN_NO_BLOCKED_F32=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line > $out/bench_generator_on_nchw_kernels.json
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line > $out/bench_100_steps.json
python bench.py --steps 20 --warmup 5 --image-size 224 --no-secondary --shape-report $out/shape_report_224x224.txt 2>/dev/null | line > $out/bench_224x224.json
python scratch/shapes.py $out/shape_report_224x224.txt > $out/per_kernel_table_224x224.md 2>&1
python bench.py --steps 20 --warmup 5 --image-size 224 --single-stream --no-cpu-baseline --no-secondary 2>/dev/null | line > $out/bench_224x224_single_stream.json
python bench.py --steps 20 --warmup 5 --image-size 224 --step-graph --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line > $out/bench_224x224_graph_four_streams.json
python bench.py --workload age-vgg-bf16 --steps 100 --warmup 5 --shape-report $out/shape_report_age_vgg64_bf16.txt 2>/dev/null | line > $out/bench_age_vgg64_bf16.json
python scratch/shapes.py $out/shape_report_age_vgg64_bf16.txt > $out/per_kernel_table_age_vgg64_bf16.md 2>&1
python bench.py --workload driving-fp16 --steps 100 --warmup 5 --shape-report $out/shape_report_driving_64x192_fp16.txt 2>/dev/null | line > $out/bench_driving_64x192_fp16.json
python scratch/shapes.py $out/shape_report_driving_64x192_fp16.txt > $out/per_kernel_table_driving_64x192_fp16.md 2>&1
SRGAN_NO_STORAGE16=1 python bench.py --workload age-vgg-bf16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $out/bench_age_vgg64_bf16_fp32_storage.json
SRGAN_NO_STORAGE16=1 python bench.py --workload driving-fp16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $out/bench_driving_64x192_fp16_fp32_storage.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --force-dp --backend nccl 2>/dev/null | line > $out/bench_forced_dp_world1.json
SRGAN_ABI_COLLECTIVES=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --force-dp --backend nccl 2>/dev/null | line > $out/bench_forced_dp_world1_torch_distributed.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --force-dp --backend nccl --grad-wire bf16 --exchange-form reduce_scatter 2>/dev/null | line > $out/bench_forced_dp_world1_bf16_reduce_scatter.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --force-dp --backend nccl --step-graph 2>/dev/null | line > $out/bench_forced_dp_world1_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --force-dp --backend nccl --image-size 224 2>/dev/null | line > $out/bench_224x224_forced_dp_world1.json
python bench.py --workload driving-fp16 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl 2>/dev/null | line > $out/bench_driving_fp16_forced_dp_bf16_reduce_scatter.json
python bench.py --workload age-vgg-bf16 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --force-dp --backend nccl 2>/dev/null | line > $out/bench_age_vgg_bf16_forced_dp.json

# 3. rocprofv3 kernel statistics of ONE step (3 steps - 1 step, halved), single stream
bash scratch/stats_one.sh $tag 512
bash scratch/stats_one.sh $tag 224 --image-size 224
bash scratch/stats_one.sh $tag age_vgg64_bf16 --workload age-vgg-bf16
bash scratch/stats_one.sh $tag driving_64x192_fp16 --workload driving-fp16

# 4. the host's share with two CPUs per rank
bash scratch/host_budget.sh $tag/host > $out/host_budget.log 2>&1
cp $out/host/table.md $out/host_budget_two_cpus.md

# 5. the GPU test suite on the same snapshot
timeout 3000 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
tail -14 $out/gpu_tests.log
for f in $out/bench*.json; do
  python - <<PY
import json
try:
    d = json.load(open("$f")); r = d.get("roofline", {}); c = d["config"].get("schedule_check")
    print("$f".split("/")[-1][:-5], round(d["value"], 2), round(d["ms_per_step"], 2), r.get("frac"), r.get("step_frac_executed"), r.get("traffic"), r.get("algorithmic_bytes_per_launch"), (d.get("cpu_baseline") or {}).get("value"), d["config"].get("host_ms_per_step"), c if isinstance(c, str) else (c["max_relative_loss_difference"], c.get("max_weight_difference")), sorted((d.get("secondary") or {}).keys()))
except Exception as e:
    print("$f FAILED", e)
PY
done
