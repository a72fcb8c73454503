"""Markdown table of the round's bench lines (gpurun_out/<tag>/bench*.json) for DESIGN.md."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r06z'
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', tag)
rows = [
    ('bench_default', '`python bench.py` (the driver\'s invocation: defaults, carries `secondary` and `cpu_baseline`)'),
    ('bench', '`--steps 20 --warmup 5` (headline schedule: four streams)'),
    ('bench_generator_on_nchw_kernels', '... `SRGAN_NO_BLOCKED_F32=1` (the generator on the NCHW kernels, as in round 5)'),
    ('bench_single_stream', '`--single-stream`'),
    ('bench_graph_four_streams', '`--step-graph` (the four chains as branches of one HIP graph)'),
    ('bench_100_steps', '`--steps 100`'),
    ('bench_forced_dp_world1', '`--force-dp --backend nccl` (default transport: RCCL through the C ABI; three compute streams + the communication stream)'),
    ('bench_forced_dp_world1_torch_distributed', '... `SRGAN_ABI_COLLECTIVES=0` (torch.distributed\'s nccl backend)'),
    ('bench_forced_dp_world1_bf16_reduce_scatter', '... `--grad-wire bf16 --exchange-form reduce_scatter`'),
    ('bench_forced_dp_world1_graph', '... `--step-graph` (one compute stream + the communication stream, replayed)'),
    ('bench_224x224', '`--image-size 224`'),
    ('bench_224x224_single_stream', '`--image-size 224 --single-stream`'),
    ('bench_224x224_graph_four_streams', '`--image-size 224 --step-graph`'),
    ('bench_224x224_forced_dp_world1', '`--image-size 224 --force-dp --backend nccl`'),
    ('bench_age_vgg64_bf16', '`--workload age-vgg-bf16 --steps 100` (BASELINE.json configs[1]; 16-bit data path)'),
    ('bench_age_vgg64_bf16_fp32_storage', '... `SRGAN_NO_STORAGE16=1` (round 5\'s path: fp32 tensors, bf16 operands)'),
    ('bench_age_vgg_bf16_forced_dp', '... `--force-dp --backend nccl`'),
    ('bench_driving_64x192_fp16', '`--workload driving-fp16 --steps 100` (configs[4]; fp16 storage, fp32 penalty chain on blocked fp32)'),
    ('bench_driving_64x192_fp16_fp32_storage', '... `SRGAN_NO_STORAGE16=1` (round 5\'s path)'),
    ('bench_driving_fp16_forced_dp_bf16_reduce_scatter', '... `--force-dp --backend nccl` (bf16 buckets, reduce-scatter)'),
]
print('| line | images/s | ms / step | contraction kernels (single-stream brackets) | host ms to enqueue a step | schedule check (losses / weights) |')
print('|---|---|---|---|---|---|')
for name, label in rows:
    path = os.path.join(root, name + '.json')
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        print(f'| {label} | (no line) | | | | |')
        continue
    r = d.get('roofline') or {}
    c = d['config'].get('schedule_check')
    check = c if isinstance(c, str) else f"{c['max_relative_loss_difference']:.1e} / {c.get('max_weight_difference', float('nan')):.1e} (limit {c['limit']:.0e})"
    if isinstance(c, str):
        check = 'n/a (one stream)'
    kernels = ''
    if r:
        kernels = f"{r['achieved']:.1f} TF/s = **{r['frac']:.3f}** of {r['peak']:.0f}; {r.get('kernel_ms_per_step', 0):.1f} ms over {r.get('launches')} launches"
        if r.get('step_frac_executed'):
            kernels += f"; step {r['step_frac_executed']:.3f}"
        if r.get('traffic'):
            kernels += f"; traffic {r['traffic'] / 1e6:.1f} MB / launch vs {r['algorithmic_bytes_per_launch'] / 1e6:.1f} algorithmic"
    cpu = d.get('cpu_baseline')
    if cpu and cpu.get('value'):
        kernels += f"; CPU oracle {cpu['value']:.3f} images/s at batch {cpu.get('batch')}"
        like = cpu.get('like_for_like')
        if like:
            kernels += f" ({like['value']:.3f} at batch {like['batch']}, committed)"
    print(f"| {label} | **{d['value']:.2f}** | {d['ms_per_step']:.1f} | {kernels} | {d['config'].get('host_ms_per_step')} | {check} |")
