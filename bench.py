"""Headline benchmark: SRGAN training images/s on the crowd workload (BASELINE.json configs[2]/[3]).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started by ``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) the
script IS a rank; started plainly (``python bench.py --gpus 8``) it launches its N ranks itself -- as a child
``torch.distributed.run`` on 127.0.0.1, BEFORE this process makes any GPU call -- relays rank 0's JSON line and exits
with the children's status.

One "step" = one iteration of the reference's training loop (srgan.py:104-118): ``dnn_training_step`` +
``gan_training_step`` (8 D forwards / 6 D backwards / GP double backward / G forward x2 + backward / 3 Adam
updates in the reference's schedule) on a synthetic batch of 16 crowd images of 512x512 per GPU (weak scaling:
global batch = 16 * N), inputs resident in HBM.  The discriminator's convolution weights are scaled (``GP_SCALE``) so
that the gradient penalty is ACTIVE (at the default initialisation every gradient norm is < 1 and the penalty's own
backward would multiply exact zeros); ``config.gradient_penalty_last`` is its value in the last timed step.
Prints ONE JSON line on rank 0.

Extra legs (rank 0, N = 1 only):
* ``roofline``: every contraction launch (the MFMA kernel family = the dominant kernels) of one extra step is
  bracketed with HIP events on its launch stream inside libsrgan_hip.so; achieved = sum(executed 2*M*N*K) /
  sum(event durations), against the fp32 MFMA peak (157.3 TF/s, MI355X_MICROARCH.md).
  ``step_frac_executed`` = the same executed FLOPs / the WHOLE step time / peak (everything else counted as lost time);
  ``step_tflops_as_written_schedule`` divides BASELINE.md's as-written FLOPs of the reference schedule by the step time: a
  speed-up statement, not a roofline fraction.  ``traffic`` comes from the rocprofv3 PMC passes committed under
  profiles/ and is null unless that file was measured at this image size / batch on these kernel sources.
* ``cpu_baseline``: the CPU oracle (PyTorch-CPU fp32 restatement of the reference step, kind "port") timed on
  this box's host cores on a bounded sample (1 warm-up + 3 timed iterations at the same 512x512 shape, batch 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this platform needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails in the legacy mode)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# algorithmic FLOPs per image of the reference's as-written schedule (conv + linear, 2*MAC)
ALGORITHMIC_GFLOP_PER_IMAGE = {512: 1188.7 + 135.6, 224: 227.5 + 26.0}
# Factor on every convolution weight of D that makes the gradient penalty active at the benchmark shapes (measured with
# scratch/gp_scale.py: mean gradient norm of the interpolates ~2-4 instead of ~0.01 at the default initialisation).
GP_SCALE = {512: 1.27, 224: 1.27}


RESTORE_PERIOD = 16                      # timed steps between weight restores (see main)
LOW_PRECISION_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 (MI355X_MICROARCH.md); never the 2:1-sparsity figure
# The other BASELINE.json configurations that name a precision mode (secondary bench lines; the headline stays crowd).
WORKLOADS = {
    'crowd': None,
    'age-vgg-bf16': dict(application='age', architecture='vgg', image_size=64, batch_per_gpu=128, gp_scale=1.3, dtype='bf16',
                         settings=dict(compute_dtype='bf16', gradient_penalty_dtype='bf16', storage_dtype='bf16',
                                       matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                                       gradient_penalty_multiplier=1e2),
                         name='age SRGAN, VGG-16 D/DNN + DCGAN G on 64x64 faces (BASELINE.json configs[1]), bf16: activations, gradients '
                              'and weight shadows of the three networks stored as bf16 in the blocked layout with fused activations (gradient-penalty '
                              'chain included), fp32 master weights / Adam / losses'),
    'driving-fp16': dict(application='driving', image_size=(64, 192), batch_per_gpu=128, gp_scale=3.0, dtype='f16',
                         settings=dict(compute_dtype='f16', gradient_penalty_dtype='f32', storage_dtype='f16', loss_scale=256.0,
                                       matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                                       gradient_penalty_multiplier=1e2,
                                       # comm-sensitive under data parallelism: half the bytes per link,
                                       # S/N to every peer over all seven xGMI links instead of S around one ring
                                       gradient_wire_dtype='bf16', gradient_exchange_form='reduce_scatter'),
                         name='driving SRGAN, DCGAN D/DNN/G on 64x192 frames (BASELINE.json configs[4]), fp16: activations, gradients and '
                              'weight shadows of the three networks stored as fp16 in the blocked layout with fused leaky_relu; the '
                              'gradient-penalty chain in fp32 on fp32 tensors; fp32 master weights / Adam / losses; static loss scale 256'),
}


def parse():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=3)
    parser.add_argument('--warmup', type=int, default=1)
    parser.add_argument('--image-size', type=int, default=512)
    parser.add_argument('--batch-per-gpu', type=int, default=16)
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-roofline', action='store_true')
    parser.add_argument('--cpu-baseline-child', action='store_true', help=argparse.SUPPRESS)
    parser.add_argument('--cpu-baseline-batch', type=int, default=4,
                        help='batch of the CPU oracle leg.  Measured on the GPU box\'s 16 host cores (profiles/r04_cpu_baseline_*.json): '
                             '0.297 images/s at batch 1 but 0.133 at the GPU leg\'s own batch of 16 (two minutes per iteration, 90 GiB): '
                             'the per-image rate falls with the batch, so batch 1 would flatter the CPU.  Default 4: the largest '
                             'batch whose 1 + 2 iterations fit the default run')
    parser.add_argument('--cpu-baseline-timed', type=int, default=2, help='timed iterations of the CPU oracle leg')
    parser.add_argument('--overlap-dnn', action='store_true', help=argparse.SUPPRESS)      # (the default now; kept for old command lines)
    parser.add_argument('--single-stream', action='store_true',
                        help='timed region on ONE stream.  Default: the DNN step, the gradient-penalty chain and the '
                             'un-differentiated D(unlabeled) forward of the generator step run on streams of their own next to '
                             'the main chain (the event-bracketed roofline step is always single-stream)')
    parser.add_argument('--backend', default='nccl', help='torch.distributed backend for --gpus > 1 (nccl = RCCL)')
    parser.add_argument('--force-dp', action='store_true',
                        help='keep the data-parallel path on at world size 1: the feature-sum all-reduce, the asynchronous '
                             'gradient buckets and the broadcasts go through --backend on ONE rank (first contact with '
                             'RCCL on a one-GPU box)')
    parser.add_argument('--single-device', action='store_true',
                        help='testing aid: every rank uses cuda:0 (with --backend gloo on a one-GPU box)')
    parser.add_argument('--shape-report', default=None, help='write the per-shape contraction timing table here')
    parser.add_argument('--reference-schedule', action='store_true',
                        help='replay the reference forward/backward order instead of sharing forwards')
    parser.add_argument('--workload', default='crowd', choices=sorted(WORKLOADS),
                        help='crowd (the headline, BASELINE.json configs[2]/[3]) or one of the mixed-precision configurations: '
                             'age-vgg-bf16 (configs[1]), driving-fp16 (configs[4]: fp16 with the gradient-penalty chain in fp32)')
    parser.add_argument('--gp-scale', type=float, default=None,
                        help='factor on D\'s convolution weights (default: GP_SCALE for the image size; 1 = default '
                             'initialisation, gradient penalty inactive)')
    parser.add_argument('--no-overlap-exchange', action='store_true',
                        help='data parallel: wait for each gradient all-reduce where it is started')
    parser.add_argument('--step-graph', action='store_true',
                        help='capture the iteration once as a HIP graph and replay it (single device): removes the host '
                             'enqueue time, which bounds the step at 224x224')
    parser.add_argument('--grad-wire', default=None, choices=['f32', 'bf16'],
                        help='data parallel: dtype of the gradient buckets on the wire (default: f32; the driving-fp16 '
                             'workload: bf16); the master gradients stay fp32')
    parser.add_argument('--exchange-form', default=None, choices=['all_reduce', 'reduce_scatter'],
                        help='data parallel: all-reduce (default) or reduce-scatter + all-gather buckets')
    parser.add_argument('--master-port', type=int, default=None, help='self-launch only: rendezvous port on 127.0.0.1')
    parser.add_argument('--no-secondary', action='store_true',
                        help='crowd headline on one GPU: do not append the two mixed-precision configurations (age-vgg-bf16, '
                             'driving-fp16) as `secondary` entries (each is this script run as a child process)')
    parser.add_argument('--secondary-steps', type=int, default=100, help='timed steps of each secondary configuration')
    return parser.parse_args()


def build_experiment(args, dp):
    import srgan_amd  # noqa: F401
    from srgan_amd.settings import Settings
    from srgan_amd.crowd.srgan import CrowdExperiment
    from srgan_amd.utility import SummaryWriter, seed_all
    settings = Settings()                                   # run.py:57-68 crowd hyper-parameters
    workload = WORKLOADS[args.workload]
    if workload is not None:
        args.batch_per_gpu, args.image_size = workload['batch_per_gpu'], workload['image_size']
    settings.batch_size = args.batch_per_gpu * (dp.world_size if dp else 1)
    settings.image_patch_size = settings.label_patch_size = args.image_size
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e3, 1e2
    settings.gradient_penalty_multiplier, settings.map_multiplier = 1e2, 1e-3
    settings.learning_rate = 1e-4
    settings.reference_schedule = args.reference_schedule
    streams = side_streams(args)
    settings.overlap_dnn_step = streams and not os.environ.get('SRGAN_NO_DNN_STREAM')
    # (the grouped weight gradients on a stream of their own gained +4 % next to ONE chain, but with three chains in flight
    # a fifth / sixth / seventh stream aliases onto the four hardware queues of the HIP runtime and creates false
    # dependencies between the chains: 70.3 or 74.9 images/s from run to run, against a steady 75.1 without it)
    settings.wgrad_stream = streams and bool(os.environ.get('SRGAN_BENCH_WGRAD_STREAM'))
    settings.overlap_generator_forwards = streams and not os.environ.get('SRGAN_NO_AUX_STREAM')
    settings.overlap_gradient_penalty = streams and not os.environ.get('SRGAN_NO_PENALTY_STREAM')
    settings.overlap_gradient_exchange = not args.no_overlap_exchange
    # the DCGAN stacks' 4x4 / stride 2 stages of every fp32 phase (the crowd generator; the fp32 gradient-penalty chain of the fp16
    # configuration) on fp32 tensors in the blocked layout: csrc/blocked16_k4s2.hip, dtype 0 (SRGAN_NO_BLOCKED_F32=1: the NCHW kernels)
    settings.blocked_fp32 = not os.environ.get('SRGAN_NO_BLOCKED_F32')
    settings.step_graph = bool(args.step_graph)
    if args.step_graph and dp is not None:
        settings.step_graph_collectives = 'abi'     # the opt-in: exchanges through the C ABI's own communicator, capturable
    if workload is not None:
        for key, value in workload['settings'].items():
            setattr(settings, key, value)
        if os.environ.get('SRGAN_NO_STORAGE16'):            # A / B: fp32 tensors, operands rounded per fragment (rounds 2-5)
            settings.storage_dtype = None
    if args.grad_wire:
        settings.gradient_wire_dtype = args.grad_wire
    if args.exchange_form:
        settings.gradient_exchange_form = args.exchange_form
    if workload is None:
        experiment = CrowdExperiment(settings)
    else:
        if workload['application'] == 'age':
            import srgan_amd.age.srgan as age
            age.model_architecture = workload['architecture']       # the reference's module-level switch (age/srgan.py:14)
            experiment = age.AgeExperiment(settings)
        else:
            from srgan_amd.driving.srgan import DrivingExperiment
            experiment = DrivingExperiment(settings)
        experiment.image_size = workload['image_size']
    experiment.dp = dp
    seed_all(0)
    experiment.dataset_setup()
    experiment.model_setup()
    scale = gp_scale(args)
    if scale != 1.0:
        with torch.no_grad():
            for module in experiment.D.modules():
                if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)):
                    module.weight.mul_(scale)
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    experiment.dnn_summary_writer = SummaryWriter(summary_period=10 ** 9)    # no host syncs inside timed steps
    experiment.gan_summary_writer = SummaryWriter(summary_period=10 ** 9)
    if dp is not None and dp.active:
        for module in (experiment.D, experiment.DNN, experiment.G):
            dp.broadcast_parameters(module._srgan_arena)
    return experiment


def side_streams(args):
    # (--single-device puts several RANKS on one GPU: a functional check of the multi-rank path on a one-GPU box.  Each
    # process's compute streams take hardware queues of the same device; two processes x three streams oversubscribe them
    # and the queue scheduler time-slices whole processes -- measured 51 s per step against 0.57 s on one stream each --
    # so that mode runs single-stream.  One process per GPU, the real layout, is not affected.)
    return not args.single_stream and not (args.single_device and args.gpus > 1)


def gp_scale(args):
    if args.gp_scale is not None:
        return args.gp_scale
    workload = WORKLOADS[args.workload]
    return workload['gp_scale'] if workload is not None else GP_SCALE.get(args.image_size, 1.27)


def restore_weights(arena, saved):
    """arena.data <- saved, and the 16-bit shadows of those weights (blocked16) behind it on the same stream."""
    arena.data.copy_(saved)
    if arena.shadows:
        from srgan_amd import blocked16
        blocked16.refresh(arena)


def one_step(experiment, labeled, unlabeled, step, eager=False):
    batch = next(labeled)
    x, labels = (batch[0], (batch[1], batch[2])) if len(batch) == 3 else batch
    u = next(unlabeled)[0]
    if eager:                                               # the event-bracketed step of the roofline leg
        experiment.dnn_training_step(x, labels, step + 1)
        experiment.gan_training_step(x, labels, u, step + 1)
    else:                                                   # step + 1 with a huge summary period: no host sync
        experiment.training_iteration(x, labels, u, step + 1)


STREAM_SETTINGS = ('overlap_dnn_step', 'wgrad_stream', 'overlap_generator_forwards', 'overlap_gradient_penalty')
# Round 5: on a stream with a workspace every K split finishes in a fixed order (csrc/split_finish.h), so the five losses an
# iteration computes before the discriminator's update are the same bits on any schedule and the sixth differs by the
# rounding of the parameter-gradient sums (still fp32 atomics): 1e-7 measured -- the limit is 1e-5.  With the round-4
# atomics (SRGAN_ATOMIC_SPLIT=1) a ReLU mask may flip between two runs of ONE schedule: 1e-4 there.
SCHEDULE_CHECK_LIMIT = 1e-5
SCHEDULE_CHECK_LIMIT_ATOMIC_SPLIT = 1e-4
# 16-bit storage (configs[1] / [4]): the two schedules add the penalty chain's share of D's gradients in a different association
# (its own buffer, added after the join), so D's updated fp32 weights differ in the last bit here and there -- and a last-bit
# difference flips the ROUNDING of that weight's bf16 / fp16 shadow now and then (2^-16 of the weights, each then off by 2^-8 / 2^-11
# relative): the generator loss, computed behind D's update, moves by 1e-5 .. 1e-4 (measured 2.2e-5 on age-vgg-bf16 after 8 steps;
# each schedule against its own repetition: 0).  The first five losses stay bit-identical.
SCHEDULE_CHECK_LIMIT_STORAGE16 = 2e-4


def schedule_check(experiment, labeled, unlabeled, step):
    """The schedule of the timed region (four streams, or their captured graph) against the single-stream eager schedule:
    ONE iteration each from the same weights, Adam state, batch and random draws; returns the largest relative difference
    of the six losses and of the updated weights.  Kernels and arithmetic are identical, so the two differ only by the
    summation order of fp32 atomics (1e-6) unless a chain read something another chain had not finished writing."""
    import itertools
    modules = (experiment.D, experiment.DNN, experiment.G)
    optimizers = (experiment.d_optimizer, experiment.dnn_optimizer, experiment.g_optimizer)
    experiment.join_dnn_stream()
    torch.cuda.synchronize()
    saved = [(m._srgan_arena.data.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_count)
             for m, o in zip(modules, optimizers)]
    batch, unlabeled_batch = next(labeled), next(unlabeled)
    generator = torch.Generator().manual_seed(1234)
    local = batch[0].shape[0]
    draws = {'z_d': torch.randn(local, experiment.G.input_size, generator=generator),
             'z_g': torch.randn(local, experiment.G.input_size, generator=generator),
             'alpha': torch.rand(local, generator=generator)}
    names = ('dnn_loss', 'labeled_loss', 'unlabeled_loss', 'fake_loss', 'gradient_penalty', 'generator_loss')
    partial = ('dnn_loss', 'labeled_loss', 'gradient_penalty')          # per-rank partial sums under data parallelism

    def run(timed_schedule):
        for (m, o), (data, exp_avg, exp_avg_sq, count) in zip(zip(modules, optimizers), saved):
            restore_weights(m._srgan_arena, data)
            o.exp_avg.copy_(exp_avg)
            o.exp_avg_sq.copy_(exp_avg_sq)
            o.step_count = count
            if o.device_state is not None:
                o.device_state[0] = count
        torch.cuda.synchronize()
        experiment.injected_draws = {k: v.clone() for k, v in draws.items()}
        one_step(experiment, itertools.repeat(batch), itertools.repeat(unlabeled_batch), step, eager=not timed_schedule)
        experiment.join_dnn_stream()
        torch.cuda.synchronize()
        losses = {name: experiment.loss_value(experiment.last_losses[name], partial=name in partial) for name in names}
        weights = [m._srgan_arena.data.clone() for m in modules]
        return losses, weights

    def difference(a, b):
        return max(abs(a[name] - b[name]) / max(abs(b[name]), 1e-12) for name in names)

    timed_losses, timed_weights = run(True)
    timed_again, _ = run(True)
    flags = {name: getattr(experiment.settings, name, False) for name in STREAM_SETTINGS}
    for name in STREAM_SETTINGS:
        setattr(experiment.settings, name, False)
    try:
        single_losses, single_weights = run(False)
        again_losses, _ = run(False)            # the same schedule twice: what the order of the fp32 atomics alone moves
    finally:
        for name, value in flags.items():
            setattr(experiment.settings, name, value)
    from srgan_amd import _lib
    ordered = bool(_lib.library().srgan_split_is_ordered(_lib.stream_handle()))
    worst = difference(timed_losses, single_losses)
    single_floor = difference(again_losses, single_losses)
    timed_floor = difference(timed_again, timed_losses)
    weight_difference = max(float((a - b).abs().max()) for a, b in zip(timed_weights, single_weights))
    for (m, o), (data, exp_avg, exp_avg_sq, count) in zip(zip(modules, optimizers), saved):     # back to the timed state
        restore_weights(m._srgan_arena, data)
        o.exp_avg.copy_(exp_avg)
        o.exp_avg_sq.copy_(exp_avg_sq)
        o.step_count = count
        if o.device_state is not None:
            o.device_state[0] = count
    torch.cuda.synchronize()
    # The limit is FIXED against the single-stream run: 1e-4, or four times what the SINGLE-STREAM schedule differs from its
    # own repetition by.  The multi-stream schedule's own repetition difference is reported (`timed_schedule_twice`) but never
    # widens the limit: a cross-chain race makes exactly that number large, so a limit scaled by it could not fail on the
    # defect the check exists to catch.
    return {'max_relative_loss_difference': worst, 'max_weight_difference': weight_difference,
            'single_stream_twice': single_floor, 'timed_schedule_twice': timed_floor,
            # (several ranks: the two schedules cut the gradient arenas into different buckets, so RCCL's ring adds an element's
            # rank contributions in another order -- the round-4 limit there; never run on more than one rank so far)
            'limit': max(SCHEDULE_CHECK_LIMIT if ordered and not (experiment.dp is not None and experiment.dp.world_size > 1)
                         else SCHEDULE_CHECK_LIMIT_ATOMIC_SPLIT,
                         SCHEDULE_CHECK_LIMIT_STORAGE16 if getattr(experiment.settings, 'storage_dtype', None) else 0.0,
                         4.0 * single_floor),
            'split_k_finish': 'fixed order through the workspace (no atomics on data)' if ordered else 'fp32 atomics',
            'what': 'one iteration on the timed schedule vs the same iteration on ONE stream (eager), from the same weights, '
                    'Adam state, batch and draws, after the timed region; *_twice = a schedule against its own repetition',
            'losses_timed_schedule': timed_losses, 'losses_single_stream': single_losses}


def pmc_traffic(args):
    """(HBM bytes of the contraction kernels of ONE step -- the caller divides by its own bracketed launches --, provenance)
    from the committed rocprofv3 PMC passes -- FETCH_SIZE x 2 +
    WRITE_SIZE per the guide's gfx950 correction, see profiles/README.md -- or (None, reason) unless the file was
    measured at THIS image size and batch on THESE kernel sources.  PMC collection needs rocprofv3 around the process, so
    it cannot be measured from inside this script; a constant from another configuration is not a measurement."""
    from srgan_amd import _build
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    switched = [name for name in ('SRGAN_NO_STORAGE16', 'SRGAN_NO_BLOCKED_F32', 'SRGAN_ATOMIC_SPLIT') if os.environ.get(name)]
    if switched:
        return None, f'the PMC passes ran on the default kernels, this run has {", ".join(switched)} set'
    try:
        entries = json.load(open(path))['entries']
    except (OSError, KeyError, ValueError):
        return None, 'profiles/pmc_traffic.json absent'
    size = args.image_size if isinstance(args.image_size, int) else args.image_size[0]
    for entry in entries:
        if (entry.get('workload', 'crowd') == args.workload and entry.get('image_size') == size and
                entry.get('batch_per_gpu') == args.batch_per_gpu and entry.get('kernel_source_id') == _build.source_id()):
            return entry['hbm_bytes_per_step'], entry.get('source', path)
    return None, (f'no PMC entry for workload {args.workload}, image size {size}, batch {args.batch_per_gpu}, kernel sources '
                  f'{_build.source_id()}')


def hbm_kernel_rates(experiment):
    """The HBM-bound pieces of the step, timed live with events on the launch stream and priced on their
    ALGORITHMIC bytes: the generator's Adam update (16 B read + 12 B written per parameter) and the
    one-pass batch-norm backward on a dense-block-3 sized tensor (g and x read, gx written: 12 B per element)."""
    import srgan_amd  # noqa: F401
    from srgan_amd import _lib
    lib = _lib.library()
    stream = _lib.stream_handle()

    def timed(fn, reps):
        fn()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn()
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) / reps * 1e-3
    optimizer = experiment.g_optimizer
    params = optimizer.arena.numel
    adam_s = timed(optimizer.step, 3)
    n, c, hw = 16, 1024, 32 * 32
    g, x, gx = (torch.randn(n, c, hw, device='cuda') for _ in range(3))
    mean, inv, gamma, beta = (torch.rand(c, device='cuda') + 0.5 for _ in range(4))
    grads = torch.zeros(2, c, device='cuda')
    bn_s = timed(lambda: lib.srgan_bn_act_bwd(g.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), 1, gx.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                             n, c, hw, 0, 0, 0, 0, 0, stream), 20)
    # the fused data gradient of a dense layer's bottleneck convolution (K = 128) with its batch-norm backward epilogue,
    # accumulating into the block's gradient buffer: HBM-bound by construction -- per pixel 4 * (128 + 3 * C_in) bytes (the
    # gradient tile in; x for the mask, the gradient buffer in and out) for 2 * 128 * C_in FLOP (dense block 3, layer 24)
    k, cin, total = 128, 1024, 1056
    buffer = torch.randn(n, total, hw, device='cuda')
    gbuf = torch.zeros(n, total, hw, device='cuda')
    gy = torch.randn(n, k, hw, device='cuda')
    weight = torch.randn(k, cin, device='cuda') / cin ** 0.5
    bn = _lib.BnRelu(mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    desc = _lib.ConvDesc(n, cin, 32, 32, k, 1, 1, 1, 1, 0, 0, 32, 32, total * hw, 0)
    dg_s = timed(lambda: _lib.check(lib.srgan_conv2d_bwd_data_bnrelu(desc, gy.data_ptr(), weight.data_ptr(), bn, buffer.data_ptr(),
                                                                     gbuf.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                                     1, stream), 'srgan_conv2d_bwd_data_bnrelu'), 20)
    dg_bytes = 4.0 * (k + 3 * cin) * n * hw
    # the per-example squared norm of the gradient penalty (reference srgan.py:371: gradients.view(B, -1).norm(dim=1)) over
    # C * H * W = 3 * S * S elements per example: 4 * C * H * W bytes per image and pass
    size = experiment.settings.image_patch_size
    per_example = 3 * size * size
    gradients = torch.randn(n, per_example, device='cuda')
    norms = torch.zeros(n, device='cuda')
    gp_s = timed(lambda: _lib.check(lib.srgan_chan_reduce(gradients.data_ptr(), gradients.data_ptr(), None, None, norms.data_ptr(),
                                                          1, n, per_example, 0, stream), 'srgan_chan_reduce'), 50)
    achievable = 6300.0                              # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s of the data sheet
    rates = {'peak_GBps': 8000.0, 'achievable_GBps': achievable,
             'adam': {'achieved_GBps': 28.0 * params / adam_s / 1e9, 'bytes_per_parameter': 28, 'parameters': params},
             'batch_norm_backward': {'achieved_GBps': 12.0 * n * c * hw / bn_s / 1e9, 'bytes_per_element': 12,
                                     'shape': [n, c, 32, 32]},
             'bottleneck_data_gradient_with_epilogue': {
                 'achieved_GBps': dg_bytes / dg_s / 1e9, 'bytes_per_pixel': 4 * (k + 3 * cin), 'shape': [n, cin, 32, 32],
                 'achieved_TFLOPs': 2.0 * k * cin * n * hw / dg_s / 1e12,
                 'note': 'both rooflines are close at this shape (21 FLOP/B): the kernel alternates matrix and epilogue '
                         'phases, see DESIGN.md'},
             'gradient_penalty_row_norm': {'achieved_GBps': 4.0 * n * per_example / gp_s / 1e9, 'bytes_per_image': 4 * per_example,
                                           'shape': [n, per_example], 'kernel': 'srgan::chan_reduce_rows_ordered_kernel '
                                           '(wave64 shuffle reduction; workgroup partials added in a fixed order by the '
                                           'row\'s last workgroup: one launch, no fp32 atomics)'}}
    for entry in rates.values():
        if isinstance(entry, dict):
            entry['fraction_of_achievable'] = entry['achieved_GBps'] / achievable
    return rates


def usable_cores():
    """Host cores this process may actually use: the CPU affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs under a 16-CPU quota; oversubscribing it stalls the oracle for hours)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(image_size, batch=1, timed=3, limit_seconds=600):
    """Runs ``cpu_baseline_child`` in a child process under a hard time limit (it never touches the GPU)."""
    import subprocess
    limit_seconds = max(limit_seconds, 150 * batch * (1 + timed) // 4)
    command = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', '--image-size', str(image_size),
               '--cpu-baseline-batch', str(batch), '--cpu-baseline-timed', str(timed)]
    try:
        output = subprocess.run(command, capture_output=True, text=True, timeout=limit_seconds).stdout
        return json.loads(output.strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, ValueError, IndexError) as error:
        return {'value': None, 'unit': 'images/s', 'cores': usable_cores(), 'kind': 'port', 'cpu': cpu_model(),
                'sample': f'not measured: {type(error).__name__} (limit {limit_seconds} s)'}


def secondary_line(workload, steps, warmup=5, limit_seconds=600):
    """One of the other BASELINE.json configurations measured by THIS script in a child process (its own timed region, schedule
    check and single-stream roofline step), reduced to the figures the headline line carries as `secondary`."""
    import subprocess
    command = [sys.executable, os.path.abspath(__file__), '--workload', workload, '--steps', str(steps), '--warmup', str(warmup),
               '--no-cpu-baseline', '--no-secondary']
    completed = None
    try:
        completed = subprocess.run(command, capture_output=True, text=True, timeout=limit_seconds)
        line = json.loads([text for text in completed.stdout.splitlines() if text.startswith('{')][-1])
    except (subprocess.TimeoutExpired, ValueError, IndexError) as error:
        return {'value': None, 'unit': 'images/s', 'error': f'{type(error).__name__}', 'command': ' '.join(command[1:]),
                'returncode': getattr(completed, 'returncode', None), 'stderr_tail': (getattr(completed, 'stderr', '') or '')[-1500:]}
    roofline = line.get('roofline') or {}
    check = line['config'].get('schedule_check')
    return {'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'],
            'steps': line['steps'], 'warmup': line['warmup'], 'dtype': line['dtype'], 'n_gpus': line['n_gpus'],
            'workload': line['config']['workload'], 'global_batch': line['config']['global_batch'],
            'gradient_penalty_last': line['config'].get('gradient_penalty_last'),
            'schedule_check_within_limit': check.get('within_limit') if isinstance(check, dict) else None,
            'roofline': {key: roofline.get(key) for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                            'launches', 'kernel_ms_per_step', 'executed_gflop_per_step',
                                                            'algorithmic_bytes_per_launch', 'fp32_part')},
            'command': 'python bench.py ' + ' '.join(command[2:])}


def committed_cpu_baseline(image_size, batch, cpu):
    """The oracle timed ONCE at the GPU leg's own batch (two minutes per iteration and 90 GiB at 512 x 512 / 16: too long
    for every default run), as committed under profiles/ -- carried next to the live sample so that the line holds the
    like-for-like figure too.  None unless a committed measurement of this image size and batch exists; `same_cpu_model`
    says whether it was taken on the CPU model this run sees."""
    path = os.path.join(ROOT, 'profiles', f'cpu_baseline_{image_size}x{image_size}_batch{batch}.json')
    try:
        entry = json.load(open(path))
    except (OSError, ValueError):
        return None
    return {'value': entry['value'], 'unit': entry['unit'], 'cores': entry['cores'], 'kind': entry['kind'], 'batch': entry['batch'],
            'cpu': entry.get('cpu'), 'same_cpu_model': entry.get('cpu') == cpu,
            'seconds_per_iteration': entry.get('seconds_per_iteration'),
            'provenance': f'profiles/{os.path.basename(path)} (python bench.py --cpu-baseline-child --cpu-baseline-batch {batch}, '
                          f'measured on a GPU box of this pool in {entry.get("measured", "round 4")}; not re-timed in this run)'}


def cpu_baseline_child(image_size, batch=1, warmup=1, timed=3):
    """The oracle's full iteration (reference srgan.py:104-118) on the host cores: ``warmup`` + ``timed`` iterations of
    the same image shape at ``batch`` images.  The per-image rate FALLS with the batch on the GPU box's 16
    host cores (profiles/r04_cpu_baseline_batch1.json: 0.297 images/s; r04_cpu_baseline_batch16.json: 0.133 images/s, 117-125 s
    per iteration), so the batch-1 sample of rounds 1-3 flattered the CPU by 2.2x against the GPU leg's batch of 16; the
    default run times batch 4 (1 warm-up + 2 timed iterations, the largest that fits a few minutes) and says so in
    ``sample``; ``--cpu-baseline-batch 16`` is the like-for-like measurement."""
    from types import SimpleNamespace
    from oracle import functional as OF, models as OM
    from oracle.experiment import OracleExperiment
    cores = usable_cores()
    torch.set_num_threads(cores)
    settings = SimpleNamespace(batch_size=batch, learning_rate=1e-4, weight_decay=0, labeled_loss_multiplier=1.0,
                               matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
                               srgan_loss_multiplier=1.0, gradient_penalty_multiplier=1e2, mean_offset=0,
                               labeled_loss_order=2, generator_training_step_period=1, normalize_feature_norm=False,
                               contrasting_distance_function=OF.abs_plus_one_sqrt_mean_neg,
                               matching_distance_function=OF.abs_mean, map_multiplier=1e-3)
    OF.seed_all(0)
    G = OM.DCGANGenerator(image_size=image_size)
    D, DNN = OM.KnnDenseNetCat(image_size=image_size), OM.KnnDenseNetCat(image_size=image_size)
    with torch.no_grad():                        # the same active-gradient-penalty weights as the GPU legs
        for module in D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                module.weight.mul_(GP_SCALE.get(image_size, 1.27))
    oracle = OracleExperiment(settings, D, DNN, G,
                              labeled_loss_function=lambda p, y, order: OF.crowd_labeled_loss(p, y, order, 1e-3))
    generator = torch.Generator().manual_seed(0)
    x = torch.rand(batch, 3, image_size, image_size, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, image_size, image_size, generator=generator) * 2 - 1
    heads = (torch.rand(batch, image_size, image_size, generator=generator) < 0.002).float()
    knn = torch.rand(batch, image_size, image_size, generator=generator)
    seconds = []
    for iteration in range(warmup + timed):
        start = time.perf_counter()
        oracle.dnn_training_step(x, (heads, knn))
        oracle.gan_training_step(x, (heads, knn), u, iteration)
        seconds.append(time.perf_counter() - start)
    elapsed = sum(seconds[warmup:])
    return {'value': batch * timed / elapsed, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu': cpu_model(),
            'batch': batch, 'seconds_per_iteration': [round(t, 2) for t in seconds[warmup:]],
            'sample': f'BATCH {batch}, i.e. a per-image rate at batch {batch} (the GPU leg runs batch 16 per GPU): '
                      f'{warmup} warm-up + {timed} timed iterations (dnn_training_step + gan_training_step) of the '
                      f'PyTorch-CPU fp32 oracle, crowd {image_size}x{image_size}: '
                      + ' / '.join(f'{t:.2f}' for t in seconds[warmup:]) + f' s (warm-up {seconds[0]:.2f} s)'}


def ensure_library():
    """libsrgan_hip.so normally arrives prebuilt with the working tree; when it is missing or was built from other
    kernel sources (compared by content: ``_build.is_current``), local rank 0 compiles it (hipcc, about a minute) and
    the other ranks wait for the finished file (it is renamed into place).  No GPU call is made here, and there is
    still no fallback: without the library nothing runs."""
    from srgan_amd import _build
    if _build.is_current():
        return
    if int(os.environ.get('LOCAL_RANK', '0')) == 0:
        _build.build()
        return
    deadline = time.time() + 900
    while not _build.is_current() and time.time() < deadline:
        time.sleep(2)


def launch_ranks(args):
    """``python bench.py --gpus N`` from a plain shell: start the N ranks as a child ``torch.distributed.run`` (one process
    per GPU, rendezvous on 127.0.0.1) and relay its output.  This process has made no GPU call (importing torch and
    compiling with hipcc do not initialise the device), and it never replaces itself with another program."""
    import socket
    import subprocess
    port = args.master_port
    if port is None:
        with socket.socket() as probe:
            probe.bind(('127.0.0.1', 0))
            port = probe.getsockname()[1]
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    environment = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SRGAN_BENCH_SELF_LAUNCHED='1')
    child = subprocess.run(command, env=environment)
    return child.returncode


def main():
    args = parse()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args.image_size, batch=args.cpu_baseline_batch, timed=args.cpu_baseline_timed)))
        return 0
    ensure_library()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but torch.distributed.run started {world} ranks')
    local_rank = 0 if args.single_device else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dp = None
    if world > 1 or args.force_dp:
        import srgan_amd  # noqa: F401
        from srgan_amd.parallel import DataParallel
        if world == 1:                       # a world of one needs no launcher: rendezvous with ourselves on 127.0.0.1
            import socket
            with socket.socket() as probe:
                probe.bind(('127.0.0.1', 0))
                free_port = probe.getsockname()[1]
            for key, value in (('RANK', '0'), ('WORLD_SIZE', '1'), ('LOCAL_RANK', '0'), ('MASTER_ADDR', '127.0.0.1'),
                               ('MASTER_PORT', str(args.master_port or free_port))):
                os.environ.setdefault(key, value)
        dp = DataParallel.from_environment(args.backend, force=args.force_dp)
        if args.force_dp:
            dp.broadcast_object({'hello': args.backend})     # the object broadcast of Experiment.train(), through the backend
    rank = dp.rank if dp else 0
    experiment = build_experiment(args, dp)
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)

    def fence():
        experiment.join_dnn_stream()       # also applies optimizer updates still waiting for their gradient exchange
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()

    for step in range(args.warmup):
        one_step(experiment, labeled, unlabeled, step)
    if args.step_graph:
        # graph capture is set-up, not a step: make sure the timed region only replays
        for extra in range(4):
            if getattr(experiment, '_captured_iteration', None) is not None and experiment._captured_iteration.replays:
                break
            one_step(experiment, labeled, unlabeled, step=0)
    fence()
    # The batches are noise and nothing is learned: left alone, the penalty-active discriminator drifts (gradient penalty 1e2
    # after 13 iterations, 4e4 after 25) and a long --steps would end in non-finite losses.  Every RESTORE_PERIOD steps the
    # three networks' weights go back to their post-warm-up values (device-to-device copies INSIDE the timed region: 0.76 GB
    # per 16 steps at 512 x 512, < 0.02 ms per step); every step still runs its full forward / backward / Adam arithmetic.
    snapshot = [(module._srgan_arena, module._srgan_arena.data.clone())
                for module in (experiment.D, experiment.DNN, experiment.G)]
    torch.cuda.synchronize()
    start = time.perf_counter()
    for step in range(args.steps):
        one_step(experiment, labeled, unlabeled, args.warmup + step)
        if step % RESTORE_PERIOD == RESTORE_PERIOD - 1 and step + 1 < args.steps:
            experiment.join_dnn_stream()
            for live, saved in snapshot:
                restore_weights(live, saved)
    fence()
    elapsed = time.perf_counter() - start
    # What the HOST needs to enqueue one iteration (eager: the Python tape and ~4000 launches; replay: input copies + one graph
    # launch): one more iteration issued into an idle device, timed until the last launch call returns -- no queue
    # back-pressure in the number, which is what matters when eight ranks share a 16-CPU quota.  Outside the timed region.
    host_seconds = float('nan')
    if not os.environ.get('SRGAN_BENCH_NO_HOST_STEP'):      # (the PMC passes count exactly the timed steps: scratch/pmc_summarise.py)
        enqueue_start = time.perf_counter()
        one_step(experiment, labeled, unlabeled, args.warmup + args.steps)
        host_seconds = time.perf_counter() - enqueue_start
        fence()
    for live, saved in snapshot:
        restore_weights(live, saved)
    del snapshot
    per_rank_ms = None
    if dp is not None:
        own = torch.zeros(world, dtype=torch.float64, device=dp._scalar_device())
        own[rank] = 1e3 * elapsed / args.steps
        per_rank_ms = [round(v, 3) for v in dp.all_reduce_sum_(own).tolist()]
        elapsed = dp.all_reduce_max_float(elapsed)
    global_batch = experiment.settings.batch_size
    images_per_second = global_batch * args.steps / elapsed
    # after the timed region: the penalty really was active, and nothing diverged
    penalty = experiment.loss_value(experiment.last_losses['gradient_penalty'], partial=True)
    finite = all(bool(torch.isfinite(v.data).all()) for v in experiment.last_losses.values() if v is not None)
    if gp_scale(args) != 1.0 and not penalty > 0.0:
        raise SystemExit(f'gradient penalty inactive ({penalty}) at GP scale {gp_scale(args)}: the benchmark would time zeros')
    if not finite:
        raise SystemExit('non-finite losses in the last timed step')
    check = None
    if side_streams(args) or args.step_graph:
        # ONE comparison against a fixed limit, no second chances: a difference above it is a failure of the schedule.
        check = schedule_check(experiment, labeled, unlabeled, args.warmup + args.steps)
        excess = check['max_relative_loss_difference'] / check['limit']
        if dp is not None:                  # one decision for all ranks: a rank that left alone would hang the others
            excess = dp.all_reduce_max_float(excess if excess == excess else float('inf'))
            check['worst_rank_excess_over_limit'] = excess
        check['within_limit'] = bool(excess <= 1.0)
        if not check['within_limit']:
            raise SystemExit(f'the timed schedule and the single-stream schedule disagree: {json.dumps(check)}')

    result = {
        'metric': 'SRGAN train images/sec (G+D step)', 'value': images_per_second, 'unit': 'images/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': WORKLOADS[args.workload]['dtype'] if WORKLOADS[args.workload] else 'f32', 'data': 'synthetic',
        'config': {'workload': (WORKLOADS[args.workload]['name'] + f', batch {args.batch_per_gpu}/GPU, dnn_training_step + '
                                'gan_training_step per step') if WORKLOADS[args.workload] else
                               f'crowd SRGAN (KnnDenseNetCat D/DNN + DCGenerator) {args.image_size}x{args.image_size}, '
                               f'batch {args.batch_per_gpu}/GPU, dnn_training_step + gan_training_step per step',
                   'global_batch': global_batch, 'image_size': args.image_size,
                   'schedule': 'reference' if args.reference_schedule else 'shared-forwards',
                   'parallelism': f'dp{world}' + (' (data-parallel exchanges forced on)' if args.force_dp else ''), 'random_init_weights': True,
                   'gradient_penalty': 'active' if penalty > 0.0 else 'inactive', 'gradient_penalty_last': penalty,
                   'discriminator_weight_scale': gp_scale(args),
                   'weights_restored_every': f'{RESTORE_PERIOD} timed steps (to the post-warm-up weights; the batches are noise, '
                                             'so an unbounded run diverges; every step runs its full arithmetic)'},
    }
    captured = getattr(experiment, '_captured_iteration', None)
    result['config']['streams'] = ('timed region: four streams = the four hardware queues of the HIP runtime -- main chain (stacked discriminator '
                                   'pass, generator step), gradient-penalty chain, DNN step, D(unlabeled) of the generator step'
                                   + (' -- captured as the parallel branches of ONE HIP graph' if args.step_graph else '') +
                                   '; roofline step: single stream' if side_streams(args) else 'single stream')
    if side_streams(args) and dp is not None and dp.active:
        result['config']['streams'] = ('timed region: THREE compute streams under data parallelism (main chain, gradient-penalty chain, DNN '
                                       'step) so that RCCL\'s communication stream has the fourth hardware queue to itself')
    if args.step_graph and dp is not None and dp.active:
        result['config']['streams'] = ('ONE compute stream + the communication stream, captured as one HIP graph (the HIP runtime crashes in '
                                       'hipStreamEndCapture when a capture holds the compute side streams and the communication stream '
                                       'together; sr-gan_amd/srgan.py:_exchanges_are_capturable)')
    result['config']['launch'] = (f'HIP graph replay ({captured.replays} replayed, {captured.eager_iterations} eager iterations)'
                                  if captured is not None else 'eager (Python tape enqueues every kernel)')
    result['config']['host_ms_per_step'] = round(1e3 * host_seconds, 3) if host_seconds == host_seconds else None
    result['config']['schedule_check'] = check if check is not None else 'not applicable: the timed region ran on one stream, eagerly'
    if dp is not None:
        import torch.distributed as dist
        if dp.abi is not None:
            seen = f'RCCL communicator of the C ABI (srgan_comm_init) saw {dp.abi.world_size} ranks'
        else:
            seen = (f'{dp.device_backend or dp.host_backend} ({"RCCL" if dp.device_backend == "nccl" else "host"}) saw '
                    f'{dist.get_world_size()} ranks')
        result['config']['collective_world'] = seen
        result['config']['collective_transport'] = dp.transport
        result['config']['control_plane'] = f'torch.distributed ({dp.host_backend or dp.device_backend})' 
        result['config']['per_rank_ms_per_step'] = per_rank_ms
        result['config']['gradient_wire'] = getattr(experiment.settings, 'gradient_wire_dtype', None) or 'f32'
        result['config']['gradient_exchange_form'] = getattr(experiment.settings, 'gradient_exchange_form', None) or 'all_reduce'
        result['config']['gradient_exchange'] = ('blocking' if args.no_overlap_exchange else
                                                 'asynchronous, overlapped with backward / next phase') + f' ({args.backend})'
    gflop = ALGORITHMIC_GFLOP_PER_IMAGE.get(args.image_size) if args.workload == 'crowd' else None
    if gflop:
        result['step_tflops_as_written_schedule'] = images_per_second * gflop / 1e3
        result['step_frac_as_written_schedule'] = result['step_tflops_as_written_schedule'] / (FP32_MFMA_PEAK_TFLOPS * world)

    if rank == 0 and world == 1 and not args.no_roofline:
        import ctypes
        from srgan_amd import _lib
        lib = _lib.library()
        # per-kernel attribution needs one kernel at a time: this extra step runs on ONE stream
        experiment.join_dnn_stream()
        torch.cuda.synchronize()
        for name in STREAM_SETTINGS:
            setattr(experiment.settings, name, False)
        lib.srgan_profile_begin()
        one_step(experiment, labeled, unlabeled, args.warmup + args.steps, eager=True)
        experiment.join_dnn_stream()
        kernel_ms, flops, mfma_flops, launches = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.srgan_profile_end(ctypes.byref(kernel_ms), ctypes.byref(flops), ctypes.byref(mfma_flops),
                                         ctypes.byref(launches)), 'srgan_profile_end')
        algorithmic_bytes = ctypes.c_double()
        _lib.check(lib.srgan_profile_bytes(ctypes.byref(algorithmic_bytes)), 'srgan_profile_bytes')
        torch.cuda.synchronize()
        if args.shape_report:
            size = lib.srgan_profile_report(None, 0)
            text = ctypes.create_string_buffer(size)
            lib.srgan_profile_report(text, size)
            with open(args.shape_report, 'w') as handle:
                handle.write('# M N K kind bm bn split akf bkf count ms bytes   (kind: 0 gg_direct 1 gg_mfma 2 conv3x3_lds '
                             '3 pointwise 4 conv3x3_wgrad 5 gg_rows 6 pointwise_wgrad 8 pointwise_ksplit 9 gg_dot 10 stem7x7_fwd 11 stem7x7_wgrad '
                             '12 stem7x7_bwd_data 13 pointwise_ring)\n')
                handle.write(text.value.decode())
        achieved = flops.value / (kernel_ms.value * 1e-3) / 1e12 if kernel_ms.value > 0 else 0.0
        traffic, traffic_source = pmc_traffic(args)
        if traffic is not None:                    # per launch like `achieved`: the counters' step total over THIS step's brackets
            traffic = traffic / max(launches.value, 1)
        step_seconds = elapsed / args.steps
        result['roofline'] = {
            'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
            'kernel': 'all contraction kernels of one step: srgan::pointwise_ring_kernel / pointwise_kernel / conv3x3_lds_kernel / conv3x3_wgrad_kernel / '
                      'pointwise_wgrad(_lds)_kernel / gg_mfma_kernel / gg_rows_kernel (every conv and linear pass)',
            'launches': launches.value, 'kernel_ms_per_step': kernel_ms.value,
            'executed_gflop_per_step': flops.value / 1e9, 'mfma_share_of_flops': mfma_flops.value / max(flops.value, 1.0),
            'avg_launch_us': 1e3 * kernel_ms.value / max(launches.value, 1),
            'algorithmic_bytes_per_step': algorithmic_bytes.value,
            'algorithmic_bytes_per_launch': algorithmic_bytes.value / max(launches.value, 1),
            'step_frac_executed': flops.value / step_seconds / 1e12 / FP32_MFMA_PEAK_TFLOPS,
        }
        if args.workload != 'crowd':
            # mixed precision: the bf16 / fp16 launches are priced against the dense low-precision peak, the fp32 remainder
            # (gradient-penalty chain of the fp16 mode, VALU kernels) against the fp32 one
            low_flops, low_ms = ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.srgan_profile_mixed(ctypes.byref(low_flops), ctypes.byref(low_ms)), 'srgan_profile_mixed')
            low = low_flops.value / (low_ms.value * 1e-3) / 1e12 if low_ms.value > 0 else 0.0
            rest_ms, rest_flops = kernel_ms.value - low_ms.value, flops.value - low_flops.value
            roofline = result['roofline']
            roofline.update({
                'achieved': low, 'peak': LOW_PRECISION_MFMA_PEAK_TFLOPS, 'frac': low / LOW_PRECISION_MFMA_PEAK_TFLOPS,
                'kernel': 'every contraction launched with bf16 / fp16 operands (v_mfma_f32_32x32x16_bf16 / _f16): the 16-bit data '
                          'path srgan::hconv3x3_kernel / hwgrad3x3_kernel / hgemm_kernel / hlinear_wgrad_kernel (blocked16*.hip) and the '
                          'fp32-storage kernels gg_mfma_kernel<..., PREC> / conv3x3_mixed_kernel where a network has no 16-bit path',
                'low_precision_kernel_ms_per_step': low_ms.value, 'low_precision_gflop_per_step': low_flops.value / 1e9,
                'fp32_part': {'kernel_ms_per_step': rest_ms, 'gflop_per_step': rest_flops / 1e9,
                              'achieved': rest_flops / (rest_ms * 1e-3) / 1e12 if rest_ms > 0 else 0.0,
                              'peak': FP32_MFMA_PEAK_TFLOPS},
                'step_frac_executed': None})
    if rank == 0 and world == 1 and not args.no_roofline and args.workload == 'crowd':
        result['hbm_kernels'] = hbm_kernel_rates(experiment)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'crowd':
        result['cpu_baseline'] = cpu_baseline(args.image_size, batch=args.cpu_baseline_batch, timed=args.cpu_baseline_timed)
        result['cpu_baseline']['like_for_like'] = committed_cpu_baseline(args.image_size, args.batch_per_gpu,
                                                                         result['cpu_baseline'].get('cpu'))
    if rank == 0 and world == 1 and dp is None and args.workload == 'crowd' and not args.no_secondary:
        # the two BASELINE.json configurations that name a precision (configs[1] bf16, configs[4] fp16): same script, child
        # processes, after this process has finished its own GPU work
        torch.cuda.synchronize()
        result['secondary'] = {'age_vgg_bf16': secondary_line('age-vgg-bf16', args.secondary_steps),
                               'driving_fp16': secondary_line('driving-fp16', args.secondary_steps)}
    if dp is not None:
        dp.barrier()
        experiment.close()                        # the C ABI's communicator, before the process group goes
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed when the process exits -- i.e. AFTER a Python print:
        # push it out first, so that the JSON line is the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
