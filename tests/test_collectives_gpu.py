"""The data-parallel exchanges on the device: RCCL through torch.distributed and through the C ABI's own entry
points at world size 1 (two ranks cannot share a device under RCCL), the bf16 wire format and the reduce-scatter form of the
gradient buckets, a data-parallel iteration captured and replayed as a HIP graph, the forced data-parallel bench lines.
(Two ranks on one GPU over gloo: test_parallel_gpu.py; world size 2 / 3 on the CPU: test_parallel_cpu.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def _bench_line(*arguments, environment=None):
    """bench.py as the driver starts it (a fresh process), small settings; returns rank 0's JSON line."""
    command = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--image-size', '64',
               '--batch-per-gpu', '2', '--no-cpu-baseline', '--no-roofline', '--no-secondary'] + list(arguments)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(environment or {}))
    done = subprocess.run(command, capture_output=True, text=True, timeout=900, env=env)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    return json.loads([line for line in done.stdout.splitlines() if line.startswith('{')][-1])


def test_bench_with_the_exchanges_forced_through_rccl_on_one_rank(pkg):
    """``bench.py --gpus 1 --force-dp --backend nccl``: an nccl (= RCCL) process group of one rank, every collective of the
    data-parallel path on it; the last step's gradient penalty must equal the plain run's."""
    plain = _bench_line()
    forced = _bench_line('--force-dp', '--backend', 'nccl')
    assert 'forced' in forced['config']['parallelism'] and 'nccl' in forced['config']['gradient_exchange']
    a, b = plain['config']['gradient_penalty_last'], forced['config']['gradient_penalty_last']
    # (the third training step on noise: rounding-level differences of the first two Adam updates have grown to ~5e-4)
    assert a > 0 and abs(a - b) <= 5e-3 * abs(a), (a, b)


def test_rccl_entry_points_of_the_abi_on_one_rank(pkg):
    """include/srgan_hip.h "collectives": a communicator of ONE rank from a unique id (two ranks cannot share a device under
    RCCL, one rank can); all-reduce, reduce-scatter, all-gather and broadcast in fp32 and bf16 on the caller's stream are the identity
    there, and they must leave exactly that; argument errors come back as -1 before RCCL is reached."""
    import ctypes
    from srgan_amd import _lib
    lib = _lib.library()
    assert lib.srgan_comm_available() == 1
    identifier = ctypes.create_string_buffer(128)
    _lib.check(lib.srgan_comm_unique_id(identifier), 'srgan_comm_unique_id')
    assert any(identifier.raw)
    comm = ctypes.c_void_p()
    _lib.check(lib.srgan_comm_init(ctypes.byref(comm), 1, 0, identifier.raw), 'srgan_comm_init')
    world = ctypes.c_int32()
    _lib.check(lib.srgan_comm_world_size(comm, ctypes.byref(world)), 'srgan_comm_world_size')
    assert world.value == 1
    stream = torch.cuda.current_stream().cuda_stream
    for dtype, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        source = torch.randn(100003, device='cuda').to(dtype)
        out = torch.zeros_like(source)
        _lib.check(lib.srgan_all_reduce_sum(comm, source.data_ptr(), out.data_ptr(), source.numel(), code, stream), 'all_reduce')
        assert torch.equal(out, source)
        in_place = source.clone()
        _lib.check(lib.srgan_all_reduce_sum(comm, in_place.data_ptr(), in_place.data_ptr(), source.numel(), code, stream), 'all_reduce')
        assert torch.equal(in_place, source)
        shard = torch.zeros_like(source)
        _lib.check(lib.srgan_reduce_scatter_sum(comm, source.data_ptr(), shard.data_ptr(), source.numel(), code, stream), 'reduce_scatter')
        gathered = torch.zeros_like(source)
        _lib.check(lib.srgan_all_gather(comm, shard.data_ptr(), gathered.data_ptr(), source.numel(), code, stream), 'all_gather')
        assert torch.equal(gathered, source)
        weights = source.clone()
        _lib.check(lib.srgan_broadcast(comm, weights.data_ptr(), weights.numel(), code, 0, stream), 'broadcast')
        assert torch.equal(weights, source)
    assert lib.srgan_broadcast(comm, source.data_ptr(), 4, 0, -1, stream) == _lib.EINVAL                       # negative root
    assert lib.srgan_all_reduce_sum(comm, source.data_ptr(), out.data_ptr(), 4, 7, stream) == _lib.EINVAL     # unknown dtype
    assert lib.srgan_all_reduce_sum(None, source.data_ptr(), out.data_ptr(), 4, 0, stream) == _lib.EINVAL
    torch.cuda.synchronize()
    _lib.check(lib.srgan_comm_destroy(comm), 'srgan_comm_destroy')


@pytest.mark.parametrize('streams', [False, True])
def test_data_parallel_step_through_the_abi_collectives_equals_the_plain_step(streams):
    """`SRGAN_ABI_COLLECTIVES=1`: the feature-sum all-reduce of the forward pass and every asynchronous gradient bucket go
    through srgan_all_reduce_sum on an RCCL communicator the ABI created (world size 1, exchanges forced on; the process
    group only carries the unique id, the broadcasts and the barrier) -- losses and updated weights equal the plain step."""
    import test_parallel_gpu as parallel_tests
    reference_result, reference_tensors = parallel_tests._step(None)
    (rank, result, tensors, launched), = parallel_tests._run_ranks(1, 'nccl', force=True, streams=streams, abi=True)
    assert launched['DNN'] and launched['D'] and launched['G'], launched
    buckets = sum(len(b) for name, runs in launched.items() if name != 'abi_collectives' for b in runs)
    assert launched['abi_collectives'] >= buckets + 3, launched          # the buckets + the forward feature sums
    for key, value in reference_result.items():
        assert abs(result[key] - value) <= 1e-4 * max(abs(value), 1e-6), (key, result[key], value)
    for key, value in reference_tensors.items():
        limit = 2.2e-4 + 1e-3 * float(np.abs(value).max())
        assert float(np.abs(tensors[key] - value).max()) <= limit, key
    import conftest
    conftest.PARITY_NOTES.append(f'data-parallel step with every exchange through the C ABI\'s RCCL entry points (world size 1, forced): '
                                 f'{launched["abi_collectives"]} collectives, {buckets} gradient buckets, losses equal the plain step '
                                 f'(side streams {"on" if streams else "off"})')


def _graph_under_dp_worker(port, queue, streams):
    import os
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.set_device(0)
    import srgan_amd  # noqa: F401
    from srgan_amd.parallel import DataParallel
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    from srgan_amd.utility import seed_all
    from test_steps_gpu import make_experiment, finish_setup, crowd_inputs
    dp = DataParallel.from_environment('nccl', force=True)
    size, batch, iterations = 64, 2, 5

    def run(step_graph):
        experiment = make_experiment(
            lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
            dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2, gradient_penalty_multiplier=1e2,
                 map_multiplier=1e-3, step_graph=step_graph, step_graph_warmup=1, steps_to_run=10 ** 9,
                 step_graph_collectives='abi',          # the opt-in: exchanges through the C ABI's communicator, capturable
                 overlap_dnn_step=streams, overlap_gradient_penalty=streams), crowd=True)
        experiment.dp = dp
        with torch.no_grad():
            for module in experiment.D.modules():
                if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    module.weight.mul_(1.27)
        finish_setup(experiment)
        for optimizer in (experiment.d_optimizer, experiment.g_optimizer, experiment.dnn_optimizer):
            optimizer.count_on_device()      # both runs through the device-counted Adam entry point
        for writer in (experiment.dnn_summary_writer, experiment.gan_summary_writer):
            writer.summary_period, writer.steps_to_run = 10 ** 9, 10 ** 9
        seed_all(5)
        generator = torch.Generator().manual_seed(11)
        losses = []
        for step in range(1, iterations + 1):
            x, labels, u = crowd_inputs(generator, batch, size)
            if step == 3:
                # an EAGER iteration between replays (what a summary step is) leaves the generator's update pending;
                # the next replay must settle it first.  The eager run does the same iteration the same way.
                experiment.dnn_training_step(x.cuda(), tuple(t.cuda() for t in labels), step)
                experiment.gan_training_step(x.cuda(), tuple(t.cuda() for t in labels), u.cuda(), step)
            else:
                experiment.training_iteration(x.cuda(), tuple(t.cuda() for t in labels), u.cuda(), step)
            losses.append({name: float(value.item()) for name, value in experiment.last_losses.items() if value is not None})
        experiment.finish_update()
        experiment.join_dnn_stream()
        torch.cuda.synchronize()
        captured = getattr(experiment, '_captured_iteration', None)
        return losses, {name: getattr(experiment, name)._srgan_arena.data.cpu().numpy() for name in ('D', 'DNN', 'G')}, \
            (captured.replays, captured.eager_iterations) if captured is not None else None

    eager = run(False)
    calls_before = dp.abi.calls if dp.abi is not None else 0
    replayed = run(True)
    queue.put((eager, replayed, dp.abi is not None, (dp.abi.calls if dp.abi is not None else 0) - calls_before))
    dp.barrier()
    if dp.abi is not None:
        dp.abi.close()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('streams', [False, True])
def test_graph_replay_under_data_parallelism_over_rccl(streams):
    """`settings.step_graph` with the data-parallel exchanges on (nccl = RCCL, world size 1, forced): the iteration -- feature-sum
    all-reduces, asynchronous gradient buckets, their waits, the three optimizer updates -- is captured ONCE as a HIP graph
    with the collectives as launches of the C ABI's RCCL entry points on the communicator's stream, and four replays leave
    the losses and weights of the eager data-parallel run."""
    import multiprocessing as mp
    import test_parallel_gpu as parallel_tests
    context = mp.get_context('spawn')
    queue = context.Queue()
    worker = context.Process(target=_graph_under_dp_worker, args=(parallel_tests._free_port(), queue, streams))
    worker.start()
    import queue as queue_module
    import time
    deadline = time.monotonic() + 600
    while True:                       # (a worker that died must not hold the suite for the whole time limit)
        try:
            result = queue.get(timeout=5)
            break
        except queue_module.Empty:
            assert worker.is_alive() or not queue.empty(), f'the rank died (exit code {worker.exitcode})'
            assert time.monotonic() < deadline, 'the rank is still running after ten minutes'
    (eager_losses, eager_weights, eager_captured), (losses, weights, captured), abi, calls_during_replays = result
    worker.join(timeout=120)
    assert worker.exitcode == 0
    assert eager_captured is None and captured == (3, 1), captured            # a warm-up iteration, an eager one in between (outside
    assert abi and calls_during_replays > 0                                   # the captured-iteration counter), three replays
    if not streams:
        # one compute stream in both runs: the replay executes the eager run's arithmetic -- losses and weights bit for bit
        assert eager_losses == losses, (eager_losses, losses)
        for name in ('D', 'DNN', 'G'):
            assert np.array_equal(eager_weights[name], weights[name]), (name, float(np.abs(eager_weights[name] - weights[name]).max()))
    else:
        # The eager run keeps its side streams, the captured one runs on one compute stream: the penalty chain's gradients of
        # D then arrive as ONE sum added to the stacked pass's (a + (b1 + b2 + ...)) instead of parameter by parameter
        # ((a + b1) + b2 ...): D's update differs at rounding level, which the first iteration shows only in generator_loss
        # (computed behind D's update) and later iterations everywhere, amplified by Adam's first updates.
        for step, (a, b) in enumerate(zip(eager_losses, losses)):
            for name in a:
                rtol = 0.0 if (step == 0 and name != 'generator_loss') else (5e-3 if step <= 1 else 3e-2)     # (1.0e-2 observed at step 3)
                assert np.isclose(a[name], b[name], rtol=rtol, atol=0.0 if rtol == 0.0 else 1e-6), f'step {step} {name}: {a[name]} vs {b[name]}'
        for name in ('D', 'DNN', 'G'):
            difference = np.abs(eager_weights[name] - weights[name])
            assert float(difference.max()) <= 2.2e-4 * 5 and float(difference.mean()) <= 0.5e-4, name
    assert losses[-1]['gradient_penalty'] > 0.0 and losses[-1] != losses[-2]
    import conftest
    conftest.PARITY_NOTES.append(f'HIP-graph replay of the data-parallel iteration (RCCL through the C ABI, world size 1, forced; side '
                                 f'streams {"on" if streams else "off"}): 4 replays equal the eager run')


def test_bf16_pack_and_unpack_kernels(pkg):
    """``srgan_pack_bf16`` / ``srgan_unpack_bf16`` (the bf16 gradient buckets): round to nearest even, bit-identical to
    torch's conversion, NaN and infinities kept, lengths that are no multiple of 8."""
    from srgan_amd import _lib
    lib = _lib.library()
    generator = torch.Generator().manual_seed(3)
    for n in (1, 7, 8, 1000003):
        values = torch.randn(n, generator=generator) * torch.logspace(-20, 20, n)
        values[0] = float('inf')
        if n > 4:
            values[1], values[2], values[3], values[4] = float('-inf'), float('nan'), 0.0, -0.0
        source = values.cuda()
        packed = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
        _lib.check(lib.srgan_pack_bf16(source.data_ptr(), packed.data_ptr(), n, _lib.stream_handle()), 'srgan_pack_bf16')
        expected = values.to(torch.bfloat16)
        got = packed.cpu()
        finite = ~torch.isnan(expected)
        assert torch.equal(got.view(torch.int16)[finite], expected.view(torch.int16)[finite]), n
        assert bool(torch.isnan(got[~finite]).all())
        back = torch.full((n,), 7.0, device='cuda')
        _lib.check(lib.srgan_unpack_bf16(packed.data_ptr(), back.data_ptr(), n, _lib.stream_handle()), 'srgan_unpack_bf16')
        assert torch.equal(back.cpu()[finite], expected.float()[finite]), n


def _exchange_worker(rank, world, port, backend, queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import srgan_amd  # noqa: F401
    from srgan_amd.parallel import DataParallel, GradientExchange
    torch.cuda.set_device(0)
    dp = DataParallel.from_environment(backend, force=True)
    generator = torch.Generator().manual_seed(50 + rank)
    source = torch.randn(3000004, generator=generator) * torch.logspace(-6, 2, 3000004)
    results = {}
    for wire, form in (('f32', 'all_reduce'), ('f32', 'reduce_scatter'), ('bf16', 'all_reduce'), ('bf16', 'reduce_scatter')):
        flat = source.cuda()
        exchange = GradientExchange(dp, flat, bucket_elements=1 << 20, min_bucket_elements=1 << 12, wire=wire, form=form)
        exchange.ready_from(2999000 // 4 * 4)        # (16-byte aligned offsets, as the arena's parameter offsets are)
        exchange.ready_from(1200004)
        exchange.finish().wait()
        torch.cuda.synchronize()
        results[f'{wire}/{form}'] = (flat.cpu().numpy(), list(exchange.launched))
    queue.put((rank, source.numpy(), results))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('backend,world', [('nccl', 1), ('gloo', 2), ('nccl', 2)])
def test_gradient_exchange_forms_on_the_device(pkg, backend, world):
    """The exchange on DEVICE buffers through the library's staging kernels: one rank over nccl (= RCCL; all-reduce,
    reduce-scatter and all-gather of a world of one are identities, so the fp32 forms return the input bit for bit and the
    bf16 forms its bf16 rounding), two ranks sharing this GPU over gloo, two ranks over nccl when the box has two GPUs."""
    import socket
    import torch.multiprocessing as mp
    if backend == 'nccl' and world > torch.cuda.device_count():
        pytest.skip('two ranks over nccl (RCCL) need two GPUs; this box has %d (the world-size-1 nccl case ran)'
                    % torch.cuda.device_count())
    with socket.socket() as probe:
        probe.bind(('127.0.0.1', 0))
        port = probe.getsockname()[1]
    context = mp.get_context('spawn')
    queue = context.Queue()
    workers = [context.Process(target=_exchange_worker, args=(rank, world, port, backend, queue)) for rank in range(world)]
    for worker in workers:
        worker.start()
    outputs = sorted((queue.get(timeout=600) for _ in workers), key=lambda item: item[0])
    for worker in workers:
        worker.join(timeout=120)
        assert worker.exitcode == 0
    sources = [output[1] for output in outputs]
    exact = sum(sources[1:], sources[0].copy())
    scale = np.max(np.abs(np.stack(sources)), axis=0)
    for key in outputs[0][2]:
        values, launched = outputs[0][2][key]
        for other in outputs[1:]:
            np.testing.assert_array_equal(values, other[2][key][0])
            assert launched == other[2][key][1]
        assert launched[-1][0] == 0 and len(launched) >= 4
        if key.startswith('f32/'):
            np.testing.assert_array_equal(values, exact)
        elif world == 1:
            np.testing.assert_array_equal(values, torch.from_numpy(exact).to(torch.bfloat16).float().numpy())
        else:
            assert np.all(np.abs(values - exact) <= 4 * 2.0 ** -8 * scale + 1e-30), key
    import conftest
    conftest.PARITY_NOTES.append(f'gradient exchange on device buffers over {backend}, world size {world}: fp32 / bf16 buckets x '
                                 'all-reduce / reduce-scatter + all-gather agree')


def test_bench_line_with_bf16_buckets_and_reduce_scatter_over_rccl(pkg):
    """``bench.py --force-dp --backend nccl --grad-wire bf16 --exchange-form reduce_scatter``: the whole timed loop with
    every gradient arena exchanged as bf16 reduce-scatter + all-gather buckets through RCCL on one rank; the line carries
    the schedule check (three compute streams under data parallelism against one)."""
    line = _bench_line('--force-dp', '--backend', 'nccl', '--grad-wire', 'bf16', '--exchange-form', 'reduce_scatter')
    config = line['config']
    assert config['gradient_wire'] == 'bf16' and config['gradient_exchange_form'] == 'reduce_scatter'
    assert 'saw 1 ranks' in config['collective_world'] and len(config['per_rank_ms_per_step']) == 1
    assert 'THREE compute streams' in config['streams']
    # (the gradients went through bf16: the schedule check compares two runs that both did, so it still holds)
    def held(check):      # ONE comparison against the fixed limit (round 5: no repetition, no limit scaled by the schedule's own noise)
        return check['within_limit'] and check['max_relative_loss_difference'] <= check['limit']
    assert held(config['schedule_check']), config['schedule_check']
    plain = _bench_line()
    assert held(plain['config']['schedule_check']), plain['config']['schedule_check']
    a, b = plain['config']['gradient_penalty_last'], config['gradient_penalty_last']
    assert a > 0 and b > 0 and abs(a - b) <= 0.1 * abs(a), (a, b)     # (two Adam updates from bf16-rounded gradients)
