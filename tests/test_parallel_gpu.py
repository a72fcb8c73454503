"""The HIP engine's data-parallel path on the test box's GPU(s).

* World size 2: two ranks, each with half of the global batch, must reproduce the single-process training step on the
  whole batch -- logged losses and the post-Adam weights of all three networks.  Over ``gloo`` both ranks share cuda:0
  (gloo moves the device tensors through the host); over ``nccl`` (= RCCL) each rank needs its own device, so that case
  is skipped -- visibly -- on a one-GPU box.
* World size 1 over ``nccl`` with the exchanges forced on (``DataParallel(force=True)``): the feature-sum all-reduce, the
  asynchronous bucketed gradient exchange with its stream-side ``work.wait()``, ``broadcast_parameters`` and
  ``broadcast_object`` all go through RCCL on ONE rank and must leave the step unchanged.  This is the first contact with
  the backend the 8-GPU runs use (reference srgan.py:264,295,304 are where the gradients complete)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import load_golden

pytestmark = pytest.mark.gpu
GOLDEN, SIZE = 'g7c_crowd64_gp_active', 64


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _step(dp, queue=None, broadcast=False, streams=False, normalize=False):
    import srgan_amd  # noqa: F401
    from test_steps_gpu import make_experiment, finish_setup, run_step, crowd_inputs
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    g = load_golden(GOLDEN)
    batch = int(g['batch_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=SIZE), KnnDenseNetCat(image_size=SIZE), KnnDenseNetCat(image_size=SIZE)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
             gradient_penalty_multiplier=1e2, map_multiplier=1e-3, overlap_dnn_step=streams, wgrad_stream=streams,
             overlap_generator_forwards=streams, overlap_gradient_penalty=streams, normalize_feature_norm=normalize),
        crowd=True)
    experiment.dp = dp
    scale = float(g['d_scale'])
    if scale != 1.0:
        with torch.no_grad():
            for m in experiment.D.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.weight.mul_(scale)
    finish_setup(experiment)
    if broadcast:
        for module in (experiment.D, experiment.DNN, experiment.G):
            dp.broadcast_parameters(module._srgan_arena)
    launched = {}
    if dp is not None:                 # record which buckets every network's exchange launched
        exchange_of = experiment.gradient_exchange

        def recording(module):
            exchange = exchange_of(module)
            name = next(n for n in ('D', 'DNN', 'G') if getattr(experiment, n) is module)
            if exchange is not None:
                launched.setdefault(name, []).append(exchange.launched)
            return exchange
        experiment.gradient_exchange = recording
    shard = dp.shard if dp is not None else (lambda t: t)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    x, y, u = crowd_inputs(generator, batch, SIZE)
    x, u = shard(x).cuda(), shard(u).cuda()
    y = tuple(shard(t).cuda() for t in y)
    sharded = {f's0/{k}': shard(torch.from_numpy(g[f's0/{k}'])).numpy() for k in ('z_d', 'z_g', 'alpha')}
    result = run_step(experiment, x, y, u, 0, sharded)
    experiment.finish_update()       # the generator's update waits for its (asynchronous) gradient exchange until G is used again
    if dp is not None:     # the logged gradient-norm mean is a local mean
        result['gradient_norm_mean'] = dp.all_reduce_sum_float(result['gradient_norm_mean']) / dp.world_size
    tensors = {}
    for name, module in (('D', experiment.D), ('DNN', experiment.DNN), ('G', experiment.G)):
        tensors.update({f'{name}/{k}': v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()})
    if queue is not None:
        report = {name: [list(b) for b in buckets] for name, buckets in launched.items()}
        if getattr(dp, 'abi', None) is not None:       # collectives issued through the C ABI's RCCL entry points
            report['abi_collectives'] = dp.abi.calls
        queue.put((dp.rank, result, tensors, report))
    return result, tensors


def _worker(rank, world_size, port, queue, backend='gloo', force=False, streams=False, normalize=False, abi=False):
    device = rank if backend == 'nccl' else 0          # RCCL: one device per rank; gloo: both ranks on cuda:0
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK=str(device), HSA_ENABLE_IPC_MODE_LEGACY='0', SRGAN_ABI_COLLECTIVES='1' if abi else '0')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.set_device(device)
    import srgan_amd  # noqa: F401
    from srgan_amd.parallel import DataParallel
    dp = DataParallel.from_environment(backend, force=force)
    assert (dp.device_backend if backend == 'nccl' else dp.host_backend) == backend, (dp.device_backend, dp.host_backend)
    if force:                                            # what Experiment.train() does before the first step
        assert dp.broadcast_object({'trial': 'x', 'skip': False}) == {'trial': 'x', 'skip': False}
    assert (dp.abi is not None) == bool(abi)
    _step(dp, queue, broadcast=force, streams=streams, normalize=normalize)
    dp.barrier()
    if dp.abi is not None:
        dp.abi.close()
    torch.distributed.destroy_process_group()


def _run_ranks(world_size, backend, force=False, streams=False, normalize=False, abi=False):
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(target=_worker, args=(rank, world_size, port, queue, backend, force, streams, normalize, abi))
               for rank in range(world_size)]
    for worker in workers:
        worker.start()
    import queue as queues
    outputs, waited = [], 0
    while len(outputs) < len(workers):
        try:
            outputs.append(queue.get(timeout=5))
        except queues.Empty:
            waited += 5
            dead = [worker.exitcode for worker in workers if worker.exitcode not in (None, 0)]
            assert not dead and waited < 600, f'a rank died (exit codes {dead}) or nothing arrived for {waited} s'
    for worker in workers:
        worker.join(timeout=120)
        assert worker.exitcode == 0
    return outputs


@pytest.mark.parametrize('streams', [False, True])
def test_one_rank_over_rccl_with_the_exchanges_forced_equals_the_plain_step(streams):
    """``streams``: bench.py's side-stream schedule on top (DNN step, penalty chain, weight gradients, D(unlabeled)): the
    collectives are then issued from several streams, D's exchange after the two chains have joined."""
    reference_result, reference_tensors = _step(None)
    (rank, result, tensors, launched), = _run_ranks(1, 'nccl', force=True, streams=streams)
    assert launched['DNN'] and launched['D'] and launched['G'], launched     # every arena went through all_reduce
    for key, value in reference_result.items():
        # one rank: every all-reduce is the identity, so only the order of the fp32 atomics differs between the runs (the
        # generator loss is evaluated after D's Adam step, where a rounding-level gradient may flip an update's sign:
        # observed 2e-5 between two plain runs; the parity bar is 1e-3)
        assert abs(result[key] - value) <= 1e-4 * max(abs(value), 1e-6), (key, result[key], value)
    for key, value in reference_tensors.items():
        limit = 2.2e-4 + 1e-3 * float(np.abs(value).max())
        assert float(np.abs(tensors[key] - value).max()) <= limit, key
        assert float(np.abs(tensors[key] - value).mean()) <= 0.05 * limit, (key, 'mean difference')
    import conftest
    conftest.PARITY_NOTES.append('data-parallel step over nccl (RCCL), world size 1 with the exchanges forced on: '
                                 f'{sum(len(b) for runs in launched.values() for b in runs)} gradient buckets all-reduced, '
                                 f'losses equal the plain step (side streams {"on" if streams else "off"})')


@pytest.mark.parametrize('backend,streams,normalize', [('gloo', False, False), ('gloo', True, False), ('gloo', False, True),
                                                       ('nccl', False, False), ('nccl', True, False)])
def test_two_ranks_equal_one_rank_on_the_global_batch(backend, streams, normalize):
    """``normalize``: ``settings.normalize_feature_norm`` -- the reference's branch whose distance runs over the (B, F)
    broadcast (srgan.py:444-447): every rank holds its own rows of it, the batch means pass their gradients through an
    all-reduce of their own."""
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('two ranks over nccl (RCCL) need two GPUs; this box has %d (the world-size-1 nccl test above ran)'
                    % torch.cuda.device_count())
    reference_result, reference_tensors = _step(None, normalize=normalize)
    outputs = [output[:3] for output in _run_ranks(2, backend, streams=streams, normalize=normalize)]
    for rank, result, tensors in outputs:
        for key, value in reference_result.items():
            assert abs(result[key] - value) <= 1e-3 * max(abs(value), 1e-6), (rank, key, result[key], value)
        for key, value in reference_tensors.items():
            # Adam's first step moves every element by ~lr * sign(gradient): where the gradient is at rounding level
            # the sign is arbitrary, so an element may legitimately differ by two learning rates.
            limit = 2.2e-4 + 1e-3 * float(np.abs(value).max())
            error = float(np.abs(tensors[key] - value).max())
            assert error <= limit, (rank, key, error, limit)
            assert float(np.abs(tensors[key] - value).mean()) <= 0.05 * limit, (rank, key, 'mean difference')
    for key in outputs[0][2]:      # both ranks hold identical weights after the synchronised update
        np.testing.assert_array_equal(outputs[0][2][key], outputs[1][2][key])
