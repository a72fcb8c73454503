"""World size 2 on ONE GPU (gloo moves the device tensors) for the HIP engine's data-parallel path: two ranks,
each with half of the global batch, must reproduce the single-process training step on the whole batch -- logged
losses and the post-Adam weights of all three networks.  (The driver's multi-GPU runs use the same code over RCCL;
this test needs only the one GPU of the test box.)"""
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


def _step(dp, queue=None):
    import srgan_amd  # noqa: F401
    from test_steps_gpu import make_experiment, finish_setup, run_step, crowd_inputs
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    g = load_golden(GOLDEN)
    batch = int(g['batch_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=SIZE), KnnDenseNetCat(image_size=SIZE), KnnDenseNetCat(image_size=SIZE)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
             gradient_penalty_multiplier=1e2, map_multiplier=1e-3), crowd=True)
    experiment.dp = dp
    scale = float(g['d_scale'])
    if scale != 1.0:
        with torch.no_grad():
            for m in experiment.D.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.weight.mul_(scale)
    finish_setup(experiment)
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
        queue.put((dp.rank, result, tensors))
    return result, tensors


def _worker(rank, world_size, port, queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK='0')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.set_device(0)
    import srgan_amd  # noqa: F401
    from srgan_amd.parallel import DataParallel
    dp = DataParallel.from_environment('gloo')
    _step(dp, queue)
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_global_batch():
    reference_result, reference_tensors = _step(None)
    context = mp.get_context('spawn')
    queue = context.Queue()
    port = _free_port()
    workers = [context.Process(target=_worker, args=(rank, 2, port, queue)) for rank in range(2)]
    for worker in workers:
        worker.start()
    outputs = [queue.get(timeout=600) for _ in workers]
    for worker in workers:
        worker.join(timeout=120)
        assert worker.exitcode == 0
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
