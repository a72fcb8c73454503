"""Host-side mechanisms around the kernels on the device: the side-stream schedules (DNN step, gradient-penalty chain, D(unlabeled)
on streams of their own) against the goldens and on NaN-poisoned allocations, and the cached launch tables of the fused dense
blocks against changed batch-norm statistics."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN, SIZE = 'g7c_crowd64_gp_active', 64


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def _crowd_experiment(g, **overrides):
    from test_steps_gpu import make_experiment
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    settings = dict(batch_size=int(g['batch_size']), matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
                    gradient_penalty_multiplier=1e2, map_multiplier=1e-3)
    settings.update(overrides)
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=SIZE), KnnDenseNetCat(image_size=SIZE), KnnDenseNetCat(image_size=SIZE)),
        settings, crowd=True)
    scale = float(g['d_scale'])
    if scale != 1.0:
        with torch.no_grad():
            for m in experiment.D.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.weight.mul_(scale)
    return experiment


def _crowd_batches(g):
    from test_steps_gpu import crowd_inputs
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    x, y, u = crowd_inputs(generator, int(g['batch_size']), SIZE)
    return x.cuda(), tuple(t.cuda() for t in y), u.cuda()


def _other_statistics(module, seed):
    """A state dict of ``module`` whose batch-norm running statistics differ from the current ones."""
    generator = torch.Generator().manual_seed(seed)
    state = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    for key, value in state.items():
        if key.endswith('running_mean'):
            value += 0.2 * torch.randn(value.shape, generator=generator)
        elif key.endswith('running_var'):
            value *= 0.5 + torch.rand(value.shape, generator=generator)
    return state


def test_new_running_statistics_reach_the_cached_launch_tables(pkg):
    """the grouped weight-gradient / batched-reduction / tangent tables of fused.py keep raw pointers to every
    layer's inverse standard deviation and mean.  After ``load_state_dict`` with other running statistics the next step
    must use the NEW statistics everywhere: compared with an experiment that has no cached table (grouped launches off,
    statistics loaded before its first step).  Learning rate 0, so both see the same weights."""
    from srgan_amd import fused
    from test_steps_gpu import finish_setup, run_step
    g = load_golden(GOLDEN)
    x, y, u = _crowd_batches(g)

    def gradients(experiment):
        torch.cuda.synchronize()
        return {name: getattr(experiment, name)._srgan_arena.grad.detach().cpu().numpy().copy() for name in ('D', 'DNN')}

    cached = _crowd_experiment(g, learning_rate=0.0)
    finish_setup(cached)
    states = {name: _other_statistics(getattr(cached, name), seed) for seed, name in enumerate(('D', 'DNN'))}
    run_step(cached, x, y, u, 0, g)                       # builds every table from the first statistics
    before = gradients(cached)
    for name, state in states.items():
        getattr(cached, name).load_state_dict(state)
    result_cached = run_step(cached, x, y, u, 0, g)
    after = gradients(cached)

    saved = fused.GROUPED_WGRAD, fused.BATCHED_REDUCE
    fused.GROUPED_WGRAD = fused.BATCHED_REDUCE = False
    try:
        fresh = _crowd_experiment(g, learning_rate=0.0)
        for name, state in states.items():
            getattr(fresh, name).load_state_dict(state)
        finish_setup(fresh)
        result_fresh = run_step(fresh, x, y, u, 0, g)
        expected = gradients(fresh)
    finally:
        fused.GROUPED_WGRAD, fused.BATCHED_REDUCE = saved
    for key, value in result_fresh.items():
        assert abs(result_cached[key] - value) <= 1e-4 * max(abs(value), 1e-6), (key, result_cached[key], value)
    for name in ('D', 'DNN'):
        scale = float(np.abs(expected[name]).max())
        stale = float(np.abs(before[name] - expected[name]).max())
        error = float(np.abs(after[name] - expected[name]).max())
        print(f'[statistics] {name}: gradient scale {scale:.3e}, with the old statistics off by {stale:.3e}, after the reload {error:.3e}')
        # (the two runs sum in different orders -- grouped launches, fp32 atomics -- through a penalty-active double backward)
        assert error <= 2e-3 * scale, (name, error, scale)
        assert stale > 10 * error and stale > 1e-2 * scale, (name, 'the statistics did not change anything', stale, error)


STREAM_CASES = [('g7b_crowd64', 2, False), ('g7b_crowd64', 1, True), ('g7c_crowd64_gp_active', 1, False),
                ('g7c_crowd64_gp_active', 1, True)]


@pytest.fixture
def single_stream_afterwards():
    from srgan_amd import fused
    saved = fused.WGRAD_STREAM
    yield
    fused.WGRAD_STREAM = saved


@pytest.mark.parametrize('name,steps,reference_schedule', STREAM_CASES)
def test_crowd_steps_with_the_side_streams(pkg, single_stream_afterwards, name, steps, reference_schedule):
    """bench.py's default schedule for the timed region: the DNN step, the grouped weight gradients of every dense block
    (first and double backward) and the generator step's D(unlabeled) on side streams -- same results as the goldens."""
    import test_steps_gpu as reference_tests
    reference_tests.test_crowd_steps(pkg, name, SIZE, steps, reference_schedule, streams=True)


@pytest.mark.parametrize('name,steps,reference_schedule', STREAM_CASES[2:])
def test_side_streams_on_poisoned_allocations(pkg, single_stream_afterwards, monkeypatch, name, steps, reference_schedule):
    """Every allocation starts as NaN: a side stream that read a tensor before its producer wrote it, or after the
    allocator recycled it, would show."""
    import test_steps_gpu as reference_tests
    from srgan_amd import functional as F
    monkeypatch.setattr(F, 'POISON', True)
    reference_tests.test_crowd_steps(pkg, name, SIZE, steps, reference_schedule, streams=True)


