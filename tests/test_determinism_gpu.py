"""Determinism on the device (the reference's CPU results are bitwise repeatable): every sum that several
workgroups share finishes in a fixed order (csrc/split_finish.h), so contractions with a split K, the few-row ordered reduction
and whole training iterations -- on one stream or four, in every configuration -- give the same bits from run to run.  (The
HIP-graph replays against the eager tape: test_step_graph_gpu.py; the 16-bit path: test_blocked16_gpu.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def test_ordered_row_reduction_with_poisoned_partials_next_to_a_heavy_stream(pkg):
    """`chan_reduce_rows_ordered_kernel`: every workgroup's partial must be visible to the row's last workgroup.  The
    workspace is filled with NaN in front of every launch (a partial read before it was written shows), a 1 GiB copy
    loop keeps HBM busy on a second stream, 300 launches over three row shapes: every result equals the first, bit for bit,
    and equals torch's float64 sum."""
    from srgan_amd import functional as F, _lib
    generator = torch.Generator().manual_seed(5)
    source = torch.empty(1 << 28, dtype=torch.float32, device='cuda').normal_()
    sink = torch.empty_like(source)
    side = torch.cuda.Stream()
    for rows, length in [(16, 3 * 512 * 512), (2, 3 * 64 * 64), (16, 3 * 224 * 224)]:
        x = torch.randn(rows, length, generator=generator)
        xv = F.leaf(x.cuda())
        want = (x.double() * x.double()).sum(1)
        handle = _lib.stream_handle()
        workspace = _lib._workspaces[(torch.cuda.current_device(), handle)]
        first = None
        results = []
        for iteration in range(100):
            if iteration % 10 == 0:
                with torch.cuda.stream(side):
                    sink.copy_(source)
            workspace.fill_(float('nan'))
            results.append(F.row_dot(xv, xv).data.clone())
        torch.cuda.synchronize()
        first = results[0]
        assert_close(first.cpu().numpy(), want.float().numpy(), rtol=2e-6, what=f'squared norms {rows} x {length}')
        for iteration, result in enumerate(results):
            assert torch.equal(result, first), f'{rows} x {length}: launch {iteration} differs from the first ({result} vs {first})'


SPLIT_CASES = [
    # what, x shape, weight shape, stride, padding      (every one splits K over several workgroups at these sizes)
    ('3x3 growth convolution on small planes (conv3x3_lds_kernel, ordered finish)', (2, 128, 32, 32), (32, 128, 3, 3), 1, 1),
    ('3x3 on 16 x 16 planes', (4, 128, 16, 16), (32, 128, 3, 3), 1, 1),
    ('1x1 bottleneck on ragged 14 x 14 planes (pointwise_kernel, ordered finish)', (4, 512, 14, 14), (128, 512, 1, 1), 1, 0),
    ('1x1 on 7 x 7 planes', (4, 896, 7, 7), (128, 896, 1, 1), 1, 0),
    ('k4 / s2 / p1 strided convolution (gg_mfma_kernel, ordered finish)', (2, 64, 16, 16), (128, 64, 4, 4), 2, 1),
    ('7x7 / s2 / p3 on a small image (generic kernel)', (1, 8, 30, 30), (16, 8, 7, 7), 2, 3),
    ('2x2 / s2 map-module convolution (1024 K slices of the generic kernel, partial outputs + ordered reduce)', (4, 8, 128, 128), (16, 8, 2, 2), 2, 0),
]


@pytest.mark.parametrize('case', SPLIT_CASES, ids=[c[0].split(' (')[0] for c in SPLIT_CASES])
def test_split_k_contractions_finish_in_a_fixed_order(pkg, case):
    """A contraction that splits K over several workgroups (split_finish.h; reference: every nn.Conv2d / ConvTranspose2d call
    of the small planes, e.g. crowd/models.py:344-345) gives the SAME BITS on every run -- with NaN in the workspace in front of
    every launch and a copy loop hammering HBM on a second stream -- for the forward pass, the data gradient and the weight
    gradient, and the values are torch's.  (Round 4: fp32 atomics in arrival order; two runs differed at rounding level.)"""
    from srgan_amd import functional as F, _lib
    what, x_shape, w_shape, stride, padding = case
    assert _lib.library().srgan_split_is_ordered(_lib.stream_handle()) == 1
    generator = torch.Generator().manual_seed(17)
    x = torch.randn(x_shape, generator=generator)
    w = torch.randn(w_shape, generator=generator) / (w_shape[1] * w_shape[2] * w_shape[3]) ** 0.5
    y_ref = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, padding)
    gy = torch.randn(y_ref.shape, generator=generator)
    gx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), stride, padding)
    gw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), stride, padding)
    xv, wv, gv = F.leaf(x.cuda()), F.leaf(w.cuda()), F.leaf(gy.cuda())
    handle = _lib.stream_handle()
    workspace = _lib._workspaces[(torch.cuda.current_device(), handle)]
    source = torch.empty(1 << 27, dtype=torch.float32, device='cuda').normal_()
    sink = torch.empty_like(source)
    side = torch.cuda.Stream()
    runs = []
    for iteration in range(12):
        if iteration % 4 == 0:
            with torch.cuda.stream(side):
                sink.copy_(source)
        outputs = []
        pairs = ((stride, stride), (padding, padding))
        for launch in (lambda: F.conv2d(xv, wv, None, *pairs),
                       lambda: F.conv2d_backward_data(gv, wv, x.shape, *pairs),
                       lambda: F.conv2d_backward_weight(xv, gv, w.shape, *pairs)):
            workspace.fill_(float('nan'))
            outputs.append(launch().data.clone())
        runs.append(outputs)
    torch.cuda.synchronize()
    for got, want, name in zip(runs[0], (y_ref, gx_ref, gw_ref), ('forward', 'data gradient', 'weight gradient')):
        scale = float(want.abs().max())
        assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * scale, f'{what}: {name}'
    for iteration, outputs in enumerate(runs[1:], 1):
        for got, first, name in zip(outputs, runs[0], ('forward', 'data gradient', 'weight gradient')):
            assert torch.equal(got, first), f'{what}: {name} of run {iteration} differs from run 0'


def test_an_iteration_is_bit_reproducible_across_runs_and_schedules(pkg):
    """The reference's CPU path is bitwise repeatable; round 4's HIP path was not: K slices and parameter sums
    met through fp32 atomics in arrival order, and a ReLU mask that flipped at rounding level moved the gradient penalty of
    two runs by up to 6e-4.  Round 5: every K split, every grouped weight gradient and every parameter sum finishes in a
    fixed order through the stream's workspace (csrc/split_finish.h) -- one full iteration (DNN step, discriminator step with
    the gradient penalty, generator step, three Adam updates) at crowd 64 x 64, batch 2 (every plane K-split) gives the SAME
    BITS in all six losses and in every updated weight, run after run, on one stream and on four; between the two schedules
    the five losses computed before the discriminator's update are the same bits as well."""
    import test_parallel_gpu as parallel_tests
    first, first_weights = parallel_tests._step(None)
    again, again_weights = parallel_tests._step(None)
    streamed, streamed_weights = parallel_tests._step(None, streams=True)
    streamed_again, streamed_again_weights = parallel_tests._step(None, streams=True)
    for key in first:
        assert first[key] == again[key], (key, first[key], again[key])
        assert streamed[key] == streamed_again[key], (key, 'four streams, two runs', streamed[key], streamed_again[key])
    # across schedules: the losses computed BEFORE the discriminator's update are functions of the weights, the batch and the
    # draws alone; the generator loss comes after it, and on four streams the penalty chain's parameter gradients are added to
    # the other three losses' as a block (srgan.py: gradients_into_alternate) -- another association of the same fp32 sum
    for key in ('dnn_loss', 'labeled_loss', 'unlabeled_loss', 'fake_loss', 'gradient_penalty'):
        assert first[key] == streamed[key], (key, 'one stream vs four', first[key], streamed[key])
    assert abs(first['generator_loss'] - streamed['generator_loss']) <= 1e-6 * abs(first['generator_loss'])
    assert first['gradient_penalty'] > 0.0
    for key, value in first_weights.items():
        assert np.array_equal(value, again_weights[key]), (key, 'two runs', float(np.abs(value - again_weights[key]).max()))
        assert np.array_equal(streamed_weights[key], streamed_again_weights[key]), (key, 'four streams, two runs')


def _one_iteration(experiment_class, configure, size, batch, d_scale, overrides=None):
    """Losses and updated weights of one dnn + gan iteration of a task experiment from seeded weights, inputs and draws."""
    from srgan_amd.settings import Settings
    from srgan_amd.utility import SummaryWriter, seed_all
    from test_steps_gpu import finish_setup
    settings = Settings()
    settings.batch_size = batch
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e2, 1e1
    settings.gradient_penalty_multiplier = 1e2
    for key, value in (overrides or {}).items():
        setattr(settings, key, value)
    experiment = experiment_class(settings)
    configure(experiment)
    seed_all(0)
    experiment.model_setup()
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear)):
                module.weight.mul_(d_scale)
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    finish_setup(experiment)
    height, width = (size, size) if isinstance(size, int) else size
    generator = torch.Generator().manual_seed(1)
    x = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    y = torch.rand(batch, generator=generator) * 85 + 10
    experiment.injected_draws = {'z_d': torch.randn(batch, 256, generator=generator), 'z_g': torch.randn(batch, 256, generator=generator),
                                 'alpha': torch.rand(batch, 1, 1, 1, generator=generator)}
    experiment.dnn_training_step(x.cuda(), y.cuda(), 0)
    experiment.gan_training_step(x.cuda(), y.cuda(), u.cuda(), 0)
    experiment.join_dnn_stream()
    torch.cuda.synchronize()
    losses = {k: float(v.item()) for k, v in experiment.last_losses.items() if v is not None}
    weights = {name: getattr(experiment, name)._srgan_arena.data.cpu().numpy().copy() for name in ('D', 'DNN', 'G')}
    return losses, weights


@pytest.mark.parametrize('task', ['driving', 'age-vgg', 'driving-fp16'])
def test_other_configurations_are_bit_reproducible_too(pkg, monkeypatch, task):
    """The DCGAN pair on 64 x 192 driving frames (every k4 / s2 convolution, transposed convolution and their weight gradients on
    the generic kernel: ordered in-kernel finish, partial outputs + ordered reduce, lanes-along-K) in fp32 and in its named
    fp16 mode, and the age task's VGG-16 discriminator at 64 x 64 (3x3 kernels, linear layers): two runs of one iteration give
    the same bits in every loss and every updated weight (BASELINE.json configs 2 and 5; the reference's CPU path is
    repeatable)."""
    if task.startswith('driving'):
        from srgan_amd.driving.srgan import DrivingExperiment as experiment_class
        size, batch, d_scale = (64, 192), 8, 2.2

        def configure(experiment):
            experiment.image_size = size
    else:
        import srgan_amd.age.srgan as age
        monkeypatch.setattr(age, 'model_architecture', 'vgg')
        experiment_class, size, batch, d_scale = age.AgeExperiment, 64, 4, 1.3

        def configure(experiment):
            experiment.image_size = 64
    overrides = dict(compute_dtype='f16', gradient_penalty_dtype='f32', loss_scale=256.0) if task.endswith('fp16') else None
    first, first_weights = _one_iteration(experiment_class, configure, size, batch, d_scale, overrides)
    again, again_weights = _one_iteration(experiment_class, configure, size, batch, d_scale, overrides)
    assert first['gradient_penalty'] > 0.0 and all(np.isfinite(v) for v in first.values())
    for key in first:
        assert first[key] == again[key], (task, key, first[key], again[key])
    for name in first_weights:
        assert np.array_equal(first_weights[name], again_weights[name]), (task, name, float(np.abs(first_weights[name] - again_weights[name]).max()))
