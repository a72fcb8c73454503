"""Round-5 GPU checks.

* H12 on the device: an UN-INJECTED coefficient training run -- the product's own `seed_all`, set-up, loaders and
  `Experiment.draw_*` -- reproduces golden g3 (reference srgan.py:286-301,364; utility.py:102-116).
* The few-row ordered reduction (the gradient penalty's per-example norms, reference srgan.py:371) under stress: NaN in the
  workspace partials before every launch, a bandwidth-heavy copy in flight on a second stream.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, golden_scalars, assert_close
from test_random_draws_cpu import coefficient_experiment, fetch_batches

pytestmark = pytest.mark.gpu
RTOL = 1e-3
TAGS = {'Generator/Loss': 'generator_loss', 'Discriminator/Labeled Loss': 'labeled_loss',
        'Discriminator/Unlabeled Loss': 'unlabeled_loss', 'Discriminator/Fake Loss': 'fake_loss',
        'Discriminator/Gradient Penalty': 'gradient_penalty', 'Discriminator/Gradient Norm': 'gradient_norm_mean',
        'Feature Norm/Labeled': 'feature_norm_labeled', 'Feature Norm/Unlabeled': 'feature_norm_unlabeled'}


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def test_uninjected_coefficient_run_reproduces_the_reference(pkg):
    """Nothing injected: three iterations of dnn_training_step + gan_training_step whose z_D, alpha and z_G come from the
    product's own draw path, in the reference's order, land on the reference's logged losses (golden g3)."""
    from srgan_amd.utility import SummaryWriter
    g = load_golden('g3_coefficient_srgan')
    experiment = coefficient_experiment(int(g['batch_size']))
    steps = int(g['steps'])
    batches = fetch_batches(experiment, steps)          # (the goldens fetched their batches before the first step too)
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    assert experiment.injected_draws is None
    for step, (x, y, u) in enumerate(batches):
        experiment.dnn_training_step(x.cuda(), y.cuda(), step)
        experiment.gan_training_step(x.cuda(), y.cuda(), u.cuda(), step)
        logged = {TAGS[tag]: values[-1][1] for tag, values in experiment.gan_summary_writer.scalars.items()}
        logged['dnn_loss'] = experiment.dnn_summary_writer.scalars['Discriminator/Labeled Loss'][-1][1]
        for key, value in golden_scalars(g, step).items():
            if key in logged:
                assert_close(logged[key], value, rtol=RTOL, atol=1e-6 if abs(value) < 1e-3 else 0.0,
                             what=f'un-injected step {step} {key}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, atol=1e-6,
                     what=f'un-injected step {step} gradient_norm')


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
