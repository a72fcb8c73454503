"""Runs that reproduce the reference as a whole on the device:

* H12: an UN-INJECTED coefficient training run -- the product's own `seed_all`, set-up, loaders and `Experiment.draw_*` --
  reproduces golden g3 (reference srgan.py:286-301,364; utility.py:102-116);
* BASELINE.json configs[1] / [4] at their own batch of 128 in fp32 against the CPU oracle (the 16-bit modes at batch 128:
  test_blocked16_gpu.py).
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


def test_age_vgg_fp32_step_at_batch_128_matches_the_oracle(pkg, monkeypatch):
    """BASELINE.json configs[1] at its STATED batch against the ORACLE (the batch-128 steps were only
    compared with the product's own fp32 step): VGG-16 discriminator on 64 x 64 faces, 128 examples, fp32 -- the five
    logged losses within 1e-3 and the post-Adam weights of all three networks against the PyTorch-CPU restatement."""
    import srgan_amd.age.srgan as age
    from oracle import models as OM
    from test_steps_gpu import _hip_and_oracle_step
    monkeypatch.setattr(age, 'model_architecture', 'vgg')

    def configure(experiment):
        experiment.image_size = 64
    _hip_and_oracle_step(age.AgeExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=64), OM.VGG16(1, 64), OM.VGG16(1, 64)),
                         size=64, batch=128, d_scale=1.3)
    import conftest
    conftest.PARITY_NOTES.append('config 2 (age, VGG-16 @ 64 x 64) fp32 step at batch 128 checked against the CPU oracle')


@pytest.mark.parametrize('blocked', [False, True])
def test_driving_fp32_step_at_batch_128_matches_the_oracle(pkg, blocked):
    """BASELINE.json configs[4] at its stated per-device batch against the ORACLE: the DCGAN pair on 64 x 192 frames, 128
    examples, fp32 -- on the fp32 NCHW kernels and (``settings.blocked_fp32``) with the 4x4 / stride 2 stages of all three
    networks on fp32 tensors in the blocked layout (csrc/blocked16_k4s2.hip, dtype 0): the same 1e-3."""
    from srgan_amd.driving.srgan import DrivingExperiment
    from oracle import models as OM
    from test_steps_gpu import _hip_and_oracle_step
    size = (64, 192)

    def configure(experiment):
        experiment.image_size = size
    _hip_and_oracle_step(DrivingExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=size), OM.DCGANDiscriminator(image_size=size),
                                  OM.DCGANDiscriminator(image_size=size)),
                         size=size, batch=128, d_scale=2.2, settings_overrides=dict(blocked_fp32=blocked))
    import conftest
    conftest.PARITY_NOTES.append('config 5 (driving, DCGAN @ 64 x 192) fp32 step at batch 128 checked against the CPU oracle'
                                 + (' (4x4 / stride 2 stages on blocked fp32 tensors)' if blocked else ''))


