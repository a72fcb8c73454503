"""The mixed-precision modes of BASELINE.json configs 2 ("bf16") and 5 ("fp16 with fp32 GP") on the GPU.

The reference is fp32-only, so fp32 stays the parity path (1e-3, the other GPU test files); here the bf16 / fp16 MFMA
operand modes of the C ABI (``srgan_conv_desc.compute_dtype`` / ``srgan_gemm``) are held against that fp32 path with
their own, stated tolerances:

* operands that bf16 / fp16 represent EXACTLY (small integers) must give bit-identical results to the fp32 kernels --
  this pins the fragment layout of v_mfma_f32_32x32x16_{bf16,f16} (which k a lane's 8 values belong to) independently of
  rounding, for every pass of a convolution and for the linear layers;
* random operands: the error is bounded by the operand rounding (2^-9 relative per operand for bf16, 2^-12 for fp16):
  max |error| <= 1.5e-2 (bf16) / 2e-3 (fp16) of the largest output magnitude, and the mode is really active (error > 0);
* whole training steps of the configurations that name these modes against the fp32 CPU oracle: losses within 5e-2
  (bf16) / 2e-2 (fp16, with the gradient-penalty chain in fp32 and static loss scaling).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND = {'bf16': 1.5e-2, 'f16': 2e-3}


@pytest.fixture(scope='module')
def F():
    import srgan_amd  # noqa: F401
    from srgan_amd import functional
    assert torch.cuda.is_available()
    return functional


CONVS = [  # (n, c, h, w, k, r, stride, pad): VGG 3x3 (32-, 16-, 8- and 4-wide planes, channel tails), DCGAN k4 s2 p1,
    # stem-like 7x7 s2, 1x1, ragged extents
    (2, 64, 16, 16, 64, 3, 1, 1), (3, 24, 12, 20, 40, 4, 2, 1), (2, 3, 31, 29, 16, 7, 2, 3), (2, 96, 8, 8, 48, 1, 1, 0),
    (1, 130, 9, 7, 70, 3, 1, 1), (2, 48, 40, 36, 100, 3, 1, 1), (4, 128, 8, 8, 96, 3, 1, 1), (3, 64, 4, 4, 64, 3, 1, 1),
    (2, 3, 64, 64, 64, 3, 1, 1),
    # the small-plane kernel (whole 4 x 4 / 8 x 8 images side by side in a tile): ragged image groups, channel tails on
    # both sides, a 32-row tile, a K split with atomics
    (19, 40, 4, 4, 72, 3, 1, 1), (5, 136, 8, 8, 24, 3, 1, 1), (32, 256, 4, 4, 128, 3, 1, 1), (9, 32, 8, 8, 200, 3, 1, 1),
    # three input channels: the data gradient has three output rows (the image gradient of the penalty chain)
    (3, 3, 24, 40, 48, 3, 1, 1), (2, 3, 32, 48, 64, 4, 2, 1),
    # k4 / s2 / p1 at the DCGAN discriminator's shapes on driving frames (round 5)
    (2, 64, 32, 96, 128, 4, 2, 1), (2, 128, 16, 48, 256, 4, 2, 1), (3, 20, 36, 44, 24, 4, 2, 1)]


def _passes(F, x, w, gy, stride, pad):
    """Forward, data gradient and weight gradient of one convolution in the active precision."""
    from srgan_amd.tape import backward
    xv, wv = F.leaf(x, requires_grad=True), F.leaf(w, requires_grad=True)
    y = F.conv2d(xv, wv, None, stride, pad)
    backward(F.sum_all(F.mul(y, F.leaf(gy))))
    return y.data.clone(), xv.grad.data.clone(), wv.grad.data.clone()


@pytest.mark.parametrize('mode', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CONVS)
def test_exactly_representable_operands_reproduce_the_fp32_kernels(F, mode, case):
    n, c, h, w, k, r, stride, pad = case
    generator = torch.Generator().manual_seed(hash(case) % 1000)
    draw = lambda *shape: torch.randint(-3, 4, shape, generator=generator).float().cuda()     # exact in bf16 and fp16
    oh, ow = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    x, weight, gy = draw(n, c, h, w), draw(k, c, r, r), draw(n, k, oh, ow)
    expected = _passes(F, x, weight, gy, stride, pad)
    with F.compute_dtype(mode):
        got = _passes(F, x, weight, gy, stride, pad)
    for name, a, b in zip(('forward', 'data gradient', 'weight gradient'), got, expected):
        assert torch.equal(a, b), f'{mode} {name} of {case}: max difference {float((a - b).abs().max())}'


@pytest.mark.parametrize('mode', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CONVS)
def test_random_operands_within_the_rounding_bound(F, mode, case):
    n, c, h, w, k, r, stride, pad = case
    generator = torch.Generator().manual_seed(7)
    oh, ow = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    x = torch.randn(n, c, h, w, generator=generator).cuda()
    weight = (torch.randn(k, c, r, r, generator=generator) * (c * r * r) ** -0.5).cuda()
    gy = torch.randn(n, k, oh, ow, generator=generator).cuda()
    expected = _passes(F, x, weight, gy, stride, pad)
    with F.compute_dtype(mode):
        got = _passes(F, x, weight, gy, stride, pad)
    for name, a, b in zip(('forward', 'data gradient', 'weight gradient'), got, expected):
        error = float((a - b).abs().max()) / float(b.abs().max())
        assert 0.0 < error <= BOUND[mode], f'{mode} {name} of {case}: relative error {error:.3e}'


@pytest.mark.parametrize('mode', ['bf16', 'f16'])
def test_linear_layers_and_transposed_convolutions(F, mode):
    from srgan_amd.tape import backward
    generator = torch.Generator().manual_seed(11)
    ints = lambda *shape: torch.randint(-3, 4, shape, generator=generator).float().cuda()
    x, weight, bias, gy = ints(37, 200), ints(70, 200), ints(70), ints(37, 70)

    def linear():
        xv, wv = F.leaf(x, requires_grad=True), F.leaf(weight, requires_grad=True)
        y = F.linear(xv, wv, F.leaf(bias))
        backward(F.sum_all(F.mul(y, F.leaf(gy))))
        return y.data.clone(), xv.grad.data.clone(), wv.grad.data.clone()
    z, tw, gz = ints(3, 20, 5, 6), ints(20, 12, 4, 4), ints(3, 12, 10, 12)

    def transposed():
        zv, wv = F.leaf(z, requires_grad=True), F.leaf(tw, requires_grad=True)
        y = F.conv_transpose2d(zv, wv, None, 2, 1)
        backward(F.sum_all(F.mul(y, F.leaf(gz))))
        return y.data.clone(), zv.grad.data.clone(), wv.grad.data.clone()
    for run in (linear, transposed):
        expected = run()
        with F.compute_dtype(mode):
            got = run()
        for a, b in zip(got, expected):
            assert torch.equal(a, b), f'{mode} {run.__name__}: max difference {float((a - b).abs().max())}'


def _step_against_fp32_oracle(experiment_class, configure, oracle_networks, size, batch, d_scale, settings_overrides, tolerance):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    from srgan_amd.settings import Settings
    from srgan_amd.utility import SummaryWriter, seed_all
    from oracle.experiment import OracleExperiment, Draws
    entry._cap_host_threads(torch)
    settings = Settings()
    settings.batch_size = batch
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e2, 1e1
    settings.gradient_penalty_multiplier = 1e2
    for key, value in settings_overrides.items():
        setattr(settings, key, value)
    experiment = experiment_class(settings)
    configure(experiment)
    seed_all(0)
    experiment.model_setup()
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear)):
                module.weight.mul_(d_scale)
    oracle_g, oracle_d, oracle_dnn = oracle_networks()
    for ours, theirs in ((experiment.G, oracle_g), (experiment.D, oracle_d), (experiment.DNN, oracle_dnn)):
        theirs.load_state_dict({k: v.detach().clone() for k, v in ours.state_dict().items()}, strict=True)
    oracle_settings = entry.settings_for_oracle(settings)
    oracle = OracleExperiment(oracle_settings, oracle_d, oracle_dnn, oracle_g)
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    height, width = (size, size) if isinstance(size, int) else size
    generator = torch.Generator().manual_seed(1)
    x = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    y = torch.rand(batch, generator=generator) * 85 + 10
    draws = Draws(torch.randn(batch, 256, generator=generator), torch.randn(batch, 256, generator=generator),
                  torch.rand(batch, 1, 1, 1, generator=generator))
    experiment.injected_draws = {'z_d': draws.z_d, 'z_g': draws.z_g, 'alpha': draws.alpha}
    experiment.dnn_training_step(x.cuda(), y.cuda(), 0)
    experiment.gan_training_step(x.cuda(), y.cuda(), u.cuda(), 0)
    torch.cuda.synchronize()
    oracle.dnn_training_step(x, y)
    expected = oracle.gan_training_step(x, y, u, 0, draws)
    got = {k: float(v.item()) for k, v in experiment.last_losses.items() if v is not None}
    worst = 0.0
    for key in ('labeled_loss', 'unlabeled_loss', 'fake_loss', 'gradient_penalty', 'generator_loss'):
        error = abs(got[key] - expected[key]) / max(abs(expected[key]), 1e-12)
        worst = max(worst, error)
        print(f'[{settings.compute_dtype}] {key}: hip {got[key]:.6g}  fp32 oracle {expected[key]:.6g}  rel {error:.2e}')
        assert error <= tolerance, (key, got[key], expected[key])
    assert expected['gradient_penalty'] > 1.0
    assert worst > 1e-7, 'results identical to fp32: the mixed-precision mode was not active'
    # the update applied to the weights is the fp32 one up to the mode's rounding: compare the mean step per tensor
    for name, ours, theirs in (('D', experiment.D, oracle_d), ('G', experiment.G, oracle_g)):
        reference = dict(theirs.named_parameters())
        for pname, p in ours.named_parameters():
            want, have = reference[pname].detach().numpy(), p.detach().cpu().numpy()
            assert np.isfinite(have).all()
            assert np.abs(have - want).max() <= 2.2e-4 + 1e-3 * np.abs(want).max(), f'{name} {pname}'


@pytest.mark.parametrize('penalty_dtype', ['f32', 'bf16'])
def test_age_vgg_step_in_bf16(monkeypatch, penalty_dtype):
    """BASELINE.json configs[1]: age SRGAN, VGG-16 discriminator on 64 x 64 faces, bf16 MFMA operands -- with the
    gradient-penalty chain in fp32 (the default) and in bf16 as well."""
    import srgan_amd.age.srgan as age
    from oracle import models as OM
    monkeypatch.setattr(age, 'model_architecture', 'vgg')

    def configure(experiment):
        experiment.image_size = 64
    _step_against_fp32_oracle(age.AgeExperiment, configure,
                              lambda: (OM.DCGANGenerator(image_size=64), OM.VGG16(1, 64), OM.VGG16(1, 64)),
                              size=64, batch=8, d_scale=1.3,
                              settings_overrides=dict(compute_dtype='bf16', gradient_penalty_dtype=penalty_dtype), tolerance=5e-2)


def test_driving_step_in_fp16_with_the_gradient_penalty_in_fp32():
    """BASELINE.json configs[4]: driving DCGAN pair on 64 x 192 frames, fp16 MFMA operands, the gradient-penalty chain in
    fp32 (settings.gradient_penalty_dtype, default 'f32') and static loss scaling."""
    import srgan_amd  # noqa: F401
    from srgan_amd.driving.srgan import DrivingExperiment
    from oracle import models as OM
    size = (64, 192)

    def configure(experiment):
        experiment.image_size = size
    _step_against_fp32_oracle(DrivingExperiment, configure,
                              lambda: (OM.DCGANGenerator(image_size=size), OM.DCGANDiscriminator(image_size=size),
                                       OM.DCGANDiscriminator(image_size=size)),
                              size=size, batch=8, d_scale=2.2,
                              settings_overrides=dict(compute_dtype='f16', gradient_penalty_dtype='f32', loss_scale=256.0),
                              tolerance=2e-2)


def test_capabilities_report_the_modes():
    import srgan_amd  # noqa: F401
    from srgan_amd import _lib
    caps = _lib.capabilities()
    assert caps.abi_version == _lib.library().srgan_version() and caps.arch == b'gfx950'
    assert caps.dtypes & 0x7 == 0x7 and caps.workspace_bytes == _lib.library().srgan_workspace_bytes()
