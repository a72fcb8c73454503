"""GPU parity of whole training steps (HIP path through the C ABI) against the golden fixtures generated from
the unmodified reference (tests/golden/make_goldens.py) and against the CPU oracle on the same seeded inputs.
Tolerance 1e-3 relative (north_star); every case runs in both schedules (shared forwards / reference order)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, golden_state, golden_scalars, checksum, assert_close

pytestmark = pytest.mark.gpu
RTOL = 1e-3


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def make_experiment(builders, settings_overrides, sgan_bins=None, crowd=False, base_class=None):
    from srgan_amd.settings import Settings
    from srgan_amd.srgan import Experiment
    from srgan_amd.sgan import SganExperiment
    from srgan_amd.utility import SummaryWriter, seed_all
    base = SganExperiment if sgan_bins is not None else Experiment
    if crowd:
        from srgan_amd.crowd.srgan import CrowdExperiment
        base = CrowdExperiment
    if base_class is not None:
        base = base_class

    class _Experiment(base):
        def dataset_setup(self):
            pass

        def model_setup(self):
            self.G, self.D, self.DNN = builders()

        def validation_summaries(self, step):
            pass

    settings = Settings()
    for key, value in settings_overrides.items():
        setattr(settings, key, value)
    experiment = _Experiment(settings)
    if sgan_bins is not None:
        experiment.bins = sgan_bins
    seed_all(0)
    experiment.model_setup()
    experiment.dnn_summary_writer = SummaryWriter()
    experiment.gan_summary_writer = SummaryWriter()
    return experiment


def finish_setup(experiment):
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()


def run_step(experiment, x, y, u, step, g):
    experiment.injected_draws = {k: torch.from_numpy(g[f's{step}/{k}']) for k in ('z_d', 'z_g', 'alpha')}
    experiment.dnn_training_step(x, y, step)
    experiment.gan_training_step(x, y, u, step)
    scalars = {tag: values[-1][1] for tag, values in experiment.gan_summary_writer.scalars.items()}
    names = {'Generator/Loss': 'generator_loss', 'Discriminator/Labeled Loss': 'labeled_loss',
             'Discriminator/Unlabeled Loss': 'unlabeled_loss', 'Discriminator/Fake Loss': 'fake_loss',
             'Discriminator/Gradient Penalty': 'gradient_penalty', 'Discriminator/Gradient Norm': 'gradient_norm_mean',
             'Feature Norm/Labeled': 'feature_norm_labeled', 'Feature Norm/Unlabeled': 'feature_norm_unlabeled'}
    result = {names[tag]: value for tag, value in scalars.items()}
    result['dnn_loss'] = experiment.dnn_summary_writer.scalars['Discriminator/Labeled Loss'][-1][1]
    return result


def check(result, expected, what):
    for key, value in expected.items():
        if key in result:
            atol = 1e-6 if abs(value) < 1e-3 else 0.0
            assert_close(result[key], value, rtol=RTOL, atol=atol, what=f'{what} {key}')


def dev(array):
    return torch.from_numpy(np.asarray(array)).cuda()


@pytest.mark.parametrize('reference_schedule', [False, True])
@pytest.mark.parametrize('name,steps', [('g3_coefficient_srgan', 3), ('g3b_coefficient_srgan_gp_active', 2)])
def test_coefficient_srgan(pkg, name, steps, reference_schedule):
    from srgan_amd.coefficient.models import MLP, Generator
    g = load_golden(name)
    experiment = make_experiment(lambda: (Generator(10), MLP(10), MLP(10)),
                                 dict(batch_size=int(g['batch_size']), reference_schedule=reference_schedule))
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN'), (experiment.G, 'init/G')):
        module.load_state_dict(golden_state(g, prefix))
    finish_setup(experiment)
    for step in range(steps):
        x, y, u = (dev(g[f's{step}/{k}']) for k in ('x', 'y', 'u'))
        result = run_step(experiment, x, y, u, step, g)
        check(result, golden_scalars(g, step), f'{name} step {step}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, atol=1e-6,
                     what='gradient_norm')
        assert_close(experiment.labeled_features.cpu().numpy(), g[f's{step}/labeled_features'], rtol=RTOL, atol=1e-5,
                     what='labeled_features')
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=2e-5,
                     what=f'final D {key}')
    for key, value in golden_state(g, 'final/G').items():
        assert_close(experiment.G.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=2e-5,
                     what=f'final G {key}')


def test_coefficient_first_step_gradients(pkg):
    from srgan_amd.coefficient.models import MLP, Generator
    g = load_golden('g3b_coefficient_srgan_gp_active')
    experiment = make_experiment(lambda: (Generator(10), MLP(10), MLP(10)), dict(batch_size=int(g['batch_size'])))
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN'), (experiment.G, 'init/G')):
        module.load_state_dict(golden_state(g, prefix))
    finish_setup(experiment)
    captured = {}
    original_step = experiment.d_optimizer.step

    def capturing_step():       # the reference's D gradients are read just before d_optimizer.step()
        captured.update({n: p.grad.cpu().numpy().copy() for n, p in experiment.D.named_parameters()})
        original_step()
    experiment.d_optimizer.step = capturing_step
    x, y, u = (dev(g[f's0/{k}']) for k in ('x', 'y', 'u'))
    run_step(experiment, x, y, u, 0, g)
    for name, actual in captured.items():
        expected = g[f's0/d_grad/{name}']
        scale = np.abs(expected).max()
        assert_close(actual, expected, rtol=RTOL, atol=1e-4 * scale, what=f'D grad {name}')
    for name, parameter in experiment.G.named_parameters():
        expected = g[f's0/g_grad/{name}']
        scale = np.abs(expected).max()
        assert_close(parameter.grad.cpu().numpy(), expected, rtol=RTOL, atol=1e-4 * scale, what=f'G grad {name}')


@pytest.mark.parametrize('name', ['g4_coefficient_sgan', 'g4c_coefficient_sgan_gp_active'])
def test_coefficient_sgan(pkg, name):
    from srgan_amd.coefficient.models import SganMLP, Generator
    g = load_golden(name)
    experiment = make_experiment(lambda: (Generator(), SganMLP(10), SganMLP(10)), dict(batch_size=int(g['batch_size'])),
                                 sgan_bins=torch.linspace(-3, 3, 10))
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN'), (experiment.G, 'init/G')):
        module.load_state_dict(golden_state(g, prefix))
    finish_setup(experiment)
    for step in range(2):
        x, y, u = (dev(g[f's{step}/{k}']) for k in ('x', 'y', 'u'))
        result = run_step(experiment, x, y, u, step, g)
        check(result, golden_scalars(g, step), f'sgan step {step}')
    if 'gp_active' in name:
        assert result['gradient_penalty'] > 10.0
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=2e-5, what=key)


@pytest.mark.parametrize('reference_schedule', [False, True])
def test_tiny_dcgan_with_active_gradient_penalty(pkg, reference_schedule):
    from srgan_amd.age.models import Generator, Discriminator
    g = load_golden('g5_tiny_dcgan')
    experiment = make_experiment(lambda: (Generator(image_size=32, conv_dim=8), Discriminator(32, 8), Discriminator(32, 8)),
                                 dict(batch_size=4, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                                      gradient_penalty_multiplier=1e2, reference_schedule=reference_schedule))
    for key, value in golden_state(g, 'init/G').items():    # constructors reproduce the reference's random stream
        np.testing.assert_array_equal(experiment.G.state_dict()[key].numpy(), value.numpy())
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN')):
        module.load_state_dict(golden_state(g, prefix))
    finish_setup(experiment)
    for step in range(2):
        x, y, u = (dev(g[f's{step}/{k}']) for k in ('x', 'y', 'u'))
        result = run_step(experiment, x, y, u, step, g)
        check(result, golden_scalars(g, step), f'tiny dcgan step {step}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, what='gn')
        assert_close(experiment.fake_features.cpu().numpy(), g[f's{step}/fake_features'], rtol=RTOL, atol=1e-4,
                     what='fake features')
    assert result['gradient_penalty'] > 1.0
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=3e-5, what=key)
    for key, value in golden_state(g, 'final/G').items():
        assert_close(experiment.G.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=3e-5, what=key)


def test_layer_kats_with_double_backward(pkg):
    """_DenseLayer / _DenseBlock / _Transition / MapModule / stem: forward, input gradient and the parameter
    gradients of a gradient-penalty style double backward, against the reference's own layers (golden g6)."""
    from collections import OrderedDict
    from srgan_amd import functional as F, nn
    from srgan_amd.crowd.models import _DenseLayer, _DenseBlock, _Transition, MapModule
    from srgan_amd.tape import backward
    g = load_golden('g6_layers')
    stem = nn.Sequential(OrderedDict([('conv0', nn.Conv2d(3, 8, kernel_size=7, stride=2, padding=3, bias=False)),
                                      ('norm0', nn.BatchNorm2d(8)), ('relu0', nn.ReLU(inplace=True)),
                                      ('pool0', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
    modules = {'dense_layer': _DenseLayer(16, 8, 4, 0), 'dense_block': _DenseBlock(3, 8, 2, 4, 0),
               'transition': _Transition(32, 16), 'map_module': MapModule(16, 4, 32), 'stem': stem}
    for prefix, module in modules.items():
        module.load_state_dict(golden_state(g, f'{prefix}/state'))
        nn.flatten_parameters(module, torch.device('cuda', 0))
        x = F.leaf(dev(g[f'{prefix}/x']), requires_grad=True)
        y = module(x)
        ys = list(y) if isinstance(y, tuple) else [y]
        scalar = None
        for i, t in enumerate(ys):
            assert_close(t.cpu().numpy(), g[f'{prefix}/y{i}'], rtol=RTOL, atol=1e-5, what=f'{prefix} y{i}')
            term = F.sum_all(F.mul(t, F.leaf(dev(g[f'{prefix}/c{i}']))))
            scalar = term if scalar is None else F.add(scalar, term)
        (gx,) = backward(scalar, inputs=[x], create_graph=True)
        assert_close(gx.cpu().numpy(), g[f'{prefix}/gx'], rtol=RTOL, atol=1e-5, what=f'{prefix} gx')
        penalty = F.mean_all(F.square(F.row_norm(F.flatten2d(gx))))
        assert_close(penalty.cpu().numpy().reshape(()), g[f'{prefix}/penalty'], rtol=RTOL, what=f'{prefix} penalty')
        module._srgan_arena.zero_grad()
        backward(penalty)
        for name, parameter in module.named_parameters():
            key = f'{prefix}/ggparam/{name}'
            if key in g.files:
                expected = g[key]
                assert_close(parameter.grad.cpu().numpy(), expected, rtol=RTOL, atol=1e-4 * max(np.abs(expected).max(), 1e-8),
                             what=f'{prefix} penalty grad {name}')


def crowd_inputs(generator, batch, size):
    x = torch.rand(batch, 3, size, size, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, size, size, generator=generator) * 2 - 1
    heads = (torch.rand(batch, size, size, generator=generator) < 0.002).float()
    knn_map = torch.rand(batch, size, size, generator=generator)
    return x, (heads, knn_map), u


@pytest.mark.parametrize('name,size,steps,reference_schedule', [
    ('g7b_crowd64', 64, 2, False), ('g7b_crowd64', 64, 1, True), ('g7c_crowd64_gp_active', 64, 1, False),
    ('g7c_crowd64_gp_active', 64, 1, True), ('g7_crowd224', 224, 1, False)])
def test_crowd_steps(pkg, name, size, steps, reference_schedule, overlap=False, streams=False, extra_settings=None):
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    g = load_golden(name)
    batch = int(g['batch_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
             gradient_penalty_multiplier=1e2, map_multiplier=1e-3, reference_schedule=reference_schedule,
             overlap_dnn_step=overlap or streams, wgrad_stream=streams, overlap_generator_forwards=streams,
             overlap_gradient_penalty=streams, **(extra_settings or {})), crowd=True)
    scale = float(g['d_scale'])
    if scale != 1.0:
        with torch.no_grad():
            for m in experiment.D.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    m.weight.mul_(scale)
    for module, prefix in ((experiment.D, 'init_ck/D'), (experiment.DNN, 'init_ck/DNN'), (experiment.G, 'init_ck/G')):
        for pname, p in module.named_parameters():
            assert_close(checksum(p), g[f'{prefix}/{pname}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {pname}')
    finish_setup(experiment)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    batches = [crowd_inputs(generator, batch, size) for _ in range(steps)]
    from srgan_amd.tape import no_grad
    from srgan_amd.srgan import as_var
    with no_grad():
        density, count, maps = experiment.D(as_var(batches[0][0]))
    assert_close(count.cpu().numpy(), g['fwd/count'], rtol=RTOL, what='count')
    assert_close(experiment.D.features.cpu().numpy(), g['fwd/features'], rtol=RTOL, atol=1e-5, what='features')
    assert_close(checksum(maps.data)[:2], g['fwd/maps_ck'], rtol=RTOL, what='density-map checksums')
    step_view = max(size // 8, 1)
    assert_close(maps.cpu().numpy()[:, :, ::step_view, ::step_view], g['fwd/maps_sample'], rtol=RTOL, atol=1e-5,
                 what='density-map samples')
    assert float(density.data.abs().sum()) == float(g['fwd/density_abs_sum']) == 0.0
    for step, (x, y, u) in enumerate(batches):
        result = run_step(experiment, x.cuda(), tuple(t.cuda() for t in y), u.cuda(), step, g)
        check(result, golden_scalars(g, step), f'{name} step {step}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, what='gn')
    if 'gp_active' in name:
        assert result['gradient_penalty'] > 10.0
    if steps == int(np.sum([1 for k in g.files if k.endswith('/alpha')])):
        # post-step weights: compare per-tensor checksums (abs-sum) of the final parameters
        for prefix, module in (('final_ck/D', experiment.D), ('final_ck/G', experiment.G)):
            for pname, p in module.named_parameters():
                expected = g[f'{prefix}/{pname}']
                # Adam's first step moves every element by ~lr * sign(g): elements whose gradient is ~0 are
                # ill-conditioned, so allow a couple of lr-sized element differences on top of 1e-3 relative.
                assert_close(checksum(p)[1], expected[1], rtol=RTOL, atol=4e-4, what=f'{prefix} {pname} abs-sum')


def test_crowd_dggan(pkg):
    """the dual-goal GAN on the crowd task (reference crowd/dggan.py on KnnDenseNetCatDggan) against
    the fixture generated from the unmodified reference losses; discriminator scaled so that the gradient penalty on the
    real/fake scores is active."""
    from srgan_amd.crowd.dggan import CrowdDgganExperiment
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCatDggan
    from srgan_amd.srgan import as_var
    from srgan_amd.tape import no_grad
    g = load_golden('g10_crowd_dggan64_gp_active')
    batch, size = int(g['batch_size']), int(g['image_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCatDggan(image_size=size), KnnDenseNetCatDggan(image_size=size)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
             gradient_penalty_multiplier=1e2, map_multiplier=1e-3), base_class=CrowdDgganExperiment)
    assert experiment.settings.dggan_loss_multiplier == float(g['dggan_loss_multiplier'])
    with torch.no_grad():
        for m in experiment.D.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                m.weight.mul_(float(g['d_scale']))
    for module, prefix in ((experiment.D, 'init_ck/D'), (experiment.DNN, 'init_ck/DNN'), (experiment.G, 'init_ck/G')):
        for pname, p in module.named_parameters():
            assert_close(checksum(p), g[f'{prefix}/{pname}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {pname}')
    finish_setup(experiment)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    batches = [crowd_inputs(generator, batch, size) for _ in range(2)]
    with no_grad():
        _, count, maps = experiment.D(as_var(batches[0][0]))
    assert_close(count.cpu().numpy(), g['fwd/count'], rtol=RTOL, what='count')
    assert_close(experiment.D.real_label.cpu().numpy(), g['fwd/real_label'], rtol=RTOL, what='real/fake score')
    assert_close(checksum(maps.data)[:2], g['fwd/maps_ck'], rtol=RTOL, what='density-map checksums')
    for step, (x, y, u) in enumerate(batches):
        result = run_step(experiment, x.cuda(), tuple(t.cuda() for t in y), u.cuda(), step, g)
        expected = golden_scalars(g, step)
        if step == 0:
            check(result, expected, 'crowd dggan step 0')
            assert_close(experiment.gradient_norm.cpu().numpy(), g['s0/gradient_norm'], rtol=RTOL, what='gn')
        else:
            # The second step starts from weights that went through one Adam update: every element moved by
            # ~lr * sign(gradient), which is ill-conditioned where the gradient is ~0, and this fixture's discriminator
            # is scaled by 1.26 per layer to make the penalty active, so that difference is amplified: the real/fake
            # scores still agree to 4e-4, the penalty (100 * (norm - 1)^2 at norm 1.5: 6x the norm's error) and the
            # cross-entropy of a score near -9 (= exp(score): relative error = absolute error of the score) less so.
            for key, value in expected.items():
                if key == 'unlabeled_loss':
                    assert abs(np.log(result[key]) - np.log(value)) < 1e-3 * abs(np.log(value / 1e4)), key
                elif key in result:
                    assert_close(result[key], value, rtol=1e-2, atol=0.0, what=f'crowd dggan step 1 {key}')
            # (the weight gradients are summed with fp32 atomics in a run-dependent order: which near-zero elements flip
            # their first Adam step differs between runs, observed 4e-4 ... 2.5e-3 on the second step's gradient norms)
            assert_close(experiment.gradient_norm.cpu().numpy(), g['s1/gradient_norm'], rtol=5e-3, what='gn')
        assert result['gradient_penalty'] > 10.0
    for prefix, module in (('final_ck/D', experiment.D), ('final_ck/G', experiment.G)):
        for pname, p in module.named_parameters():
            assert_close(checksum(p)[1], g[f'{prefix}/{pname}'][1], rtol=RTOL, atol=8e-4, what=f'{prefix} {pname} abs-sum')


def test_age_dcgan128(pkg):
    from srgan_amd.age.models import Generator, Discriminator
    g = load_golden('g8_age_dcgan128')
    experiment = make_experiment(lambda: (Generator(), Discriminator(), Discriminator()),
                                 dict(batch_size=4, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                                      gradient_penalty_multiplier=1e2))
    for pname, p in experiment.D.named_parameters():
        assert_close(checksum(p), g[f'init_ck/D/{pname}'], rtol=1e-9, atol=1e-12, what=pname)
    finish_setup(experiment)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    for step in range(2):
        x = torch.rand(4, 3, 128, 128, generator=generator) * 2 - 1
        u = torch.rand(4, 3, 128, 128, generator=generator) * 2 - 1
        y = torch.rand(4, generator=generator) * 85 + 10
        result = run_step(experiment, x.cuda(), y.cuda(), u.cuda(), step, g)
        check(result, golden_scalars(g, step), f'age step {step}')
    for prefix, module in (('final_ck/D', experiment.D), ('final_ck/G', experiment.G)):
        for pname, p in module.named_parameters():
            assert_close(checksum(p)[1], g[f'{prefix}/{pname}'][1], rtol=RTOL, atol=4e-4, what=f'{prefix} {pname}')


def test_vgg224(pkg):
    from srgan_amd.age.models import Generator
    from srgan_amd.age.vgg import vgg16
    g = load_golden('g8b_vgg224')
    experiment = make_experiment(lambda: (Generator(image_size=224), vgg16(num_classes=1), vgg16(num_classes=1)),
                                 dict(batch_size=2, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                                      gradient_penalty_multiplier=1e2))
    for pname, p in experiment.D.named_parameters():
        assert_close(checksum(p), g[f'init_ck/D/{pname}'], rtol=1e-9, atol=1e-12, what=pname)
    finish_setup(experiment)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    x = torch.rand(2, 3, 224, 224, generator=generator) * 2 - 1
    u = torch.rand(2, 3, 224, 224, generator=generator) * 2 - 1
    y = torch.rand(2, generator=generator) * 85 + 10
    result = run_step(experiment, x.cuda(), y.cuda(), u.cuda(), 0, g)
    check(result, golden_scalars(g, 0), 'vgg step 0')
    assert_close(experiment.labeled_features.cpu().numpy(), g['s0/labeled_features'], rtol=RTOL, atol=1e-4,
                 what='vgg features')


def test_crowd_step_with_the_dnn_step_on_a_second_stream(pkg):
    """settings.overlap_dnn_step: the DNN step enqueued on a side stream, concurrent with the GAN step -- same
    results as the sequential order (golden g7b, two steps)."""
    test_crowd_steps(pkg, 'g7b_crowd64', 64, 2, False, overlap=True)


def test_crowd_step_at_the_benchmark_size_matches_the_oracle(pkg):
    """512 x 512 (the bench.py configuration's image size; batch 1, i.e. 3 stacked examples through D): the five
    logged losses and the post-Adam discriminator / generator weights against the CPU oracle.  Exercises every kernel
    on the benchmark's plane sizes (128 ... 16 pixels) instead of the goldens' 64 / 224."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as entry
    experiment, oracle = entry.step_against_oracle(512, 1, tag='crowd512')
    for name, ours, theirs in (('D', experiment.D, oracle.D), ('G', experiment.G, oracle.G), ('DNN', experiment.DNN, oracle.DNN)):
        reference = dict(theirs.named_parameters())
        for pname, p in ours.named_parameters():
            expected = reference[pname].detach().numpy()
            got = p.detach().cpu().numpy()
            # Adam's first step moves every element by ~lr * sign(gradient): elements whose gradient is at rounding
            # level may differ by two learning rates; the bulk must agree.
            assert np.abs(got - expected).max() <= 2.2e-4 + 1e-3 * np.abs(expected).max(), f'{name} {pname}'
            assert np.abs(got - expected).mean() <= 2e-5 + 1e-4 * np.abs(expected).mean(), f'{name} {pname} (mean)'


def _hip_and_oracle_step(experiment_class, configure, oracle_networks, size, batch, d_scale, settings_overrides=None):
    """One dnn + gan step of a task experiment on the HIP path and of the oracle composed from the same
    architecture, from identical weights (state copied HIP -> oracle), inputs and random draws."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as entry
    from srgan_amd.settings import Settings
    from srgan_amd.utility import SummaryWriter, seed_all
    from oracle.experiment import OracleExperiment, Draws
    entry._cap_host_threads(torch)
    settings = Settings()
    settings.batch_size = batch
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e2, 1e1
    settings.gradient_penalty_multiplier = 1e2
    for key, value in (settings_overrides or {}).items():
        setattr(settings, key, value)
    experiment = experiment_class(settings)
    configure(experiment)
    seed_all(0)
    experiment.model_setup()
    with torch.no_grad():                       # scale D so that the gradient penalty is active
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear)):
                module.weight.mul_(d_scale)
    oracle_g, oracle_d, oracle_dnn = oracle_networks()
    for ours, theirs in ((experiment.G, oracle_g), (experiment.D, oracle_d), (experiment.DNN, oracle_dnn)):
        theirs.load_state_dict({k: v.detach().clone() for k, v in ours.state_dict().items()}, strict=True)
    oracle = OracleExperiment(entry.settings_for_oracle(settings), oracle_d, oracle_dnn, oracle_g)
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    finish_setup(experiment)
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
    for key in ('labeled_loss', 'unlabeled_loss', 'fake_loss', 'gradient_penalty', 'generator_loss'):
        assert_close(got[key], expected[key], rtol=RTOL, atol=0.0, what=key)
    assert expected['gradient_penalty'] > 1.0
    for name, ours, theirs in (('D', experiment.D, oracle_d), ('G', experiment.G, oracle_g),
                               ('DNN', experiment.DNN, oracle_dnn)):
        reference = dict(theirs.named_parameters())
        for pname, p in ours.named_parameters():
            want, have = reference[pname].detach().numpy(), p.detach().cpu().numpy()
            # Adam's first step moves every element by ~lr * sign(gradient): rounding-level gradients may flip
            assert np.abs(have - want).max() <= 2.2e-4 + 1e-3 * np.abs(want).max(), f'{name} {pname}'
            assert np.abs(have - want).mean() <= 2e-5 + 1e-4 * np.abs(want).mean(), f'{name} {pname} (mean)'


def test_age_vgg_step_at_64_pixels_matches_the_oracle(pkg, monkeypatch):
    """BASELINE.json configs[1]: age SRGAN with the VGG-16 discriminator on 64x64 faces,
    classifier in-features 512 * (64 / 32)^2.  The 224^2 graph is pinned to the reference by g8b_vgg224; here the
    same layers at 64^2 against the oracle's VGG16(image_size=64)."""
    import srgan_amd.age.srgan as age
    from oracle import models as OM
    monkeypatch.setattr(age, 'model_architecture', 'vgg')       # module-level switch, as in reference age/srgan.py:14

    def configure(experiment):
        experiment.image_size = 64
    _hip_and_oracle_step(age.AgeExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=64), OM.VGG16(1, 64), OM.VGG16(1, 64)),
                         size=64, batch=8, d_scale=1.3)


def test_driving_step_on_rectangular_frames_matches_the_oracle(pkg):
    """BASELINE.json configs[4]: the driving DCGAN pair on 64x192 frames -- the seed
    transposed convolution and the discriminator's last convolution have the rectangular (4, 12) kernel."""
    from srgan_amd.driving.srgan import DrivingExperiment
    from oracle import models as OM
    size = (64, 192)

    def configure(experiment):
        experiment.image_size = size
    _hip_and_oracle_step(DrivingExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=size), OM.DCGANDiscriminator(image_size=size),
                                  OM.DCGANDiscriminator(image_size=size)),
                         size=size, batch=8, d_scale=2.2)


def test_training_loop_and_checkpoint_interchange(pkg, tmp_path):
    """H1 (srgan.py:52-129): ``Experiment.train()`` end to end on the coefficient task -- loop, learning-rate
    schedule, validation summaries, ``model_<step>.pth`` -- then the checkpoint loads (strictly) into the oracle's
    plain-torch modules and reproduces the predictions on the CPU, and a second experiment continues from it."""
    import os
    from srgan_amd.settings import Settings
    from srgan_amd.coefficient.srgan import CoefficientExperiment
    from srgan_amd.srgan import as_var
    from srgan_amd.tape import no_grad
    from oracle import models as OM

    def settings(steps):
        s = Settings()
        s.trial_name, s.logs_directory = 'loop', str(tmp_path)
        s.steps_to_run, s.summary_step_period = steps, 3
        s.batch_size, s.labeled_dataset_size, s.unlabeled_dataset_size, s.validation_dataset_size = 64, 128, 512, 64
        s.gradient_penalty_multiplier = 1e1
        s.skip_completed_experiment = False
        return s
    first = CoefficientExperiment(settings(6))
    first.train()
    checkpoint_path = os.path.join(first.trial_directory, 'model_6.pth')
    assert os.path.exists(checkpoint_path)
    checkpoint = torch.load(checkpoint_path, map_location='cpu')
    assert set(checkpoint) == {'DNN', 'dnn_optimizer', 'D', 'd_optimizer', 'G', 'g_optimizer', 'step'}
    assert checkpoint['d_optimizer']['state'][0]['step'] == 6 and checkpoint['step'] == 6
    tags = first.gan_summary_writer.scalars
    assert '1 Validation Error/MAE' in tags and 'Discriminator/Gradient Penalty' in tags
    # the reference-side modules accept the checkpoint as is
    oracle_d, oracle_g = OM.CoefficientMLP(10), OM.CoefficientGenerator(10)
    oracle_d.load_state_dict(checkpoint['D'], strict=True)
    oracle_g.load_state_dict(checkpoint['G'], strict=True)
    examples = torch.from_numpy(first.validation_dataset.examples.astype(np.float32))
    with no_grad():
        ours = first.D(as_var(examples)).cpu().numpy()
    assert_close(ours, oracle_d(examples).detach().numpy(), rtol=1e-4, atol=1e-5, what='D predictions from the checkpoint')
    reference_optimizer = torch.optim.Adam(oracle_d.parameters(), lr=1e-4)
    reference_optimizer.load_state_dict(checkpoint['d_optimizer'])          # torch.optim.Adam-compatible
    # continue the same trial: weights, Adam moments and the step counter come back
    resumed_settings = settings(8)
    resumed_settings.trial_name = os.path.basename(first.trial_directory)
    resumed_settings.continue_existing_experiments = True
    second = CoefficientExperiment(resumed_settings)
    second.train()
    assert second.starting_step == 7
    assert second.d_optimizer.step_count == 6 + 1            # steps 7 .. 7 (steps_to_run = 8 is exclusive)
    assert os.path.exists(os.path.join(second.trial_directory, 'model_8.pth'))


def test_crowd_sliding_window_inference(pkg):
    """``CrowdExperiment.predict_full_example`` against the reference's own function (golden g9:
    images larger than, equal to and smaller than one patch; window step 24, batches of 4)."""
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    from srgan_amd.crowd.data import CrowdExample, ImageSlidingWindowDataset
    g = load_golden('g9_crowd_sliding_window')
    size = int(g['image_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=int(g['batch_size']), image_patch_size=size, test_sliding_window_size=int(g['window_step'])),
        crowd=True)
    for pname, p in experiment.D.named_parameters():
        assert_close(checksum(p), g[f'init_ck/D/{pname}'], rtol=1e-9, atol=1e-12, what=f'init {pname}')
    finish_setup(experiment)
    experiment.eval_mode()
    for index in range(3):
        image = g[f'e{index}/image']
        example = CrowdExample(image=image, label=np.zeros(image.shape[:2], dtype=np.float32))
        count, density = experiment.predict_full_example(example, experiment.D)
        assert density.shape == image.shape[:2]
        assert_close(count, float(g[f'e{index}/count']), rtol=RTOL, what=f'example {index} count')
        assert_close(float(np.abs(density).sum()), float(g[f'e{index}/label_abs_sum']), rtol=RTOL, atol=1e-6,
                     what=f'example {index} density')
    # window bookkeeping: a 40 x 90 image with 64-pixel patches has one (padded) row of centres
    dataset = ImageSlidingWindowDataset(CrowdExample(image=g['e2/image']), size, int(g['window_step']))
    assert dataset.y_positions == [8] and dataset.x_positions == [32, 56, 58]
    patch, x, y = dataset[0]
    assert tuple(patch.shape) == (3, size, size) and float(patch.min()) >= -1.0 and float(patch.max()) <= 1.0


def test_crowd_evaluation_summaries(pkg):
    """the crowd evaluation summaries against the reference's own functions (golden g11):
    ``evaluation_epoch`` over patch batches (count ME / MAE / MSE over all batches, kNN-map errors over the first batch
    only, as the reference computes them) for DNN and D, and ``test_summaries`` over three full images through the
    sliding-window prediction; every logged scalar is compared."""
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    g = load_golden('g11_crowd_evaluation')
    size = int(g['image_size'])
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=int(g['batch_size']), image_patch_size=size, test_sliding_window_size=int(g['window_step']),
             test_summary_size=None, map_directory_name='unused'), crowd=True)
    for module, prefix in ((experiment.D, 'init_ck/D'), (experiment.DNN, 'init_ck/DNN')):
        for pname, p in module.named_parameters():
            assert_close(checksum(p), g[f'{prefix}/{pname}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {pname}')
    finish_setup(experiment)
    experiment.eval_mode()
    patches = list(zip(torch.from_numpy(g['patches/images']), torch.from_numpy(g['patches/labels']),
                       torch.from_numpy(g['patches/maps'])))
    dnn_mae = experiment.evaluation_epoch(experiment.settings, experiment.DNN, patches, experiment.dnn_summary_writer,
                                          '1 Validation Error', shuffle=False)
    experiment.evaluation_epoch(experiment.settings, experiment.D, patches, experiment.gan_summary_writer,
                                '1 Validation Error', comparison_value=dnn_mae, shuffle=False)
    scenes = [(g[f'scene{i}/image'], g[f'scene{i}/label'], None) for i in range(3)]

    class TestDataset:
        length = len(scenes)

        def __init__(self, dataset, map_directory_name):
            assert dataset == 'test'

        def __getitem__(self, index):
            return scenes[index]
    experiment.dataset_class = TestDataset
    experiment.test_summaries()
    compared = 0
    for prefix, writer in (('dnn', experiment.dnn_summary_writer), ('gan', experiment.gan_summary_writer)):
        logged = {tag: values[-1][1] for tag, values in writer.scalars.items()}
        for key in g.files:
            if key.startswith(prefix + '/'):
                tag = key[len(prefix) + 1:]
                assert_close(float(logged[tag]), float(g[key]), rtol=RTOL, atol=1e-6, what=f'{prefix} {tag}')
                compared += 1
    assert compared == 23


def test_dnn_only_experiment(pkg, tmp_path):
    """SURVEY.md 8(f) N3, the DNN method (reference dnn.py:15-100, crowd/dnn.py:20): the DNN-only loop end to end on the
    coefficient task (checkpoint with the reference's three keys), and the crowd mix-in's step equal to the DNN step of
    the full crowd experiment on identical weights."""
    import os
    from srgan_amd.settings import Settings
    from srgan_amd.coefficient.dnn import CoefficientDnnExperiment
    from srgan_amd.crowd.dnn import CrowdDnnExperiment
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    from srgan_amd.utility import SummaryWriter, seed_all
    s = Settings()
    s.trial_name, s.logs_directory, s.steps_to_run, s.summary_step_period = 'dnn', str(tmp_path), 5, 2
    s.batch_size, s.labeled_dataset_size, s.unlabeled_dataset_size, s.validation_dataset_size = 64, 128, 512, 64
    s.skip_completed_experiment = False
    experiment = CoefficientDnnExperiment(s)
    experiment.train()
    checkpoint = torch.load(os.path.join(experiment.trial_directory, 'model_5.pth'), map_location='cpu')
    assert set(checkpoint) == {'DNN', 'dnn_optimizer', 'step'}
    assert experiment.D is None and experiment.G is None
    assert '1 Validation Error/MAE' in experiment.dnn_summary_writer.scalars
    losses = [v for _, v in experiment.dnn_summary_writer.scalars['Discriminator/Labeled Loss']]
    assert len(losses) >= 2 and all(np.isfinite(losses))
    # crowd: same DNN weights, same batch -> same loss and same updated weights as the full experiment's DNN step
    size = 64
    full = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=2, map_multiplier=1e-3), crowd=True)
    crowd_settings = Settings()
    crowd_settings.batch_size, crowd_settings.image_patch_size, crowd_settings.map_multiplier = 2, size, 1e-3
    alone = CrowdDnnExperiment(crowd_settings)
    seed_all(0)
    alone.model_setup()
    alone.DNN.load_state_dict(full.DNN.state_dict())
    alone.dnn_summary_writer = SummaryWriter()
    finish_setup(full)
    alone.prepare_optimizers()
    alone.train_mode()
    generator = torch.Generator().manual_seed(3)
    x, y, _ = crowd_inputs(generator, 2, size)
    full.dnn_training_step(x.cuda(), tuple(t.cuda() for t in y), 0)
    alone.dnn_training_step(x.cuda(), tuple(t.cuda() for t in y), 0)
    assert_close(alone.dnn_summary_writer.scalars['Discriminator/Labeled Loss'][-1][1],
                 full.dnn_summary_writer.scalars['Discriminator/Labeled Loss'][-1][1], rtol=1e-6, what='DNN loss')
    for (name, ours), (_, theirs) in zip(alone.DNN.named_parameters(), full.DNN.named_parameters()):
        assert_close(ours.detach().cpu().numpy(), theirs.detach().cpu().numpy(), rtol=1e-5, atol=2.2e-4, what=name)


def test_regression_evaluation_epochs(pkg):
    """the age / driving evaluation epochs (forward passes only): MAE / MSE / NMAE / the GAN-to-DNN
    ratio as summary scalars, checked against the oracle's plain-torch discriminator on the CPU."""
    from srgan_amd.settings import Settings
    from srgan_amd.driving.srgan import DrivingExperiment
    from srgan_amd.utility import SummaryWriter, seed_all
    from oracle import models as OM
    s = Settings()
    s.batch_size = 8
    experiment = DrivingExperiment(s)
    experiment.image_size = 64
    seed_all(0)
    experiment.dataset_setup()
    experiment.model_setup()
    finish_setup(experiment)
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    experiment.eval_mode()
    experiment.validation_summaries(0)
    gan, dnn = experiment.gan_summary_writer.scalars, experiment.dnn_summary_writer.scalars
    for tag in ('1 Validation Error/MAE', '1 Validation Error/NMAE', '1 Validation Error/MSE', '2 Train Error/MAE'):
        assert tag in gan and tag in dnn, tag
    oracle_d = OM.DCGANDiscriminator(64)
    oracle_d.load_state_dict(experiment.D.state_dict())
    images, angles = experiment.validation_dataset_loader.batches[0]
    expected = float((oracle_d(images.cpu()).detach().reshape(-1) - angles.cpu()).abs().mean())
    assert_close(gan['1 Validation Error/MAE'][-1][1], expected, rtol=RTOL, what='D validation MAE')
    ratio = gan['1 Validation Error/MAE'][-1][1] / dnn['1 Validation Error/MAE'][-1][1]
    assert_close(gan['1 Validation Error/Ratio MAE GAN DNN'][-1][1], ratio, rtol=1e-9, what='ratio')


@pytest.mark.parametrize('reference_schedule', [False, True])
def test_coefficient_dggan(pkg, reference_schedule):
    """the dual-goal GAN step (reference coefficient/dggan.py:13-64) against golden g4b (gradient
    penalty active on the per-example scores): logged losses, gradient norms, post-Adam weights."""
    from srgan_amd.coefficient.dggan import CoefficientDgganExperiment
    from srgan_amd.coefficient.models import DgganMLP, Generator
    g = load_golden('g4b_coefficient_dggan_gp_active')

    class _Experiment(CoefficientDgganExperiment):
        def dataset_setup(self):
            pass

        def validation_summaries(self, step):
            pass
    from srgan_amd.settings import Settings
    from srgan_amd.utility import SummaryWriter, seed_all
    settings = Settings()
    settings.batch_size, settings.gradient_penalty_multiplier = int(g['batch_size']), 1e1
    settings.reference_schedule = reference_schedule
    experiment = _Experiment(settings)
    seed_all(0)
    experiment.model_setup()
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN'), (experiment.G, 'init/G')):
        module.load_state_dict(golden_state(g, prefix))
    finish_setup(experiment)
    for step in range(int(g['steps'])):
        x, y, u = (dev(g[f's{step}/{k}']) for k in ('x', 'y', 'u'))
        result = run_step(experiment, x, y, u, step, g)
        check(result, golden_scalars(g, step), f'dggan step {step}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, atol=1e-6,
                     what='gradient_norm')
    assert float(g['s0/gradient_penalty']) > 1.0
    for name, module in (('D', experiment.D), ('G', experiment.G), ('DNN', experiment.DNN)):
        for key, value in golden_state(g, f'final/{name}').items():
            assert_close(module.state_dict()[key].cpu().numpy(), value.numpy(), rtol=RTOL, atol=2e-5,
                         what=f'final {name} {key}')


def test_crowd_step_fed_by_the_device_patch_loader(pkg):
    """batches cut on the device from resident full scenes satisfy the crowd batch contract -- one
    full iteration (DNN step + GAN step) runs on them and yields finite losses."""
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    from srgan_amd.crowd.data import CrowdExample, DeviceCrowdPatchLoader
    size, batch = 64, 2
    generator = np.random.RandomState(2)
    scenes = [CrowdExample(image=generator.randint(0, 256, size=(96, 128, 3)).astype(np.uint8),
                           label=(generator.rand(96, 128) < 0.003).astype(np.float32),
                           map_=generator.rand(96, 128).astype(np.float32)) for _ in range(3)]
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2,
             gradient_penalty_multiplier=1e2, map_multiplier=1e-3), crowd=True)
    finish_setup(experiment)
    labeled = iter(DeviceCrowdPatchLoader(scenes, batch, size, seed=0))
    unlabeled = iter(DeviceCrowdPatchLoader(scenes, batch, size, seed=100))
    x, heads, knn = next(labeled)
    u = next(unlabeled)[0]
    experiment.dnn_training_step(x, (heads, knn), 0)
    experiment.gan_training_step(x, (heads, knn), u, 0)
    values = [v[-1][1] for v in experiment.gan_summary_writer.scalars.values()]
    assert len(values) >= 6 and all(np.isfinite(values))


def test_crowd_experiment_on_a_preprocessed_database(pkg, tmp_path, monkeypatch):
    """SURVEY.md 8(f) N4 + N2 end to end: with ``SRGAN_CROWD_DATABASE`` pointing at a database in the reference's
    preprocessed layout, ``CrowdExperiment.dataset_setup`` keeps the training scenes resident and cuts batches on the
    device; one iteration runs on them and the full-image test summaries walk the test split."""
    import os
    from srgan_amd.settings import Settings
    from srgan_amd.crowd.srgan import CrowdExperiment
    from srgan_amd.utility import SummaryWriter, seed_all
    generator = np.random.RandomState(4)
    for split, count in (('train', 3), ('test', 2)):
        for index in range(count):
            shape = (80 + 8 * index, 100)
            arrays = {'images': generator.randint(0, 256, size=shape + (3,)).astype(np.uint8),
                      'labels': (generator.rand(*shape) < 0.004).astype(np.float32),
                      'i1nn_maps': generator.rand(*shape).astype(np.float32)}
            for directory, array in arrays.items():
                os.makedirs(tmp_path / 'part_A' / f'{split}_data' / directory, exist_ok=True)
                np.save(tmp_path / 'part_A' / f'{split}_data' / directory / f'IMG_{index}.npy', array)
    monkeypatch.setenv('SRGAN_CROWD_DATABASE', str(tmp_path))
    monkeypatch.setenv('SRGAN_CROWD_DATABASE_PART', 'part_A')
    settings = Settings()
    settings.batch_size, settings.image_patch_size, settings.test_sliding_window_size = 2, 64, 32
    settings.labeled_dataset_size, settings.unlabeled_dataset_size, settings.test_summary_size = 2, 3, None
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e3, 1e2
    settings.gradient_penalty_multiplier, settings.map_multiplier = 1e2, 1e-3
    experiment = CrowdExperiment(settings)
    experiment.dataset_setup()
    assert len(experiment.train_dataset_loader.images) == 2 and len(experiment.unlabeled_dataset_loader.images) == 3
    seed_all(0)
    experiment.model_setup()
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    finish_setup(experiment)
    x, heads, knn = next(iter(experiment.train_dataset_loader))
    u = next(iter(experiment.unlabeled_dataset_loader))[0]
    assert tuple(x.shape) == (2, 3, 64, 64) and tuple(heads.shape) == (2, 64, 64) == tuple(knn.shape)
    experiment.dnn_training_step(x, (heads, knn), 0)
    experiment.gan_training_step(x, (heads, knn), u, 0)
    experiment.eval_mode()
    experiment.validation_summaries(0)
    logged = {tag: values[-1][1] for tag, values in experiment.gan_summary_writer.scalars.items()}
    for tag in ('0 Test Error/MAE count', '0 Test Error/RMSE count', '0 Test Error/Ratio MAE GAN DNN',
                'Discriminator/Gradient Penalty'):
        assert np.isfinite(logged[tag]), tag
