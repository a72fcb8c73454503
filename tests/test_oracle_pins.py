"""Pins the CPU oracle against the golden fixtures generated from the unmodified reference
(tests/golden/make_goldens.py).  CPU-only; this is what makes the oracle trustworthy as the checker."""
import numpy as np
import pytest
import torch

from oracle import functional as OF
from oracle import models as OM
from oracle.experiment import OracleExperiment, OracleSganExperiment, Draws, freeze_batch_norm
from helpers import load_golden, golden_state, golden_scalars, make_settings, checksum, assert_close

TIGHT = dict(rtol=2e-6, atol=1e-7)   # same arithmetic (torch CPU) -> expect (near) bitwise agreement


def test_golden_torch_version_matches():
    assert str(load_golden('g1_distance')['torch_version']) == torch.__version__


def test_toy_data_and_mixture():
    g = load_golden('g0_toydata')
    examples, labels = OF.toy_dataset(int(g['size']), int(g['seed']))
    np.testing.assert_array_equal(examples, g['examples'])
    np.testing.assert_array_equal(labels, g['labels'])
    from scipy.stats import norm
    OF.seed_all(int(g['mixture_seed']))
    offset = float(g['mixture_offset'])
    np.testing.assert_array_equal(OF.mixture_rvs([norm(-offset, 1), norm(offset, 1)], size=[6, 5]), g['mixture_rvs'])


def test_distance_functions():
    g = load_golden('g1_distance')
    d = torch.from_numpy(g['d'])
    for name, fn in list(OF.DISTANCES.items()) + [('abs_plus_one_square_root', OF.abs_plus_one_square_root)]:
        np.testing.assert_allclose(fn(d).numpy(), g[name], rtol=1e-6, err_msg=name)
    base = torch.from_numpy(g['base']).requires_grad_()
    other = torch.from_numpy(g['other']).requires_grad_()
    np.testing.assert_allclose(OF.feature_distance_loss(base, other, OF.abs_mean).item(), g['fdl_default'], rtol=1e-6)
    loss = OF.feature_distance_loss(base, other, OF.abs_plus_one_sqrt_mean_neg)
    np.testing.assert_allclose(loss.item(), g['fdl_contrasting'], rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(base.grad.numpy(), g['fdl_contrasting_grad_base'], rtol=1e-6)
    np.testing.assert_allclose(other.grad.numpy(), g['fdl_contrasting_grad_other'], rtol=1e-6)
    p, y = torch.from_numpy(g['ll_p']), torch.from_numpy(g['ll_y'])
    np.testing.assert_allclose(OF.labeled_loss(p, y, 2).item(), g['ll_order2'], rtol=1e-6)
    np.testing.assert_allclose(OF.labeled_loss(p, y, 1).item(), g['ll_order1'], rtol=1e-6)


def test_sgan_math():
    g = load_golden('g2_sgan_math')
    logits = torch.from_numpy(g['logits'])
    np.testing.assert_allclose(OF.logsumexp(logits, dim=1).numpy(), g['lse_dim1'], rtol=1e-6)
    np.testing.assert_allclose(OF.logsumexp(logits).numpy(), g['lse_all'], rtol=1e-6)
    np.testing.assert_allclose(OF.logsumexp(logits, dim=1, keepdim=True).numpy(), g['lse_keepdim'], rtol=1e-6)
    bins, reals = torch.from_numpy(g['bins']), torch.from_numpy(g['reals'])
    np.testing.assert_array_equal(OF.real_numbers_to_bin_indexes(reals, bins).numpy(), g['bin_indexes'])
    np.testing.assert_allclose(OF.logits_to_bin_values(logits, bins).numpy(), g['bin_values'])


def replay(experiment, g, steps, crowd=False):
    """Feed the recorded batches and draws to the oracle; return per-step scalar dicts."""
    results = []
    for step in range(steps):
        x, u = torch.from_numpy(g[f's{step}/x']), torch.from_numpy(g[f's{step}/u'])
        y = torch.from_numpy(g[f's{step}/y'])
        experiment.dnn_training_step(x, y)
        draws = Draws(*(torch.from_numpy(g[f's{step}/{k}']) for k in ('z_d', 'z_g', 'alpha')))
        result = experiment.gan_training_step(x, y, u, step, draws)
        result['dnn_loss'] = experiment.scalars['dnn_loss']
        results.append(result)
    return results


def check_scalars(results, g, **tol):
    for step, result in enumerate(results):
        expected = golden_scalars(g, step)
        for key, value in result.items():
            assert_close(value, expected[key], what=f'step {step} {key}', **tol)


def build_coefficient(g, sgan=False):
    settings = make_settings(batch_size=int(g['batch_size']))
    if sgan:
        D, DNN, G = OM.coefficient_sgan_mlp(10), OM.coefficient_sgan_mlp(10), OM.CoefficientGenerator()
    else:
        D, DNN, G = OM.CoefficientMLP(10), OM.CoefficientMLP(10), OM.CoefficientGenerator(10)
    for module, prefix in ((D, 'init/D'), (DNN, 'init/DNN'), (G, 'init/G')):
        module.load_state_dict(golden_state(g, prefix))
    if sgan:
        return OracleSganExperiment(settings, D, DNN, G, bins=torch.linspace(-3, 3, 10))
    return OracleExperiment(settings, D, DNN, G)


@pytest.mark.parametrize('name,steps', [('g3_coefficient_srgan', 3), ('g3b_coefficient_srgan_gp_active', 2)])
def test_coefficient_srgan_steps(name, steps):
    g = load_golden(name)
    experiment = build_coefficient(g)
    results = replay(experiment, g, steps)
    check_scalars(results, g, **TIGHT)
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].numpy(), value.numpy(), what=f'final D {key}', **TIGHT)
    for key, value in golden_state(g, 'final/G').items():
        assert_close(experiment.G.state_dict()[key].numpy(), value.numpy(), what=f'final G {key}', **TIGHT)
    assert_close(experiment.gradient_norm.detach().numpy(), g[f's{steps - 1}/gradient_norm'], what='gn', **TIGHT)


def test_coefficient_first_step_gradients():
    g = load_golden('g3b_coefficient_srgan_gp_active')
    experiment = build_coefficient(g)
    replay(experiment, g, 1)
    for key, value in experiment.d_grads.items():
        assert_close(value.numpy(), g[f's0/d_grad/{key}'], rtol=1e-5, atol=1e-7, what=f'd_grad {key}')


@pytest.mark.parametrize('name', ['g4_coefficient_sgan', 'g4c_coefficient_sgan_gp_active'])
def test_coefficient_sgan_steps(name):
    g = load_golden(name)
    experiment = build_coefficient(g, sgan=True)
    results = replay(experiment, g, 2)
    check_scalars(results, g, rtol=1e-5, atol=1e-7)
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].numpy(), value.numpy(), rtol=1e-5, atol=1e-7, what=key)


def test_tiny_dcgan_steps_with_active_gradient_penalty():
    g = load_golden('g5_tiny_dcgan')
    settings = make_settings(batch_size=4, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                             gradient_penalty_multiplier=1e2)
    D, DNN = OM.DCGANDiscriminator(32, 8), OM.DCGANDiscriminator(32, 8)
    G = OM.DCGANGenerator(image_size=32, conv_dim=8)
    # Fresh constructors must already reproduce the reference's initial G weights (same RNG stream).
    for key, value in golden_state(g, 'init/G').items():
        np.testing.assert_array_equal(G.state_dict()[key].numpy(), value.numpy())
    for module, prefix in ((D, 'init/D'), (DNN, 'init/DNN')):
        module.load_state_dict(golden_state(g, prefix))
    experiment = OracleExperiment(settings, D, DNN, G)
    results = replay(experiment, g, 2)
    assert results[0]['gradient_penalty'] > 100.0
    check_scalars(results, g, rtol=1e-5, atol=1e-6)
    for key, value in golden_state(g, 'final/D').items():
        assert_close(experiment.D.state_dict()[key].numpy(), value.numpy(), rtol=1e-5, atol=1e-7, what=key)


def test_layer_kats():
    g = load_golden('g6_layers')
    modules = {'dense_layer': OM.DenseLayer(16, 8, 4), 'dense_block': OM.dense_block(3, 8, 2, 4),
               'transition': OM.transition(32, 16), 'map_module': OM.MapModule(16, 4, 32)}
    for prefix, module in modules.items():
        module.load_state_dict(golden_state(g, f'{prefix}/state'))
        module.train()
        module.apply(freeze_batch_norm)
        x = torch.from_numpy(g[f'{prefix}/x']).requires_grad_()
        y = module(x)
        ys = list(y) if isinstance(y, tuple) else [y]
        scalar = 0
        for i, t in enumerate(ys):
            assert_close(t.detach().numpy(), g[f'{prefix}/y{i}'], what=f'{prefix} y{i}', **TIGHT)
            scalar = scalar + (t * torch.from_numpy(g[f'{prefix}/c{i}'])).sum()
        (gx,) = torch.autograd.grad(scalar, x, create_graph=True)
        assert_close(gx.detach().numpy(), g[f'{prefix}/gx'], rtol=1e-5, atol=1e-6, what=f'{prefix} gx')
        penalty = (gx.reshape(gx.shape[0], -1).norm(dim=1) ** 2).mean()
        assert_close(penalty.item(), g[f'{prefix}/penalty'], rtol=1e-5, what=f'{prefix} penalty')


def crowd_inputs(generator, batch, size):
    x = torch.rand(batch, 3, size, size, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, size, size, generator=generator) * 2 - 1
    heads = (torch.rand(batch, size, size, generator=generator) < 0.002).float()
    knn_map = torch.rand(batch, size, size, generator=generator)
    return x, (heads, knn_map), u


def build_crowd(g, size):
    OF.seed_all(0)
    G = OM.DCGANGenerator(image_size=size)
    D, DNN = OM.KnnDenseNetCat(image_size=size), OM.KnnDenseNetCat(image_size=size)
    scale = float(g['d_scale'])
    if scale != 1.0:
        with torch.no_grad():
            for m in D.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    m.weight.mul_(scale)
    return D, DNN, G


def crowd_experiment(g, size):
    from functools import partial
    D, DNN, G = build_crowd(g, size)
    for module, prefix in ((D, 'init_ck/D'), (DNN, 'init_ck/DNN'), (G, 'init_ck/G')):
        for name, p in module.named_parameters():
            assert_close(checksum(p), g[f'{prefix}/{name}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {name}')
    settings = make_settings(batch_size=int(g['batch_size']), matching_loss_multiplier=1e3,
                             contrasting_loss_multiplier=1e2, gradient_penalty_multiplier=1e2, map_multiplier=1e-3)
    return OracleExperiment(settings, D, DNN, G,
                            labeled_loss_function=lambda p, y, order: OF.crowd_labeled_loss(p, y, order, 1e-3))


@pytest.mark.parametrize('name,size,steps', [('g7b_crowd64', 64, 2), ('g7c_crowd64_gp_active', 64, 1),
                                             ('g7_crowd224', 224, 1)])
def test_crowd_steps(name, size, steps):
    g = load_golden(name)
    experiment = crowd_experiment(g, size)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    batches = [crowd_inputs(generator, int(g['batch_size']), size) for _ in range(steps)]
    experiment.D.apply(freeze_batch_norm)
    with torch.no_grad():
        density, count, maps = experiment.D(batches[0][0])
    assert_close(count.numpy(), g['fwd/count'], rtol=1e-5, what='count')
    assert_close(experiment.D.features.numpy(), g['fwd/features'], rtol=1e-5, atol=1e-7, what='features')
    assert_close(checksum(maps)[:2], g['fwd/maps_ck'], rtol=1e-6, what='maps checksum')
    for step, (x, y, u) in enumerate(batches):
        experiment.dnn_training_step(x, y)
        draws = Draws(*(torch.from_numpy(g[f's{step}/{k}']) for k in ('z_d', 'z_g', 'alpha')))
        result = experiment.gan_training_step(x, y, u, step, draws)
        result['dnn_loss'] = experiment.scalars['dnn_loss']
        expected = golden_scalars(g, step)
        for key, value in result.items():
            assert_close(value, expected[key], rtol=2e-5, atol=1e-6, what=f'{name} step {step} {key}')
    if name == 'g7c_crowd64_gp_active':
        assert result['gradient_penalty'] > 10.0


def test_age_dcgan128_steps():
    g = load_golden('g8_age_dcgan128')
    settings = make_settings(batch_size=4, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                             gradient_penalty_multiplier=1e2)
    OF.seed_all(0)
    G, D, DNN = OM.DCGANGenerator(), OM.DCGANDiscriminator(), OM.DCGANDiscriminator()
    for name, p in D.named_parameters():
        assert_close(checksum(p), g[f'init_ck/D/{name}'], rtol=1e-9, atol=1e-12, what=name)
    experiment = OracleExperiment(settings, D, DNN, G)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    for step in range(2):
        x = torch.rand(4, 3, 128, 128, generator=generator) * 2 - 1
        u = torch.rand(4, 3, 128, 128, generator=generator) * 2 - 1
        y = torch.rand(4, generator=generator) * 85 + 10
        experiment.dnn_training_step(x, y)
        draws = Draws(*(torch.from_numpy(g[f's{step}/{k}']) for k in ('z_d', 'z_g', 'alpha')))
        result = experiment.gan_training_step(x, y, u, step, draws)
        expected = golden_scalars(g, step)
        for key, value in result.items():
            assert_close(value, expected[key], rtol=2e-5, atol=1e-6, what=f'age step {step} {key}')


def test_vgg224_step():
    g = load_golden('g8b_vgg224')
    settings = make_settings(batch_size=2, matching_loss_multiplier=1e2, contrasting_loss_multiplier=1e1,
                             gradient_penalty_multiplier=1e2)
    OF.seed_all(0)
    G = OM.DCGANGenerator(image_size=224)
    D, DNN = OM.VGG16(num_classes=1), OM.VGG16(num_classes=1)
    for name, p in D.named_parameters():
        assert_close(checksum(p), g[f'init_ck/D/{name}'], rtol=1e-9, atol=1e-12, what=name)
    experiment = OracleExperiment(settings, D, DNN, G)
    generator = torch.Generator().manual_seed(int(g['input_seed']))
    x = torch.rand(2, 3, 224, 224, generator=generator) * 2 - 1
    u = torch.rand(2, 3, 224, 224, generator=generator) * 2 - 1
    y = torch.rand(2, generator=generator) * 85 + 10
    experiment.dnn_training_step(x, y)
    draws = Draws(*(torch.from_numpy(g[f's0/{k}']) for k in ('z_d', 'z_g', 'alpha')))
    result = experiment.gan_training_step(x, y, u, 0, draws)
    expected = golden_scalars(g, 0)
    for key, value in result.items():
        assert_close(value, expected[key], rtol=2e-5, atol=1e-6, what=f'vgg {key}')
    assert_close(experiment.labeled_features.detach().numpy(), g['s0/labeled_features'], rtol=1e-4, atol=1e-6,
                 what='vgg features')
