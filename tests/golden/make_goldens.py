"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED upstream reference on CPU.

Test infrastructure only. Runs in the build container (needs /root/reference, which never travels
to the GPU box); the emitted ``*.npz`` files are data: inputs, RNG draws, initial weights (or seeds +
checksums) and the reference's outputs. Usage::

    python tests/golden/make_goldens.py [g0 g1 ... | all]

Every fixture records ``torch_version`` so a drift of the third-party arithmetic (torch is unpinned
upstream, requirements.txt:12) is visible.
"""
import os
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstubs  # noqa: E402

_refstubs.install()

import torch  # noqa: E402
import srgan as ref_srgan  # noqa: E402
import utility as ref_utility  # noqa: E402
from settings import Settings  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    arrays['torch_version'] = np.array(torch.__version__)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays)')


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def state_arrays(prefix, module):
    return {f'{prefix}/{k}': v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grad_arrays(prefix, module):
    return {f'{prefix}/{k}': p.grad.detach().cpu().numpy().copy() for k, p in module.named_parameters()
            if p.grad is not None}


def adam_arrays(prefix, module, optimizer):
    out = {}
    for name, p in module.named_parameters():
        st = optimizer.state[p]
        out[f'{prefix}/{name}/exp_avg'] = st['exp_avg'].detach().cpu().numpy().copy()
        out[f'{prefix}/{name}/exp_avg_sq'] = st['exp_avg_sq'].detach().cpu().numpy().copy()
        out[f'{prefix}/{name}/step'] = np.array(float(st['step']))
    return out


def checksum_arrays(prefix, module, grads=False):
    """Per-tensor (sum, abs-sum, first, last) in float64: a compact pin for full-size networks."""
    out = {}
    for name, p in module.named_parameters():
        t = p.grad if grads else p
        if t is None:
            continue
        t64 = t.detach().double().reshape(-1)
        out[f'{prefix}/{name}'] = np.array([t64.sum().item(), t64.abs().sum().item(), t64[0].item(), t64[-1].item()])
    return out


class Recorder:
    """Captures the random draws of one gan_training_step and the D grads just before d_optimizer.step()."""

    def __init__(self, experiment):
        self.experiment = experiment
        self.z_d = None
        self.z_g = None
        self.alpha = None
        self.d_grads = None

    @contextlib.contextmanager
    def recording(self):
        recorder = self
        original_mixture = ref_srgan.MixtureModel
        original_randn, original_rand = torch.randn, torch.rand

        class RecordingMixture(original_mixture):
            def rvs(self, size):
                values = super().rvs(size)
                recorder.z_d = np.asarray(values).astype(np.float32)
                return values

        def randn(*args, **kwargs):
            value = original_randn(*args, **kwargs)
            recorder.z_g = value.detach().cpu().numpy().copy()
            return value

        def rand(*args, **kwargs):
            value = original_rand(*args, **kwargs)
            recorder.alpha = value.detach().cpu().numpy().copy()
            return value

        d_optimizer = self.experiment.d_optimizer
        original_step = d_optimizer.step

        def step(*args, **kwargs):
            recorder.d_grads = grad_arrays('d_grad', recorder.experiment.D)
            return original_step(*args, **kwargs)

        ref_srgan.MixtureModel = RecordingMixture
        torch.randn, torch.rand = randn, rand
        d_optimizer.step = step
        try:
            yield self
        finally:
            ref_srgan.MixtureModel = original_mixture
            torch.randn, torch.rand = original_randn, original_rand
            d_optimizer.step = original_step


def attach_writers(experiment):
    experiment.dnn_summary_writer = ref_utility.SummaryWriter()
    experiment.gan_summary_writer = ref_utility.SummaryWriter()
    experiment.dnn_summary_writer.summary_period = 1
    experiment.gan_summary_writer.summary_period = 1


def last_scalars(writer):
    return {tag: values[-1][1] for tag, values in writer.scalars.items()}


GAN_TAGS = {'Generator/Loss': 'generator_loss', 'Discriminator/Labeled Loss': 'labeled_loss',
            'Discriminator/Unlabeled Loss': 'unlabeled_loss', 'Discriminator/Fake Loss': 'fake_loss',
            'Discriminator/Gradient Penalty': 'gradient_penalty', 'Discriminator/Gradient Norm': 'gradient_norm_mean',
            'Feature Norm/Labeled': 'feature_norm_labeled', 'Feature Norm/Unlabeled': 'feature_norm_unlabeled'}


def run_recorded_steps(experiment, batches, out, with_grads_on_step0=True, features=True):
    """Run dnn_training_step + gan_training_step over ``batches`` and record everything into ``out``."""
    for step, (x, y, u) in enumerate(batches):
        experiment.dnn_training_step(x, y, step)
        out[f's{step}/dnn_loss'] = np.array(last_scalars(experiment.dnn_summary_writer)['Discriminator/Labeled Loss'])
        recorder = Recorder(experiment)
        with recorder.recording():
            experiment.gan_training_step(x, y, u, step)
        scalars = last_scalars(experiment.gan_summary_writer)
        for tag, key in GAN_TAGS.items():
            if tag in scalars:
                out[f's{step}/{key}'] = np.array(scalars[tag])
        out[f's{step}/z_d'] = recorder.z_d
        out[f's{step}/z_g'] = recorder.z_g
        out[f's{step}/alpha'] = recorder.alpha
        if experiment.gradient_norm is not None:
            out[f's{step}/gradient_norm'] = np32(experiment.gradient_norm)
        if features:
            for name in ('labeled_features', 'unlabeled_features', 'fake_features', 'interpolates_features'):
                value = getattr(experiment, name)
                if value is not None:
                    out[f's{step}/{name}'] = np32(value)
        if with_grads_on_step0 and step == 0:
            out.update({f's0/{k}': v for k, v in recorder.d_grads.items()})
            out.update({f's0/{k}': v for k, v in grad_arrays('g_grad', experiment.G).items()})
            out.update({f's0/{k}': v for k, v in grad_arrays('dnn_grad', experiment.DNN).items()})


# ----------------------------------------------------------------------------------------------- g0
def g0_toydata():
    """Pins the polynomial generator (coefficient/data.py:31-67) and MixtureModel.rvs (utility.py:102-107)."""
    from coefficient.data import ToyDataset
    from scipy.stats import norm
    settings = Settings()
    settings.batch_size = 16
    dataset = ToyDataset(dataset_size=64, observation_count=10, settings=settings, seed=7)
    out = {'examples': dataset.examples, 'labels': dataset.labels, 'seed': np.array(7), 'size': np.array(64)}
    ref_utility.seed_all(3)
    out['mixture_seed'] = np.array(3)
    out['mixture_offset'] = np.array(0.5)
    out['mixture_rvs'] = ref_utility.MixtureModel([norm(-0.5, 1), norm(0.5, 1)]).rvs(size=[6, 5])
    save('g0_toydata', **out)


# ----------------------------------------------------------------------------------------------- g1
def g1_distance():
    """Distance-function KATs (utility.py:201-243) and feature_distance_loss (srgan.py:438-449)."""
    generator = torch.Generator().manual_seed(11)
    d = torch.randn(37, generator=generator)
    d[3] = 0.0
    d[5] = -2.5
    out = {'d': np32(d)}
    for name in ('abs_plus_one_log_neg', 'abs_plus_one_log_mean_neg', 'abs_plus_one_sqrt_mean_neg', 'abs_mean_neg',
                 'abs_mean', 'norm_mean', 'square_mean', 'abs_plus_one_square_root'):
        out[name] = np32(getattr(ref_utility, name)(d))
    base = torch.randn(5, 37, generator=generator)
    other = torch.randn(5, 37, generator=generator) * 2 + 0.3
    out['base'], out['other'] = np32(base), np32(other)

    class _E(ref_srgan.Experiment):
        def dataset_setup(self): pass
        def model_setup(self): pass
        def validation_summaries(self, step): pass

    experiment = _E(Settings())
    out['fdl_default'] = np32(experiment.feature_distance_loss(base, other))
    out['fdl_contrasting'] = np32(experiment.feature_distance_loss(
        base, other, distance_function=experiment.settings.contrasting_distance_function))
    base_g = base.clone().requires_grad_()
    other_g = other.clone().requires_grad_()
    experiment.feature_distance_loss(base_g, other_g,
                                     distance_function=ref_utility.abs_plus_one_sqrt_mean_neg).backward()
    out['fdl_contrasting_grad_base'], out['fdl_contrasting_grad_other'] = np32(base_g.grad), np32(other_g.grad)
    # The labeled loss (srgan.py:414-417).
    p, y = torch.randn(9, generator=generator), torch.randn(9, generator=generator)
    out['ll_p'], out['ll_y'] = np32(p), np32(y)
    out['ll_order2'] = np32(experiment.labeled_loss_function(p, y, order=2))
    out['ll_order1'] = np32(experiment.labeled_loss_function(p, y, order=1))
    save('g1_distance', **out)


# ----------------------------------------------------------------------------------------------- g2
def g2_sgan_math():
    """logsumexp / bin helpers (utility.py:141-151,161-182) and the SGAN criteria (sgan.py:14-67)."""
    generator = torch.Generator().manual_seed(12)
    logits = torch.randn(6, 10, generator=generator) * 3
    logits[2, 4] = 40.0  # stability case
    out = {'logits': np32(logits),
           'lse_dim1': np32(ref_utility.logsumexp(logits, dim=1)),
           'lse_all': np32(ref_utility.logsumexp(logits)),
           'lse_keepdim': np32(ref_utility.logsumexp(logits, dim=1, keepdim=True))}
    bins = torch.linspace(-3, 3, 10)
    reals = torch.tensor([-5.0, -3.0, -2.66, -0.34, 0.0, 0.33, 0.34, 2.9, 7.0])
    out['bins'], out['reals'] = np32(bins), np32(reals)
    out['bin_indexes'] = ref_utility.real_numbers_to_bin_indexes(reals, bins).numpy()
    out['bin_values'] = np32(ref_utility.logits_to_bin_values(logits, bins))
    labels = ref_utility.real_numbers_to_bin_indexes(reals[:6], bins)
    out['ce'] = np32(torch.nn.CrossEntropyLoss()(logits, labels))
    lse = ref_utility.logsumexp(logits, dim=1)
    out['bce_ones'] = np32(torch.nn.BCEWithLogitsLoss()(lse, torch.ones_like(lse)))
    out['bce_zeros'] = np32(torch.nn.BCEWithLogitsLoss()(lse, torch.zeros_like(lse)))
    save('g2_sgan_math', **out)


# ----------------------------------------------------------------------------------------------- g3/g4
def _coefficient(experiment_class, name, steps, batch_size=256, seed_offset=0, d_scale=3.0):
    settings = Settings()
    settings.batch_size = batch_size
    settings.labeled_dataset_size = 2 * batch_size
    settings.unlabeled_dataset_size = 4 * batch_size
    settings.validation_dataset_size = batch_size
    settings.pin_memory = False
    settings.gradient_penalty_multiplier = 1e1
    settings.number_of_data_workers = 0
    experiment = experiment_class(settings)
    ref_utility.seed_all(0)
    experiment.dataset_setup()
    experiment.model_setup()
    experiment.prepare_optimizers()
    experiment.train_mode()
    attach_writers(experiment)
    out = {'batch_size': np.array(batch_size), 'steps': np.array(steps),
           'hidden_size': np.array(settings.hidden_size)}
    out.update(state_arrays('init/D', experiment.D))
    out.update(state_arrays('init/DNN', experiment.DNN))
    out.update(state_arrays('init/G', experiment.G))
    if seed_offset:
        # Scale the discriminator so that the gradient penalty is active (SURVEY.md §8c constraint 3).
        with torch.no_grad():
            for p in experiment.D.parameters():
                p.mul_(d_scale)
        out.update(state_arrays('init/D', experiment.D))
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
    batches = []
    for step in range(steps):
        x, y = next(labeled)
        u = next(unlabeled)[0]
        batches.append((x, y, u))
        out[f's{step}/x'], out[f's{step}/y'], out[f's{step}/u'] = x.numpy(), y.numpy(), u.numpy()
    run_recorded_steps(experiment, batches, out, features=(name.endswith('srgan') or 'gp' in name))
    out.update(state_arrays('final/D', experiment.D))
    out.update(state_arrays('final/DNN', experiment.DNN))
    out.update(state_arrays('final/G', experiment.G))
    out.update(adam_arrays('final_adam/D', experiment.D, experiment.d_optimizer))
    save(name, **out)


def g3_coefficient_srgan():
    from coefficient.srgan import CoefficientExperiment
    _coefficient(CoefficientExperiment, 'g3_coefficient_srgan', steps=3)
    _coefficient(CoefficientExperiment, 'g3b_coefficient_srgan_gp_active', steps=2, batch_size=64, seed_offset=1)


def g4_coefficient_sgan():
    from coefficient.sgan import CoefficientSganExperiment
    _coefficient(CoefficientSganExperiment, 'g4_coefficient_sgan', steps=2, batch_size=64)
    # the SGAN penalty differentiates a batch-MEAN scalar (sgan.py:58), so per-example gradient norms are 1/B of the
    # SRGAN ones: a much larger discriminator scale is needed before any of them exceeds 1
    _coefficient(CoefficientSganExperiment, 'g4c_coefficient_sgan_gp_active', steps=2, batch_size=64, seed_offset=1,
                 d_scale=8.0)


def g4b_coefficient_dggan():
    """the dual-goal GAN on the coefficient task (coefficient/dggan.py:13-64), discriminator scaled
    so that the gradient penalty (on the per-example fake scores) is active."""
    from coefficient.dggan import CoefficientDgganExperiment
    _coefficient(CoefficientDgganExperiment, 'g4b_coefficient_dggan_gp_active', steps=2, batch_size=64, seed_offset=1)


# ----------------------------------------------------------------------------------------------- image models
class _ImageExperiment(ref_srgan.Experiment):
    """The reference Experiment with only the three abstract hooks filled in."""
    builders = None

    def dataset_setup(self):
        pass

    def model_setup(self):
        self.G, self.D, self.DNN = self.builders()

    def validation_summaries(self, step):
        pass


def _image_experiment(builders, batch_size, multipliers=None, crowd=False):
    settings = Settings()
    settings.batch_size = batch_size
    for key, value in (multipliers or {}).items():
        setattr(settings, key, value)
    experiment = _ImageExperiment(settings)
    experiment.builders = builders
    if crowd:
        from crowd.srgan import CrowdExperiment
        experiment.labeled_loss_function = CrowdExperiment.labeled_loss_function.__get__(experiment)
    ref_utility.seed_all(0)
    experiment.model_setup()
    experiment.prepare_optimizers()
    experiment.train_mode()
    attach_writers(experiment)
    return experiment


def _uniform_images(generator, batch, size, channels=3):
    return torch.rand(batch, channels, size, size, generator=generator) * 2 - 1


def g5_tiny_dcgan():
    """Tiny DCGAN (conv_dim 8, 32x32, B=4) full step with the gradient penalty ACTIVE, every tensor kept."""
    from age.models import Generator, Discriminator

    def builders():
        return (Generator(image_size=32, conv_dim=8), Discriminator(image_size=32, conv_dim=8),
                Discriminator(image_size=32, conv_dim=8))

    experiment = _image_experiment(builders, batch_size=4, multipliers={
        'matching_loss_multiplier': 1e2, 'contrasting_loss_multiplier': 1e1, 'gradient_penalty_multiplier': 1e2})
    with torch.no_grad():
        for p in experiment.D.parameters():
            p.mul_(3.0)
    out = {'batch_size': np.array(4), 'image_size': np.array(32), 'conv_dim': np.array(8), 'd_scale': np.array(3.0)}
    out.update(state_arrays('init/D', experiment.D))
    out.update(state_arrays('init/DNN', experiment.DNN))
    out.update(state_arrays('init/G', experiment.G))
    generator = torch.Generator().manual_seed(5)
    batches = []
    for step in range(2):
        x, u = _uniform_images(generator, 4, 32), _uniform_images(generator, 4, 32)
        y = torch.rand(4, generator=generator) * 85 + 10
        batches.append((x, y, u))
        out[f's{step}/x'], out[f's{step}/y'], out[f's{step}/u'] = np32(x), np32(y), np32(u)
    run_recorded_steps(experiment, batches, out)
    out.update(state_arrays('final/D', experiment.D))
    out.update(state_arrays('final/DNN', experiment.DNN))
    out.update(state_arrays('final/G', experiment.G))
    out.update(adam_arrays('final_adam/D', experiment.D, experiment.d_optimizer))
    out.update(adam_arrays('final_adam/G', experiment.G, experiment.g_optimizer))
    save('g5_tiny_dcgan', **out)


def _layer_kat(prefix, module, x, out, tuple_output=False):
    """Forward, first-order grads for a random cotangent, and a gradient-penalty style double backward."""
    generator = torch.Generator().manual_seed(99)
    module.train()
    module.apply(ref_srgan.disable_batch_norm_updates)
    x = x.clone().requires_grad_()
    y = module(x)
    ys = list(y) if tuple_output else [y]
    cotangents = [torch.randn(t.shape, generator=generator) for t in ys]
    out[f'{prefix}/x'] = np32(x)
    out.update(state_arrays(f'{prefix}/state', module))
    for i, (t, c) in enumerate(zip(ys, cotangents)):
        out[f'{prefix}/y{i}'] = np32(t)
        out[f'{prefix}/c{i}'] = np32(c)
    scalar = sum((t * c).sum() for t, c in zip(ys, cotangents))
    module.zero_grad()
    (gx,) = torch.autograd.grad(scalar, x, create_graph=True)
    out[f'{prefix}/gx'] = np32(gx)
    first = torch.autograd.grad(scalar, list(module.parameters()), retain_graph=True, allow_unused=True)
    for (name, _), g in zip(module.named_parameters(), first):
        if g is not None:
            out[f'{prefix}/gparam/{name}'] = np32(g)
    penalty = (gx.reshape(gx.shape[0], -1).norm(dim=1) ** 2).mean()
    out[f'{prefix}/penalty'] = np32(penalty)
    second = torch.autograd.grad(penalty, list(module.parameters()) + [x], allow_unused=True)
    names = [n for n, _ in module.named_parameters()] + ['__x__']
    for name, g in zip(names, second):
        if g is not None:
            out[f'{prefix}/ggparam/{name}'] = np32(g)


def g6_layers():
    """Per-layer KATs at reduced sizes: _DenseLayer, _DenseBlock, _Transition, MapModule, conv_layer1 stem."""
    from collections import OrderedDict
    from torch import nn
    from crowd.models import _DenseLayer, _DenseBlock, _Transition, MapModule
    out = {}
    generator = torch.Generator().manual_seed(6)

    def randomise_batch_norm(module):
        for m in module.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.data = torch.rand(m.weight.shape, generator=generator) + 0.5
                m.bias.data = torch.randn(m.bias.shape, generator=generator) * 0.1
                m.running_mean.data = torch.randn(m.running_mean.shape, generator=generator) * 0.2
                m.running_var.data = torch.rand(m.running_var.shape, generator=generator) + 0.5

    ref_utility.seed_all(0)
    layer = _DenseLayer(16, 8, 4, 0)
    randomise_batch_norm(layer)
    _layer_kat('dense_layer', layer, torch.randn(3, 16, 6, 6, generator=generator), out)
    block = _DenseBlock(num_layers=3, num_input_features=8, bn_size=2, growth_rate=4, drop_rate=0)
    randomise_batch_norm(block)
    _layer_kat('dense_block', block, torch.randn(2, 8, 5, 5, generator=generator), out)
    transition = _Transition(32, 16)
    randomise_batch_norm(transition)
    _layer_kat('transition', transition, torch.randn(2, 32, 8, 8, generator=generator), out)
    map_module = MapModule(in_features=16, input_size=4, label_size=32)
    _layer_kat('map_module', map_module, torch.randn(2, 16, 4, 4, generator=generator), out, tuple_output=True)
    stem = nn.Sequential(OrderedDict([
        ('conv0', nn.Conv2d(3, 8, kernel_size=7, stride=2, padding=3, bias=False)),
        ('norm0', nn.BatchNorm2d(8)), ('relu0', nn.ReLU(inplace=True)),
        ('pool0', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
    randomise_batch_norm(stem)
    _layer_kat('stem', stem, torch.randn(2, 3, 20, 20, generator=generator), out)
    save('g6_layers', **out)


def _crowd_builders(size, pretrained=False):
    from crowd.models import DCGenerator, KnnDenseNetCat, MapModule
    from torch.nn.functional import avg_pool2d

    if size == 224:
        def builders():
            return DCGenerator(), KnnDenseNetCat(pretrained=False), KnnDenseNetCat(pretrained=False)
        return builders

    class SizedKnnDenseNetCat(KnnDenseNetCat):
        """Size-generalised oracle composed from the reference's own layers.  The three
        MapModules are built once, at the generalised sizes, at the point the reference builds them (so the
        constructor consumes the same random stream as a native-size build would)."""

        def __init__(self):
            import crowd.models as cm
            original = cm.MapModule
            remap = {28: size // 8, 14: size // 16, 7: size // 32}
            cm.MapModule = lambda in_features, input_size, label_size: original(
                in_features=in_features, input_size=remap[input_size], label_size=label_size)
            try:
                super().__init__(pretrained=False, label_patch_size=size)
            finally:
                cm.MapModule = original

        def forward(self, x):
            import crowd.models as cm
            original = cm.avg_pool2d
            cm.avg_pool2d = lambda t, kernel_size, stride: avg_pool2d(t, kernel_size=size // 32, stride=stride)
            try:
                return super().forward(x)
            finally:
                cm.avg_pool2d = original

    def builders():
        return DCGenerator(image_size=size), SizedKnnDenseNetCat(), SizedKnnDenseNetCat()
    return builders


def _crowd_inputs(generator, batch, size):
    x, u = _uniform_images(generator, batch, size), _uniform_images(generator, batch, size)
    heads = (torch.rand(batch, size, size, generator=generator) < 0.002).float()
    knn_map = torch.rand(batch, size, size, generator=generator)
    return x, (heads, knn_map), u


CROWD_MULTIPLIERS = {'matching_loss_multiplier': 1e3, 'contrasting_loss_multiplier': 1e2,
                     'gradient_penalty_multiplier': 1e2, 'map_multiplier': 1e-3}


def _crowd(name, size, batch, steps, d_scale=None, keep_grad_checksums=True):
    experiment = _image_experiment(_crowd_builders(size), batch, CROWD_MULTIPLIERS, crowd=True)
    out = {'batch_size': np.array(batch), 'image_size': np.array(size), 'input_seed': np.array(70 + size),
           'd_scale': np.array(d_scale if d_scale else 1.0)}
    if d_scale:
        with torch.no_grad():
            for module_name, module in experiment.D.named_modules():
                if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    module.weight.mul_(d_scale)
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('init_ck/G', experiment.G))
    generator = torch.Generator().manual_seed(70 + size)
    batches = []
    for step in range(steps):
        x, y, u = _crowd_inputs(generator, batch, size)
        batches.append((x, y, u))
        out[f's{step}/x_ck'] = np.array([x.double().sum().item(), x.double().abs().sum().item()])
        out[f's{step}/heads_sum'] = np32(y[0].sum(dim=(1, 2)))
    # One plain forward first: count + map checksums + features of D(x) (density maps are part of the parity bar).
    experiment.D.apply(ref_srgan.disable_batch_norm_updates)
    with torch.no_grad():
        density, count, maps = experiment.D(batches[0][0])
    out['fwd/count'] = np32(count)
    out['fwd/features'] = np32(experiment.D.features)
    out['fwd/maps_ck'] = np.array([maps.double().sum().item(), maps.double().abs().sum().item()])
    out['fwd/maps_sample'] = np32(maps[:, :, ::max(size // 8, 1), ::max(size // 8, 1)])
    out['fwd/density_abs_sum'] = np.array(density.abs().sum().item())
    run_recorded_steps(experiment, batches, out, with_grads_on_step0=False)
    if keep_grad_checksums:
        out.update(checksum_arrays('final_ck/D', experiment.D))
        out.update(checksum_arrays('final_ck/DNN', experiment.DNN))
        out.update(checksum_arrays('final_ck/G', experiment.G))
        out.update(checksum_arrays('last_grad_ck/D', experiment.D, grads=True))
        out.update(checksum_arrays('last_grad_ck/G', experiment.G, grads=True))
    save(name, **out)


def g7_crowd():
    _crowd('g7_crowd224', 224, batch=2, steps=1)
    _crowd('g7b_crowd64', 64, batch=2, steps=2)


def g7c_crowd_gp_active():
    _crowd('g7c_crowd64_gp_active', 64, batch=2, steps=1, d_scale=GP_ACTIVE_CROWD_SCALE)


GP_ACTIVE_CROWD_SCALE = 1.27


def g8_age():
    """age/driving DCGAN at the native 128x128 (age/models.py:32-80), B=4, run.py multipliers."""
    from age.models import Generator, Discriminator

    def builders():
        return Generator(), Discriminator(), Discriminator()

    experiment = _image_experiment(builders, 4, {'matching_loss_multiplier': 1e2, 'contrasting_loss_multiplier': 1e1,
                                                 'gradient_penalty_multiplier': 1e2})
    out = {'batch_size': np.array(4), 'image_size': np.array(128), 'input_seed': np.array(8)}
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/G', experiment.G))
    generator = torch.Generator().manual_seed(8)
    batches = []
    for step in range(2):
        x, u = _uniform_images(generator, 4, 128), _uniform_images(generator, 4, 128)
        y = torch.rand(4, generator=generator) * 85 + 10
        batches.append((x, y, u))
    run_recorded_steps(experiment, batches, out, with_grads_on_step0=False, features=False)
    out['s1/labeled_features_ck'] = np.array([experiment.labeled_features.double().sum().item(),
                                              experiment.labeled_features.double().abs().sum().item()])
    out.update(checksum_arrays('final_ck/D', experiment.D))
    out.update(checksum_arrays('final_ck/G', experiment.G))
    save('g8_age_dcgan128', **out)


def g8b_vgg():
    """VGG16 D/DNN + DCGAN G at 224x224 (age/vgg.py:28-53,151-162; age/srgan.py:44-47 without the download)."""
    from age.models import Generator
    from age.vgg import vgg16

    def builders():
        return Generator(image_size=224), vgg16(num_classes=1), vgg16(num_classes=1)

    experiment = _image_experiment(builders, 2, {'matching_loss_multiplier': 1e2, 'contrasting_loss_multiplier': 1e1,
                                                 'gradient_penalty_multiplier': 1e2})
    out = {'batch_size': np.array(2), 'image_size': np.array(224), 'input_seed': np.array(9)}
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('init_ck/G', experiment.G))
    generator = torch.Generator().manual_seed(9)
    x, u = _uniform_images(generator, 2, 224), _uniform_images(generator, 2, 224)
    y = torch.rand(2, generator=generator) * 85 + 10
    run_recorded_steps(experiment, [(x, y, u)], out, with_grads_on_step0=False, features=True)
    out.update(checksum_arrays('final_ck/D', experiment.D))
    save('g8b_vgg224', **out)


def g9_crowd_sliding_window():
    """``CrowdExperiment.predict_full_example`` (crowd/srgan.py:332-395) -- sliding-window patches
    of a full image through the discriminator, per-pixel averaging of the overlapping predictions.  The reference
    resizes each predicted density patch with ``scipy.misc.imresize`` (removed from SciPy); here the prediction
    already has the patch size, where that call is the identity, so the stand-in below only accepts that case."""
    import scipy.misc
    from crowd.srgan import CrowdExperiment
    from crowd.data import CrowdExample

    def imresize_identity(array, size, mode=None):
        assert tuple(array.shape) == tuple(size) and mode == 'F'
        return np.asarray(array, dtype=np.float32)
    scipy.misc.imresize = imresize_identity
    size = 64
    experiment = _image_experiment(_crowd_builders(size), 4, CROWD_MULTIPLIERS, crowd=True)
    experiment.settings.image_patch_size = size
    experiment.settings.test_sliding_window_size = 24
    experiment.settings.number_of_data_workers = 0
    experiment.settings.pin_memory = False
    for module in (experiment.D, experiment.DNN, experiment.G):
        module.eval()
    generator = np.random.RandomState(5)
    out = {'image_size': np.array(size), 'window_step': np.array(24), 'batch_size': np.array(4)}
    out.update(checksum_arrays('init_ck/D', experiment.D))
    for index, shape in enumerate([(100, 150), (64, 64), (40, 90)]):     # larger, exact and smaller than a patch
        image = generator.randint(0, 256, size=shape + (3,)).astype(np.uint8)
        example = CrowdExample(image=image, label=np.zeros(shape, dtype=np.float32))
        with torch.no_grad():
            count, label = CrowdExperiment.predict_full_example(experiment, example, experiment.D)
        out[f'e{index}/image'] = image
        out[f'e{index}/count'] = np.array(count, dtype=np.float64)
        out[f'e{index}/label_abs_sum'] = np.array(np.abs(label).sum(), dtype=np.float64)
    save('g9_crowd_sliding_window', **out)


def g10_crowd_dggan():
    """the dual-goal GAN on the crowd task (crowd/dggan.py:9-49 on KnnDenseNetCatDggan,
    crowd/models.py:903-1046) at 64x64 -- the network is size-generalised exactly like the SRGAN crowd oracle above, the
    loss methods are the reference's own, bound onto the three-hook experiment.  Discriminator scaled so that the
    gradient penalty (on the per-example real/fake scores) is active."""
    import crowd.models as cm
    from crowd.dggan import CrowdDgganExperiment
    from crowd.srgan import CrowdExperiment
    from torch.nn.functional import avg_pool2d
    size, batch, steps, d_scale = 64, 2, 2, 1.26

    class SizedKnnDenseNetCatDggan(cm.KnnDenseNetCatDggan):
        def __init__(self):
            original = cm.MapModuleDggan
            remap = {28: size // 8, 14: size // 16, 7: size // 32}
            cm.MapModuleDggan = lambda in_features, input_size, label_size: original(
                in_features=in_features, input_size=remap[input_size], label_size=label_size)
            try:
                super().__init__(pretrained=False, label_patch_size=size)
            finally:
                cm.MapModuleDggan = original

        def forward(self, x):
            original = cm.avg_pool2d
            cm.avg_pool2d = lambda t, kernel_size, stride: avg_pool2d(t, kernel_size=size // 32, stride=stride)
            try:
                return super().forward(x)
            finally:
                cm.avg_pool2d = original

    def builders():
        return cm.DCGenerator(image_size=size), SizedKnnDenseNetCatDggan(), SizedKnnDenseNetCatDggan()

    settings = Settings()
    settings.batch_size = batch
    for key, value in CROWD_MULTIPLIERS.items():
        setattr(settings, key, value)
    experiment = _ImageExperiment(settings)
    experiment.builders = builders
    experiment.labeled_loss_function = CrowdExperiment.labeled_loss_function.__get__(experiment)
    for method in ('unlabeled_loss_calculation', 'fake_loss_calculation', 'interpolate_loss_calculation',
                   'generator_loss_calculation'):
        setattr(experiment, method, getattr(CrowdDgganExperiment, method).__get__(experiment))
    ref_utility.seed_all(0)
    experiment.model_setup()
    experiment.prepare_optimizers()
    experiment.train_mode()
    attach_writers(experiment)
    out = {'batch_size': np.array(batch), 'image_size': np.array(size), 'input_seed': np.array(170 + size),
           'd_scale': np.array(d_scale), 'dggan_loss_multiplier': np.array(settings.dggan_loss_multiplier)}
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                module.weight.mul_(d_scale)
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('init_ck/G', experiment.G))
    generator = torch.Generator().manual_seed(170 + size)
    batches = [_crowd_inputs(generator, batch, size) for _ in range(steps)]
    experiment.D.apply(ref_srgan.disable_batch_norm_updates)
    with torch.no_grad():
        _, count, maps = experiment.D(batches[0][0])
    out['fwd/count'] = np32(count)
    out['fwd/real_label'] = np32(experiment.D.real_label)
    out['fwd/maps_ck'] = np.array([maps.double().sum().item(), maps.double().abs().sum().item()])
    run_recorded_steps(experiment, batches, out, with_grads_on_step0=False, features=False)
    out.update(checksum_arrays('final_ck/D', experiment.D))
    out.update(checksum_arrays('final_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('final_ck/G', experiment.G))
    save('g10_crowd_dggan64_gp_active', **out)


def g11_crowd_evaluation():
    """the crowd evaluation summaries -- ``CrowdExperiment.evaluation_epoch`` over patch batches
    (crowd/srgan.py:149-191: count ME / MAE / MSE over all batches, the kNN-map errors over the FIRST batch only, as the
    reference's indentation has it) and ``test_summaries`` over full images through ``predict_full_example``
    (crowd/srgan.py:261-300), both run from the unmodified reference on a 64x64 discriminator pair."""
    import scipy.misc
    from torch.utils.data import TensorDataset
    from crowd.srgan import CrowdExperiment

    def imresize_identity(array, size, mode=None):
        assert tuple(array.shape) == tuple(size) and mode == 'F'
        return np.asarray(array, dtype=np.float32)
    scipy.misc.imresize = imresize_identity
    size, batch = 64, 4
    experiment = _image_experiment(_crowd_builders(size), batch, CROWD_MULTIPLIERS, crowd=True)
    experiment.settings.image_patch_size = size
    experiment.settings.test_sliding_window_size = 32
    experiment.settings.number_of_data_workers = 0
    experiment.settings.pin_memory = False
    experiment.settings.test_summary_size = None
    experiment.settings.map_directory_name = 'unused'
    for module in (experiment.D, experiment.DNN, experiment.G):
        module.eval()
    for method in ('evaluation_epoch', 'images_to_predicted_labels', 'test_summaries', 'predict_full_example'):
        setattr(experiment, method, getattr(CrowdExperiment, method).__get__(experiment))
    out = {'image_size': np.array(size), 'batch_size': np.array(batch), 'window_step': np.array(32)}
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/DNN', experiment.DNN))
    generator = torch.Generator().manual_seed(211)
    images = torch.rand(10, 3, size, size, generator=generator) * 2 - 1
    labels = (torch.rand(10, size, size, generator=generator) < 0.004).float()
    maps = torch.rand(10, size, size, generator=generator)
    out['patches/images'], out['patches/labels'], out['patches/maps'] = np32(images), np32(labels), np32(maps)
    dataset = TensorDataset(images, labels, maps)
    with torch.no_grad():
        dnn_mae = experiment.evaluation_epoch(experiment.settings, experiment.DNN, dataset, experiment.dnn_summary_writer,
                                              '1 Validation Error', shuffle=False)
        experiment.evaluation_epoch(experiment.settings, experiment.D, dataset, experiment.gan_summary_writer,
                                    '1 Validation Error', comparison_value=dnn_mae, shuffle=False)
    random_state = np.random.RandomState(7)
    scenes = []
    for index, shape in enumerate([(80, 120), (64, 64), (50, 100)]):
        image = random_state.randint(0, 256, size=shape + (3,)).astype(np.uint8)
        label = (random_state.rand(*shape) < 0.004).astype(np.float32)
        scenes.append((image, label, np.zeros(shape, dtype=np.float32)))
        out[f'scene{index}/image'], out[f'scene{index}/label'] = image, label

    class TestDataset:
        length = len(scenes)

        def __init__(self, dataset, map_directory_name):
            assert dataset == 'test'

        def __getitem__(self, index):
            return scenes[index]
    experiment.dataset_class = TestDataset
    with torch.no_grad():
        experiment.test_summaries()
    for prefix, writer in (('dnn', experiment.dnn_summary_writer), ('gan', experiment.gan_summary_writer)):
        for tag, value in last_scalars(writer).items():
            out[f'{prefix}/{tag}'] = np.array(value, dtype=np.float64)
    save('g11_crowd_evaluation', **out)


def g12_crowd_labels():
    """the offline labels of the reference's database preprocessor (crowd/database_preprocessor.py:
    253-290): the point map of the annotated heads and the ikNN maps 1 / (generate_knn_map(k) + 1) for k = 1..5, on two
    small scenes (one with fewer heads than neighbours)."""
    from crowd.database_preprocessor import generate_density_label, generate_knn_map, generate_point_density_map
    random_state = np.random.RandomState(12)
    out = {}
    for index, (shape, heads) in enumerate((((40, 56), 30), ((24, 33), 3), ((72, 90), 400))):
        positions = random_state.rand(heads, 2) * (np.array(shape) - 1)          # (y, x) order, as the preprocessors pass
        out[f'scene{index}/shape'] = np.array(shape)
        out[f'scene{index}/heads_yx'] = positions
        density, outside = generate_point_density_map(positions, shape)
        out[f'scene{index}/point_map'] = density
        assert outside == 0
        for k in (1, 2, 3, 4, 5):
            out[f'scene{index}/i{k}nn_map'] = 1 / (generate_knn_map(positions, shape, number_of_neighbors=k) + 1)
        out[f'scene{index}/i3nn_map_bounded'] = 1 / (generate_knn_map(positions, shape, number_of_neighbors=3,
                                                                          upper_bound=6.0) + 1)
    # The "density{beta}" labels (crowd/database_preprocessor.py:82-91).  Under NumPy 2 the reference's window clipping
    # at the TOP / LEFT border wraps around (uint32 head coordinate minus a Python int), so these scenes keep their
    # heads a window away from those two borders; clipping at the bottom / right border is exercised.
    for index, (shape, heads, margin) in enumerate((((90, 110), 40, 34), ((72, 90), 400, 10))):
        positions = margin + random_state.rand(heads, 2) * (np.array(shape) - 1 - margin)
        out[f'dscene{index}/shape'] = np.array(shape)
        out[f'dscene{index}/heads_yx'] = positions
        for beta in (0.1, 0.3, 0.5):
            out[f'dscene{index}/density_beta{beta}'] = generate_density_label(
                positions, shape, perspective_resizing=True, yx_order=True, neighbor_deviation_beta=beta)
    save('g12_crowd_labels', **out)


def g12b_crowd_label_variants():
    """The other branches of the reference's ``generate_density_label`` (crowd/database_preprocessor.py:113-223): a
    perspective map (head sigma = 0.2 m x perspective), ``include_body`` (second, anisotropic Gaussian below the head),
    ``ignore_tiny`` (heads with a perspective < 3.1 dropped and not counted), ``perspective_resizing=False`` (sigma 8),
    ``force_full_image_count_normalize=False`` and the (x, y) position order.  Heads keep a window's distance from the top /
    left border (under NumPy 2 the reference's clipping there wraps around, see g12); bottom / right clipping is exercised."""
    from crowd.database_preprocessor import generate_density_label
    random_state = np.random.RandomState(121)
    out = {}
    for index, (shape, heads) in enumerate((((96, 120), 60), ((80, 100), 300))):
        margin = 18
        positions = margin + random_state.rand(heads, 2) * (np.array(shape) - 1 - margin)          # (y, x)
        ys, xs = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing='ij')
        perspective = (1.2 + 7.5 * ys / shape[0] + 0.5 * np.sin(xs / 7.0)).astype(np.float32)
        out[f'scene{index}/shape'] = np.array(shape)
        out[f'scene{index}/heads_yx'] = positions
        out[f'scene{index}/perspective_map'] = perspective
        tiny = int(sum(perspective[int(np.rint(y)), int(np.rint(x))] < 3.1 for y, x in positions))
        assert 0 < tiny < heads
        out[f'scene{index}/tiny_heads'] = np.array(tiny)
        variants = {
            'perspective': dict(perspective=perspective),
            'perspective_body': dict(perspective=perspective, include_body=True),
            'perspective_tiny': dict(perspective=perspective, ignore_tiny=True),
            'perspective_body_tiny': dict(perspective=perspective, include_body=True, ignore_tiny=True),
            'perspective_body_unnormalized': dict(perspective=perspective, include_body=True,
                                                  force_full_image_count_normalize=False),
            'body_without_perspective': dict(include_body=True, neighbor_deviation_beta=0.3),
            'fixed_sigma': dict(perspective_resizing=False),
            'fixed_sigma_unnormalized': dict(perspective_resizing=False, force_full_image_count_normalize=False),
        }
        for name, arguments in variants.items():
            with contextlib.redirect_stdout(None):
                out[f'scene{index}/{name}'] = generate_density_label(positions, shape, yx_order=True, **arguments)
        with contextlib.redirect_stdout(None):
            out[f'scene{index}/perspective_body_xy'] = generate_density_label(positions[:, ::-1], shape, perspective=perspective,
                                                                                include_body=True, yx_order=False)
        assert np.array_equal(out[f'scene{index}/perspective_body_xy'], out[f'scene{index}/perspective_body'])
    save('g12b_crowd_label_variants', **out)


def g13_crowd_patches():
    """the training-batch assembly of the reference's crowd pipeline, by its own transforms
    (crowd/shanghai_tech_data.py:76-104 composes them): ``ExtractPatchForPosition(allow_padded=True)`` around a centre
    (crowd/data.py:370-453, zero padding where the patch leaves the scene), ``RandomHorizontalFlip`` (crowd/data.py:
    92-112; driven by ``random.choice`` -- the fixture seeds ``random`` per example and records what was chosen),
    ``NegativeOneToOneNormalizeImage`` and ``NumpyArraysToTorchTensors`` (crowd/data.py:41-62,115-128).  Scenes larger
    than, equal to and smaller than a patch; centres inside, on the border and in every corner."""
    import random
    from crowd.data import (CrowdExample, ExtractPatchForPosition, RandomHorizontalFlip, NegativeOneToOneNormalizeImage,
                            NumpyArraysToTorchTensors)
    generator = np.random.RandomState(13)
    size = 32
    out = {'patch_size': np.array(size)}
    scenes = []
    for index, shape in enumerate([(50, 70), (32, 32), (20, 45), (64, 33)]):
        image = generator.randint(0, 256, size=shape + (3,)).astype(np.uint8)
        label = (generator.rand(*shape) < 0.02).astype(np.float32) * generator.rand(*shape).astype(np.float32)
        map_ = generator.rand(*shape).astype(np.float32)
        scenes.append((image, label, map_))
        out[f'scene{index}/image'], out[f'scene{index}/label'], out[f'scene{index}/map'] = image, label, map_
    draws = [(0, 16, 16), (0, 25, 35), (0, 34, 54), (0, 0, 0), (0, 49, 69), (0, 3, 60), (0, 45, 8), (1, 16, 16),
             (1, 0, 31), (2, 10, 22), (2, 19, 0), (2, 0, 44), (3, 32, 16), (3, 63, 32), (3, 5, 17), (0, 16, 53)]
    extract = ExtractPatchForPosition(size, size, allow_padded=True)
    records = []
    for number, (scene, y, x) in enumerate(draws):
        image, label, map_ = scenes[scene]
        example = extract(CrowdExample(image=image, label=label, map_=map_), y, x)
        random.seed(1000 + number)
        flipped = random.choice([True, False])
        random.seed(1000 + number)
        example = RandomHorizontalFlip()(example)
        example = NumpyArraysToTorchTensors()(NegativeOneToOneNormalizeImage()(example))
        records.append((scene, y, x, int(flipped)))
        out[f'patch{number}/image'] = example.image.numpy()
        out[f'patch{number}/label'] = example.label.numpy()
        out[f'patch{number}/map'] = example.map.numpy()
        assert example.image.shape == (3, size, size) and example.label.shape == (size, size)
    out['draws'] = np.array(records, dtype=np.int32)
    assert 0 < out['draws'][:, 3].sum() < len(draws)          # both flip states occur
    save('g13_crowd_patches', **out)


def g14_crowd_sgan(size=64, batch=4, steps=2, d_scale=3.0):
    """the crowd SGAN (reference crowd/sgan.py:10-87 on ``JointDCDiscriminator``,
    crowd/models.py:150-178, dispatched at run.py:55).  Upstream the experiment is stale -- its ``model_setup`` pairs a
    224-pixel generator with a 128-pixel discriminator and ``CrowdExperiment`` no longer produces quarter-resolution
    density labels -- so, as for g10, the fixture binds the reference's own loss methods onto an experiment whose three
    networks are the reference classes at ONE image size and feeds it density-label batches (B, S/4, S/4): two whole
    training iterations through the reference's dnn_training_step / gan_training_step.  D's convolutions are scaled so
    that the gradient penalty (on the batch-mean binary cross-entropy, multiplier applied twice: Appendix A.7) is active."""
    import crowd.models as cm
    from crowd.sgan import CrowdSganExperiment
    bins_count = 10

    def builders():
        return (cm.DCGenerator(image_size=size), cm.JointDCDiscriminator(image_size=size, number_of_outputs=bins_count),
                cm.JointDCDiscriminator(image_size=size, number_of_outputs=bins_count))
    settings = Settings()
    settings.batch_size, settings.number_of_bins = batch, bins_count
    settings.matching_loss_multiplier, settings.gradient_penalty_multiplier = 1e1, 1e1
    experiment = _ImageExperiment(settings)
    experiment.builders = builders
    experiment.labeled_criterion, experiment.gan_criterion = torch.nn.CrossEntropyLoss(), torch.nn.BCEWithLogitsLoss()
    experiment.bins = torch.linspace(0, 300, bins_count)
    for method in ('dnn_loss_calculation', 'labeled_loss_calculation', 'unlabeled_loss_calculation', 'fake_loss_calculation',
                   'interpolate_loss_calculation', 'generator_loss_calculation', 'images_to_predicted_labels'):
        setattr(experiment, method, getattr(CrowdSganExperiment, method).__get__(experiment))
    ref_utility.seed_all(0)
    experiment.model_setup()
    experiment.prepare_optimizers()
    experiment.train_mode()
    attach_writers(experiment)
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                module.weight.mul_(d_scale)
    out = {'batch_size': np.array(batch), 'image_size': np.array(size), 'input_seed': np.array(140 + size),
           'd_scale': np.array(d_scale), 'number_of_bins': np.array(bins_count),
           'matching_loss_multiplier': np.array(1e1), 'gradient_penalty_multiplier': np.array(1e1)}
    out.update(checksum_arrays('init_ck/D', experiment.D))
    out.update(checksum_arrays('init_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('init_ck/G', experiment.G))
    generator = torch.Generator().manual_seed(140 + size)
    batches = []
    for _ in range(steps):
        x, u = _uniform_images(generator, batch, size), _uniform_images(generator, batch, size)
        # quarter-resolution density labels: sparse heads, counts spread over several of the bins of [0, 300]
        density = (torch.rand(batch, size // 4, size // 4, generator=generator) < 0.1).float() * \
            torch.rand(batch, 1, 1, generator=generator) * 8
        batches.append((x, density, u))
    experiment.D.apply(ref_srgan.disable_batch_norm_updates)
    with torch.no_grad():
        density, logits = experiment.D(batches[0][0])
        _, counts = experiment.images_to_predicted_labels(experiment.D, batches[0][0])
    out['fwd/density'] = np32(density)
    out['fwd/count_logits'] = np32(logits)
    out['fwd/predicted_counts'] = np32(counts)
    out['fwd/label_counts'] = np32(batches[0][1].sum(1).sum(1))
    run_recorded_steps(experiment, batches, out, with_grads_on_step0=False, features=False)
    assert float(out['s0/gradient_penalty']) > 0 and float(out[f's{steps - 1}/gradient_penalty']) > 0, 'penalty inactive'
    print({k: float(v) for k, v in out.items() if k.startswith('s') and np.ndim(v) == 0})
    out.update(checksum_arrays('final_ck/D', experiment.D))
    out.update(checksum_arrays('final_ck/DNN', experiment.DNN))
    out.update(checksum_arrays('final_ck/G', experiment.G))
    save('g14_crowd_sgan64_gp_active', **out)


ALL = {'g0': g0_toydata, 'g1': g1_distance, 'g2': g2_sgan_math, 'g3': g3_coefficient_srgan,
       'g4': g4_coefficient_sgan, 'g4b': g4b_coefficient_dggan, 'g5': g5_tiny_dcgan, 'g6': g6_layers, 'g7': g7_crowd, 'g7c': g7c_crowd_gp_active, 'g8': g8_age,
       'g8b': g8b_vgg, 'g9': g9_crowd_sliding_window, 'g10': g10_crowd_dggan, 'g11': g11_crowd_evaluation, 'g12': g12_crowd_labels, 'g12b': g12b_crowd_label_variants,
       'g13': g13_crowd_patches, 'g14': g14_crowd_sgan}

if __name__ == '__main__':
    wanted = sys.argv[1:] or ['all']
    for key, function in ALL.items():
        if 'all' in wanted or key in wanted:
            print(f'== {key}')
            function()
