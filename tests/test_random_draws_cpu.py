"""H12: the PRODUCT's own random-draw path against fixtures written by the unmodified reference.

The step tests inject z_D / alpha / z_G from the goldens; here nothing is injected: the product's `seed_all`,
`dataset_setup`, `model_setup`, loaders and `Experiment.draw_*` must walk the NumPy / torch host streams exactly as
the reference does (utility.py:102-116, srgan.py:286-289,301,364, coefficient/srgan.py:17-38).  CPU only: the draws are
host tensors on both sides (the reference's device draw of alpha came from the CPU generator when the goldens were made).
The same sequence through an un-injected training step on the device is tests/test_reference_runs_gpu.py."""
import numpy as np
import torch

from helpers import load_golden, golden_state


def coefficient_experiment(batch_size):
    """The product's coefficient experiment configured as tests/golden/make_goldens.py:_coefficient configures the
    reference's, up to (not including) the first training step."""
    import srgan_amd  # noqa: F401
    from srgan_amd.settings import Settings
    from srgan_amd.coefficient.srgan import CoefficientExperiment
    from srgan_amd.utility import seed_all
    settings = Settings()
    settings.batch_size = batch_size
    settings.labeled_dataset_size = 2 * batch_size
    settings.unlabeled_dataset_size = 4 * batch_size
    settings.validation_dataset_size = batch_size
    settings.pin_memory = False
    settings.gradient_penalty_multiplier = 1e1
    settings.number_of_data_workers = 0
    experiment = CoefficientExperiment(settings)
    seed_all(0)
    experiment.dataset_setup()
    experiment.model_setup()
    return experiment


def fetch_batches(experiment, steps):
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
    batches = []
    for _ in range(steps):
        x, y = next(labeled)
        batches.append((x, y, next(unlabeled)[0]))
    return batches


def test_mixture_model_draw_is_the_references():
    """`MixtureModel.rvs` (reference utility.py:102-107) from NumPy's global stream: golden g0 `mixture_rvs`."""
    from scipy.stats import norm
    from srgan_amd.utility import MixtureModel, seed_all
    g = load_golden('g0_toydata')
    seed_all(int(g['mixture_seed']))
    offset = float(g['mixture_offset'])
    np.testing.assert_array_equal(MixtureModel([norm(-offset, 1), norm(offset, 1)]).rvs(size=[6, 5]), g['mixture_rvs'])


def test_experiment_draw_discriminator_noise_is_the_references_mixture():
    """`Experiment.draw_discriminator_noise` = the float64 mixture cast to float32 (reference srgan.py:286-289), with the
    settings' `mean_offset`: seeded like g0, it returns g0's values."""
    from srgan_amd.utility import seed_all
    g = load_golden('g0_toydata')
    experiment = coefficient_experiment(64)
    experiment.settings.mean_offset = float(g['mixture_offset'])
    experiment.G.input_size = 5
    seed_all(int(g['mixture_seed']))
    drawn = experiment.draw_discriminator_noise(6)
    assert drawn.dtype == torch.float32 and tuple(drawn.shape) == (6, 5)
    np.testing.assert_array_equal(drawn.numpy(), g['mixture_rvs'].astype(np.float32))


def test_uninjected_draws_follow_the_reference_streams():
    """From `seed_all(0)` through set-up and three fetched batches, the product's initial weights, batches and the
    (z_D, alpha, z_G) draws of three consecutive iterations are the reference's, bit for bit (golden g3)."""
    g = load_golden('g3_coefficient_srgan')
    batch = int(g['batch_size'])
    experiment = coefficient_experiment(batch)
    for module, prefix in ((experiment.D, 'init/D'), (experiment.DNN, 'init/DNN'), (experiment.G, 'init/G')):
        state = module.state_dict()
        for key, value in golden_state(g, prefix).items():
            np.testing.assert_array_equal(state[key].numpy(), value.numpy(), err_msg=f'{prefix}/{key}')
    steps = int(g['steps'])
    for step, (x, y, u) in enumerate(fetch_batches(experiment, steps)):
        np.testing.assert_array_equal(x.numpy(), g[f's{step}/x'])
        np.testing.assert_array_equal(y.numpy(), g[f's{step}/y'])
        np.testing.assert_array_equal(u.numpy(), g[f's{step}/u'])
    for step in range(steps):
        # the order of one gan_training_step: z_D (NumPy global stream), alpha, then z_G (both torch's CPU stream)
        z_d = experiment.draw_discriminator_noise(batch)
        alpha = experiment.draw_interpolation_alpha(batch)
        z_g = experiment.draw_generator_noise(batch)
        np.testing.assert_array_equal(z_d.numpy(), g[f's{step}/z_d'], err_msg=f'z_D of step {step}')
        np.testing.assert_array_equal(alpha.numpy(), g[f's{step}/alpha'].reshape(-1), err_msg=f'alpha of step {step}')
        np.testing.assert_array_equal(z_g.numpy(), g[f's{step}/z_g'], err_msg=f'z_G of step {step}')
