"""CPU: the product's network definitions reproduce the reference's parameter initialisation (same construction order,
same random stream) -- checked against the checksums stored in the reference-generated fixtures."""
import torch

from helpers import load_golden, checksum, assert_close


def test_crowd_dggan_networks_initialise_like_the_reference():
    """KnnDenseNetCatDggan / DCGenerator at 64x64 (reference crowd/models.py:903-1046,127-147):
    seed_all(0), then G, D, DNN in the order of the reference's model_setup."""
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCatDggan
    from srgan_amd.utility import seed_all
    g = load_golden('g10_crowd_dggan64_gp_active')
    size = int(g['image_size'])
    seed_all(0)
    generator = DCGenerator(image_size=size)
    discriminator, dnn = KnnDenseNetCatDggan(image_size=size), KnnDenseNetCatDggan(image_size=size)
    with torch.no_grad():
        for m in discriminator.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                m.weight.mul_(float(g['d_scale']))
    for module, prefix in ((discriminator, 'init_ck/D'), (dnn, 'init_ck/DNN'), (generator, 'init_ck/G')):
        names = [name for name, _ in module.named_parameters()]
        assert sorted(names) == sorted(k[len(prefix) + 1:] for k in g.files if k.startswith(prefix + '/'))
        for name, parameter in module.named_parameters():
            assert_close(checksum(parameter), g[f'{prefix}/{name}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {name}')
    assert discriminator.count_layer.weight.shape[0] == 2 and discriminator.map_module1.count_layer.weight.shape[0] == 2


def test_product_toy_dataset_reproduces_the_reference_draws():
    """coefficient/data.py: the same seed gives the reference's dataset bit for bit (golden g0); small datasets are tiled
    up to the batch size."""
    from types import SimpleNamespace
    import numpy as np
    import srgan_amd  # noqa: F401
    from srgan_amd.coefficient.data import ToyDataset
    g = load_golden('g0_toydata')
    dataset = ToyDataset(int(g['size']), 10, SimpleNamespace(batch_size=1), seed=int(g['seed']))
    np.testing.assert_array_equal(dataset.examples, g['examples'])
    np.testing.assert_array_equal(dataset.labels, g['labels'])
    example, label = dataset[3]
    assert example.shape == (50,) and example.dtype == np.float32 and len(dataset) == int(g['size'])
    assert len(ToyDataset(3, 10, SimpleNamespace(batch_size=7), seed=1)) == 6


def test_crowd_sgan_networks_initialise_like_the_reference():
    """JointDCDiscriminator (reference crowd/models.py:150-178) and DCGenerator at 64 x 64 with ten class outputs: same
    parameter names and initial values as the reference classes (golden g14), and the run.py dispatch of the method."""
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.models import DCGenerator, JointDCDiscriminator
    from srgan_amd.crowd.sgan import CrowdSganExperiment
    from srgan_amd.run import EXPERIMENTS
    from srgan_amd.settings import ApplicationName, MethodName
    from srgan_amd.utility import seed_all
    g = load_golden('g14_crowd_sgan64_gp_active')
    size, bins = int(g['image_size']), int(g['number_of_bins'])
    seed_all(0)
    generator = DCGenerator(image_size=size)
    discriminator, dnn = (JointDCDiscriminator(image_size=size, number_of_outputs=bins) for _ in range(2))
    with torch.no_grad():
        for m in discriminator.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                m.weight.mul_(float(g['d_scale']))
    for module, prefix in ((discriminator, 'init_ck/D'), (dnn, 'init_ck/DNN'), (generator, 'init_ck/G')):
        names = [name for name, _ in module.named_parameters()]
        assert sorted(names) == sorted(k[len(prefix) + 1:] for k in g.files if k.startswith(prefix + '/'))
        for name, parameter in module.named_parameters():
            assert_close(checksum(parameter), g[f'{prefix}/{name}'], rtol=1e-9, atol=1e-12, what=f'{prefix} {name}')
    assert tuple(discriminator.density_layer5[0].weight.shape) == ((size // 4) ** 2, 512, size // 16, size // 16)
    assert EXPERIMENTS[ApplicationName.crowd][MethodName.sgan] is CrowdSganExperiment
