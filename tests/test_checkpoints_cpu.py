"""SURVEY.md 8(f) N1, host-side only (no kernels): loading torchvision densenet201 weights into the crowd trunk with
the reference's key renaming (crowd/models.py:1103-1129)."""
from collections import OrderedDict

import pytest
import torch


def _torchvision_style(trunk_state, legacy):
    """Spell a trunk state dict the way torchvision's densenet201 checkpoint does (optionally the pre-0.4 'norm.1'
    member names the reference still handles), plus the ImageNet classifier the loader must drop."""
    out = OrderedDict()
    for key, value in trunk_state.items():
        if legacy and key.endswith('num_batches_tracked'):     # checkpoints of that era predate the counter
            continue
        if key.startswith('dense_blocks.'):
            key = 'features.' + key[len('dense_blocks.'):]
        elif key.startswith('transition_layers.'):
            key = 'features.' + key[len('transition_layers.'):]
        elif key.startswith('conv_layer1.'):
            key = 'features.' + key[len('conv_layer1.'):]
        elif key.startswith('norm5.'):
            key = 'features.' + key
        if legacy and 'denselayer' in key:
            for member in ('norm', 'conv'):
                for index in ('1', '2'):
                    key = key.replace(f'.{member}{index}.', f'.{member}.{index}.')
        out[key] = value.clone()
    out['classifier.weight'] = torch.zeros(1000, 8)
    out['classifier.bias'] = torch.zeros(1000)
    return out


@pytest.mark.parametrize('legacy', [False, True])
def test_torchvision_densenet_keys_load_into_the_trunk(tmp_path, monkeypatch, legacy):
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.models import KnnDenseNetCat
    small = dict(growth_rate=4, block_config=(2, 2, 2, 2), num_init_features=8, bn_size=2, image_size=64)
    donor = KnnDenseNetCat(**small)
    trunk = OrderedDict((k, v) for k, v in donor.state_dict().items()
                        if k.split('.')[0] in ('conv_layer1', 'dense_blocks', 'transition_layers', 'norm5'))
    generator = torch.Generator().manual_seed(0)
    for value in trunk.values():
        if value.dtype.is_floating_point:
            value.copy_(torch.randn(value.shape, generator=generator))
    checkpoint = _torchvision_style(trunk, legacy)
    assert any(k.startswith('features.denseblock1.denselayer1.') for k in checkpoint)
    if legacy:
        assert any('.norm.1.' in k for k in checkpoint)
    path = tmp_path / 'densenet.pth'
    torch.save(checkpoint, path)
    for source in (checkpoint, str(path)):
        model = KnnDenseNetCat(pretrained=source, **small)
        for key, value in trunk.items():
            if not (legacy and key.endswith('num_batches_tracked')):
                assert torch.equal(model.state_dict()[key], value), key
    monkeypatch.setenv(KnnDenseNetCat.TORCHVISION_WEIGHTS_ENV, str(path))
    model = KnnDenseNetCat(pretrained=True, **small)
    assert torch.equal(model.state_dict()['norm5.weight'], trunk['norm5.weight'])
    monkeypatch.delenv(KnnDenseNetCat.TORCHVISION_WEIGHTS_ENV)
    with pytest.raises(RuntimeError):
        KnnDenseNetCat(pretrained=True, **small)
    broken = OrderedDict(checkpoint)
    del broken['features.norm5.weight']
    with pytest.raises(RuntimeError):
        KnnDenseNetCat(pretrained=broken, **small)


def test_vgg16_pretrained_follows_the_reference_strict_false_semantics():
    """age/vgg.py:157-162: torchvision keys are applied with strict=False to differently named modules, so only
    classifier.0 lands; everything else keeps its value (init_weights is switched off by pretrained)."""
    import srgan_amd  # noqa: F401
    from srgan_amd.age import vgg
    generator = torch.Generator().manual_seed(1)
    checkpoint = OrderedDict([('features.0.weight', torch.randn(64, 3, 3, 3, generator=generator)),
                              ('features.0.bias', torch.randn(64, generator=generator)),
                              ('classifier.0.weight', torch.randn(4096, 512, generator=generator)),
                              ('classifier.0.bias', torch.randn(4096, generator=generator)),
                              ('classifier.3.weight', torch.randn(4096, 4096, generator=generator)),
                              ('classifier.6.bias', torch.randn(1000, generator=generator))])
    model = vgg.vgg16(pretrained=checkpoint, num_classes=1, image_size=32)       # 512 * 1 * 1 classifier inputs
    state = model.state_dict()
    assert torch.equal(state['classifier.0.weight'], checkpoint['classifier.0.weight'])
    assert torch.equal(state['classifier.0.bias'], checkpoint['classifier.0.bias'])
    assert not torch.equal(state['feature_layers.0.weight'], checkpoint['features.0.weight'])
    with pytest.raises(RuntimeError):
        vgg.vgg16(pretrained=True)
