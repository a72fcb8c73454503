"""Crowd-counting networks on HIP kernels: the DCGAN generator and the DenseNet-201 based
``KnnDenseNetCat`` discriminator (surface of reference crowd/models.py:127-147,335-371,763-786,1049-1166).

Sizes the reference hard-codes (map-head inputs 28/14/7, the 7x7 final pool) are derived from
``image_size`` as S/8, S/16, S/32; at 224 it is the reference graph.
The unused experimental networks of the reference file are out of scope."""
from collections import OrderedDict

from torch import nn as torch_nn

from .. import functional as F
from .. import nn
from .. import fused
from ..age.models import Generator as _DCGANGenerator, convolution, LEAK
from ..utility import seed_all


FUSED_STEM_POOL = True     # norm0 -> relu0 -> pool0 of the DenseNet stem as one pass each way (functional.bn_relu_max_pool2d)


class DCGenerator(_DCGANGenerator):
    """reference crowd/models.py:127-147 (defaults to 224x224)."""

    def __init__(self, z_dim=256, image_size=224, conv_dim=64):
        super().__init__(z_dim=z_dim, image_size=image_size, conv_dim=conv_dim)


class JointDCDiscriminator(nn.Module):
    """DCGAN-like discriminator with two heads on the 512 x S/16 x S/16 trunk: ``count_layer5`` -> count (B) or class
    logits (B, n), ``density_layer5`` -> a quarter-resolution density map (B, S/4, S/4); ``features`` = the flattened
    trunk (reference crowd/models.py:150-178; no batch-norm: its module switch is off, crowd/models.py:108)."""

    def __init__(self, image_size=128, conv_dim=64, number_of_outputs=1):
        seed_all(0)
        super().__init__()
        self.number_of_outputs = number_of_outputs
        self.layer1 = convolution(3, conv_dim, 4, bn=False)
        self.layer2 = convolution(conv_dim, conv_dim * 2, 4)
        self.layer3 = convolution(conv_dim * 2, conv_dim * 4, 4)
        self.layer4 = convolution(conv_dim * 4, conv_dim * 8, 4)
        self.count_layer5 = convolution(conv_dim * 8, self.number_of_outputs, int(image_size / 16), 1, 0, False)
        self.density_layer5 = convolution(conv_dim * 8, int(image_size / 4) ** 2, int(image_size / 16), 1, 0, False)
        self.features = None

    def forward(self, x):
        out = x
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            out = F.leaky_relu(stage(out), LEAK)
        self.features = F.flatten2d(out)
        count = self.count_layer5(out)
        count = F.view(count, (-1,) if self.number_of_outputs == 1 else (-1, self.number_of_outputs))
        side = int(x.shape[2] / 4)
        density = F.view(self.density_layer5(out), (-1, side, side))
        return density, count


class _DenseLayer(nn.Sequential):
    """BN-ReLU-conv1x1(bn_size*k)-BN-ReLU-conv3x3(k); output is cat([x, new]) (crowd/models.py:335-353)."""

    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate=0):
        super().__init__()
        self.add_module('norm1', nn.BatchNorm2d(num_input_features))
        self.add_module('relu1', nn.ReLU(inplace=True))
        self.add_module('conv1', nn.Conv2d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1,
                                           bias=False))
        self.add_module('norm2', nn.BatchNorm2d(bn_size * growth_rate))
        self.add_module('relu2', nn.ReLU(inplace=True))
        self.add_module('conv2', nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1,
                                           bias=False))
        if drop_rate:
            raise NotImplementedError('drop_rate > 0 is never used on the hot path (crowd/models.py:1063)')

    def forward(self, x):
        # BN + ReLU are one fused kernel each (same arithmetic as running norm1, relu1, ... in sequence)
        bottleneck = self.conv1(self.norm1(x, relu=True))
        return F.cat_channels([x, self.conv2(self.norm2(bottleneck, relu=True))])


class _DenseBlock(nn.Sequential):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate=0):
        super().__init__()
        for i in range(num_layers):
            self.add_module('denselayer%d' % (i + 1),
                            _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate))

    def forward(self, x):
        """The whole block is one concat-free node (fused.py), first and second order (its recorded backward is
        itself a node: the gradient penalty); ``fused.ENABLED = False`` selects the layer-by-layer primitive ops."""
        if not fused.ENABLED:
            return super().forward(x)
        return fused.dense_block(x, list(self.children()))


class _Transition(nn.Sequential):
    def __init__(self, num_input_features, num_output_features):
        super().__init__()
        self.add_module('norm', nn.BatchNorm2d(num_input_features))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1,
                                          bias=False))
        self.add_module('pool', nn.AvgPool2d(kernel_size=2, stride=2))

    def forward(self, x):
        if fused.POOL_FIRST:
            # avg_pool2d and a 1x1 convolution commute -- both are linear and the convolution acts per pixel -- so the
            # reference's conv -> pool (crowd/models.py:369-371) is evaluated as pool -> conv: the convolution, its data
            # and weight gradients and their double-backward forms run on a QUARTER of the pixels (the transitions are
            # 12 % of the network's convolution FLOPs).  Same function, fp32 summation order aside (~1e-7 relative).
            pooled = None
            if fused.ENABLED and self.pool.kernel_size == 2 and self.pool.stride == 2:
                inv_std, mean = self.norm._inverse_std()
                pooled = F.bn_relu_avg_pool2d(x, mean, inv_std, nn.P(self.norm.weight), nn.P(self.norm.bias))
            if pooled is None:
                pooled = self.pool(self.norm(x, relu=True))
            return self.conv(pooled)
        y = fused.bn_relu_conv(x, self.norm, self.conv) if fused.ENABLED else None
        if y is None:
            y = self.conv(self.norm(x, relu=True))
        return self.pool(y)


class MapModule(nn.Module):
    """Transposed conv to the label size, then conv k2 s2 x3, a whole-map "linear" conv and a count layer,
    leaky_relu 0.01 throughout (reference crowd/models.py:763-786)."""

    def __init__(self, in_features, input_size, label_size, count_outputs=1):
        super().__init__()
        kernel_size = label_size // input_size
        self.map_transposed_conv_layer = nn.ConvTranspose2d(in_channels=in_features, out_channels=1,
                                                            kernel_size=kernel_size, stride=kernel_size)
        self.conv1 = nn.Conv2d(in_channels=1, out_channels=8, kernel_size=2, stride=2)
        self.conv2 = nn.Conv2d(in_channels=8, out_channels=16, kernel_size=2, stride=2)
        self.conv3 = nn.Conv2d(in_channels=16, out_channels=32, kernel_size=2, stride=2)
        self.linear1 = nn.Conv2d(in_channels=32, out_channels=20, kernel_size=label_size // (2 ** 3))
        self.count_layer = nn.Conv2d(in_channels=20, out_channels=count_outputs, kernel_size=1)

    def forward(self, x):
        map_ = F.leaky_relu(self.map_transposed_conv_layer(x))
        out = F.leaky_relu(self.conv1(map_))
        out = F.leaky_relu(self.conv2(out))
        out = F.leaky_relu(self.conv3(out))
        out = F.leaky_relu(self.linear1(out))
        return map_, self.count_layer(out), out


class KnnDenseNetCat(nn.Module):
    """DenseNet-201 trunk, three map heads on the transition outputs and a count head; ``features`` is the
    concatenation of the four 20-d hidden vectors, shape (B, 80, 1, 1) (reference crowd/models.py:1049-1166).

    Returns ``(density, count, maps)`` like the reference; ``density`` is the reference's all-zero
    (B, S, S) placeholder, produced once per shape and reused (Appendix A.11)."""

    COUNT_OUTPUTS = 1      # 2 in the dual-goal variant: (count, real/fake score)

    def __init__(self, growth_rate=32, block_config=(6, 12, 48, 32), num_init_features=64, bn_size=4, drop_rate=0,
                 pretrained=False, label_patch_size=224, image_size=None):
        super().__init__()
        image_size = image_size or label_patch_size
        self.label_patch_size = image_size
        self.dense_blocks = nn.ModuleList()
        self.transition_layers = nn.ModuleList()
        self.conv_layer1 = nn.Sequential(OrderedDict([
            ('conv0', nn.Conv2d(3, num_init_features, kernel_size=7, stride=2, padding=3, bias=False)),
            ('norm0', nn.BatchNorm2d(num_init_features)),
            ('relu0', nn.ReLU(inplace=True)),
            ('pool0', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
        num_features = num_init_features
        widths = []
        for i, num_layers in enumerate(block_config):
            block = _DenseBlock(num_layers=num_layers, num_input_features=num_features, bn_size=bn_size,
                                growth_rate=growth_rate, drop_rate=drop_rate)
            self.dense_blocks.add_module('denseblock%d' % (i + 1), block)
            num_features = num_features + num_layers * growth_rate
            if i != len(block_config) - 1:
                self.transition_layers.add_module('transition%d' % (i + 1),
                                                  _Transition(num_features, num_features // 2))
                num_features = num_features // 2
                widths.append(num_features)
        self.norm5 = nn.BatchNorm2d(num_features)
        for m in self.modules():      # "official init from torch repo" (crowd/models.py:1094-1101)
            if isinstance(m, torch_nn.Conv2d):
                torch_nn.init.kaiming_normal_(m.weight.data)
            elif isinstance(m, torch_nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, torch_nn.Linear):
                m.bias.data.zero_()
        if pretrained:      # the trunk only, before the heads exist -- as in the reference (crowd/models.py:1103-1129)
            self.load_torchvision_densenet201(pretrained)
        outputs = self.COUNT_OUTPUTS
        self.map_module1 = MapModule(widths[0], image_size // 8, image_size, count_outputs=outputs)
        self.map_module2 = MapModule(widths[1], image_size // 16, image_size, count_outputs=outputs)
        self.map_module3 = MapModule(widths[2], image_size // 32, image_size, count_outputs=outputs)
        self.final_count_feature_layer = nn.Conv2d(in_channels=num_features, out_channels=20, kernel_size=1)
        self.count_layer = nn.Conv2d(in_channels=20, out_channels=outputs, kernel_size=1)
        self.final_pool_size = image_size // 32
        self.features = None
        self._density_cache = None

    TORCHVISION_WEIGHTS_ENV = 'SRGAN_DENSENET201_WEIGHTS'

    @staticmethod
    def rename_torchvision_keys(state_dict):
        """torchvision ``densenet201`` state-dict keys -> this trunk's keys (reference crowd/models.py:1103-1127):
        old checkpoints spell dense-layer members ``norm.1`` / ``conv.2`` (dots are no longer legal in module names)
        -> ``norm1`` / ``conv2``; ``features.denseblockN`` -> ``dense_blocks.denseblockN``; ``features.transitionN``
        -> ``transition_layers.transitionN``; ``features.norm5`` -> ``norm5``; the other ``features.*`` (stem) ->
        ``conv_layer1.*``; the ImageNet classifier is dropped."""
        import re
        legacy = re.compile(r'^(.*denselayer\d+\.(?:norm|relu|conv))\.((?:[12])\.(?:weight|bias|running_mean|running_var))$')
        renamed = OrderedDict()
        for key, value in state_dict.items():
            if key.startswith('classifier.'):
                continue
            match = legacy.match(key)
            if match:
                key = match.group(1) + match.group(2)
            key = key.replace('features.denseblock', 'dense_blocks.denseblock')
            key = key.replace('features.transition', 'transition_layers.transition')
            key = key.replace('features.', '') if 'norm5' in key else key.replace('features.', 'conv_layer1.')
            renamed[key] = value
        return renamed

    def load_torchvision_densenet201(self, source=True):
        """Loads ImageNet densenet201 weights into the trunk (strictly: every trunk tensor must be present).
        ``source``: a state dict, a path to one saved with ``torch.save``, or True = the path in the environment
        variable SRGAN_DENSENET201_WEIGHTS (this build never downloads: there is no network on the training boxes)."""
        import os
        if source is True:
            source = os.environ.get(self.TORCHVISION_WEIGHTS_ENV)
            if not source:
                raise RuntimeError('pretrained=True needs the torchvision densenet201 state dict on disk: set '
                                   f'{self.TORCHVISION_WEIGHTS_ENV} to its path (the reference downloads it, '
                                   'crowd/models.py:1110; this build does not touch the network)')
        if isinstance(source, (str, os.PathLike)):
            import torch
            source = torch.load(source, map_location="cpu")
        self.load_state_dict(self.rename_torchvision_keys(source), strict=True)

    def _density(self, batch_size, like):
        cache = self._density_cache
        if cache is None or cache.shape[0] != batch_size or cache.data.device != like.data.device:
            cache = F.full((batch_size, self.label_patch_size, self.label_patch_size), 0.0, like)
            self._density_cache = cache
        return cache

    def forward(self, x):
        count, map_, hidden = self._heads(x)
        batch_size = x.shape[0]
        self.features = F.cat_channels([F.view(t, (batch_size, -1, 1, 1)) for t in hidden])
        return self._density(batch_size, x), F.view(count, (batch_size,)), map_

    def _heads(self, x):
        """Summed count-layer outputs (B, COUNT_OUTPUTS, 1, 1), the three maps (B, 3, S, S) and the four 20-d hidden
        vectors."""
        batch_size = x.shape[0]
        stem = self.conv_layer1
        out, first = None, stem.conv0(x)
        if FUSED_STEM_POOL and isinstance(stem.pool0.kernel_size, int):
            inv_std, mean = stem.norm0._inverse_std()
            out = F.bn_relu_max_pool2d(first, mean, inv_std, nn.parameter_var(stem.norm0.weight),
                                       nn.parameter_var(stem.norm0.bias), stem.pool0.kernel_size, stem.pool0.stride,
                                       stem.pool0.padding)
        if out is None:
            out = stem.pool0(stem.norm0(first, relu=True))
        t1_out = self.transition_layers.transition1(self.dense_blocks.denseblock1(out))
        t2_out = self.transition_layers.transition2(self.dense_blocks.denseblock2(t1_out))
        t3_out = self.transition_layers.transition3(self.dense_blocks.denseblock3(t2_out))
        db4_out = self.dense_blocks.denseblock4(t3_out)
        n5_relu_out = self.norm5(db4_out, relu=True)
        final_pool = F.avg_pool2d(n5_relu_out, kernel_size=self.final_pool_size, stride=1)
        final_count_features = F.leaky_relu(self.final_count_feature_layer(final_pool))
        final_count = self.count_layer(final_count_features)
        map1, count1, h1 = self.map_module1(t1_out)
        map2, count2, h2 = self.map_module2(t2_out)
        map3, count3, h3 = self.map_module3(t3_out)
        count = F.add(F.add(F.add(count1, count2), count3), final_count)
        return count, F.cat_channels([map1, map2, map3]), (h1, h2, h3, final_count_features)


class KnnDenseNetCatDggan(KnnDenseNetCat):
    """The dual-goal variant (reference crowd/models.py:903-1046): every count layer has two outputs, the second one is
    the real/fake score, kept in ``real_label`` (B); ``features`` is not produced (the reference never sets it)."""
    COUNT_OUTPUTS = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.real_label = None

    def forward(self, x):
        outputs, map_, _ = self._heads(x)
        batch_size = x.shape[0]
        self.real_label = F.view(F.slice_channels(outputs, 1, 2), (batch_size,))
        return self._density(batch_size, x), F.view(F.slice_channels(outputs, 0, 1), (batch_size,)), map_
