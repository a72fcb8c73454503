"""Oracle model graphs in torch-CPU.  Test infrastructure.

Attribute names, construction order and initialisation calls follow the reference so that
(a) ``state_dict`` keys are interchangeable and (b) a fresh model draws exactly the same random
stream as the reference's constructor.  Sizes the reference hard-codes are generalised exactly as
SURVEY.md §8d prescribes (identical graph at the reference's native size).
"""
import math
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import functional as F

from .functional import seed_all, IRRELEVANT_DATA_MULTIPLIER, OBSERVATION_COUNT


def _pair(value):
    return (value, value) if isinstance(value, int) else tuple(value)


# ------------------------------------------------------------------------------------------- coefficient
class CoefficientGenerator(nn.Module):
    """10 -> h -> h -> h -> 50 MLP, leaky 0.01, no reseed (reference coefficient/models.py:12-28)."""

    def __init__(self, hidden_size=10):
        super().__init__()
        self.input_size = 10
        self.linear1 = nn.Linear(self.input_size, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, hidden_size)
        self.linear4 = nn.Linear(hidden_size, OBSERVATION_COUNT * IRRELEVANT_DATA_MULTIPLIER)

    def forward(self, z):
        h = F.leaky_relu(self.linear1(z))
        h = F.leaky_relu(self.linear2(h))
        h = F.leaky_relu(self.linear3(h))
        return self.linear4(h)


class CoefficientMLP(nn.Module):
    """50 -> h -> h -> h -> out; ``features`` = third hidden layer (reference coefficient/models.py:31-50;
    ``SganMLP`` :75-93 is the same with h=100, out=bins and no feature tap)."""

    def __init__(self, hidden_size=10, outputs=1, tap_features=True):
        super().__init__()
        seed_all(0)
        self.linear1 = nn.Linear(OBSERVATION_COUNT * IRRELEVANT_DATA_MULTIPLIER, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, hidden_size)
        self.linear4 = nn.Linear(hidden_size, outputs)
        self.tap_features = tap_features
        self.features = None

    def forward(self, x):
        h = F.leaky_relu(self.linear1(x))
        h = F.leaky_relu(self.linear2(h))
        h = F.leaky_relu(self.linear3(h))
        if self.tap_features:
            self.features = h
        return self.linear4(h).squeeze()


def coefficient_sgan_mlp(number_of_bins=10):
    return CoefficientMLP(hidden_size=100, outputs=number_of_bins, tap_features=False)


# ------------------------------------------------------------------------------------------- DCGAN
class DCGANGenerator(nn.Module):
    """z -> convT(k = S/16) -> 3x [convT k4 s2 p1 + leaky 0.05] -> convT k4 s2 p1 -> tanh
    (reference age/models.py:32-52 = driving/models.py = crowd/models.py:127-147).  ``image_size`` may be
    (H, W) for the rectangular driving shape."""

    def __init__(self, z_dim=256, image_size=128, conv_dim=64):
        seed_all(0)
        super().__init__()
        height, width = _pair(image_size)
        seed_kernel = (int(height / 16), int(width / 16))
        self.fc = nn.Sequential(nn.ConvTranspose2d(z_dim, conv_dim * 8, seed_kernel, 1, 0))
        self.layer1 = nn.Sequential(nn.ConvTranspose2d(conv_dim * 8, conv_dim * 4, 4, 2, 1))
        self.layer2 = nn.Sequential(nn.ConvTranspose2d(conv_dim * 4, conv_dim * 2, 4, 2, 1))
        self.layer3 = nn.Sequential(nn.ConvTranspose2d(conv_dim * 2, conv_dim, 4, 2, 1))
        self.layer4 = nn.Sequential(nn.ConvTranspose2d(conv_dim, 3, 4, 2, 1))
        self.input_size = z_dim

    def forward(self, z):
        out = self.fc(z.view(z.size(0), z.size(1), 1, 1))
        out = F.leaky_relu(self.layer1(out), 0.05)
        out = F.leaky_relu(self.layer2(out), 0.05)
        out = F.leaky_relu(self.layer3(out), 0.05)
        return torch.tanh(self.layer4(out))


class DCGANDiscriminator(nn.Module):
    """4x [conv k4 s2 p1 + leaky 0.05], features = flatten, conv k = S/16 -> outputs
    (reference age/models.py:55-80)."""

    def __init__(self, image_size=128, conv_dim=64, number_of_outputs=1):
        seed_all(0)
        super().__init__()
        height, width = _pair(image_size)
        self.number_of_outputs = number_of_outputs
        self.layer1 = nn.Sequential(nn.Conv2d(3, conv_dim, 4, 2, 1))
        self.layer2 = nn.Sequential(nn.Conv2d(conv_dim, conv_dim * 2, 4, 2, 1))
        self.layer3 = nn.Sequential(nn.Conv2d(conv_dim * 2, conv_dim * 4, 4, 2, 1))
        self.layer4 = nn.Sequential(nn.Conv2d(conv_dim * 4, conv_dim * 8, 4, 2, 1))
        self.layer5 = nn.Sequential(nn.Conv2d(conv_dim * 8, number_of_outputs,
                                              (int(height / 16), int(width / 16)), 1, 0))
        self.features = None

    def forward(self, x):
        out = F.leaky_relu(self.layer1(x), 0.05)
        out = F.leaky_relu(self.layer2(out), 0.05)
        out = F.leaky_relu(self.layer3(out), 0.05)
        out = F.leaky_relu(self.layer4(out), 0.05)
        self.features = out.view(out.size(0), -1)
        out = self.layer5(out)
        return out.view(-1) if self.number_of_outputs == 1 else out.view(-1, self.number_of_outputs)


# ------------------------------------------------------------------------------------------- VGG
VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class VGG16(nn.Module):
    """VGG-16 with the feature tap after the two FC+ReLU layers and an un-squeezed (B, 1) output
    (reference age/vgg.py:28-53,70-92,151-162).  ``image_size`` generalises Linear(512*7*7, .) to
    512*(S/32)^2; at 224 it is the reference graph."""

    def __init__(self, num_classes=1, image_size=224):
        super().__init__()
        layers, channels = [], 3
        for v in VGG16_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(channels, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                channels = v
        self.feature_layers = nn.Sequential(*layers)
        side = image_size // 32
        self.classifier = nn.Sequential(nn.Linear(512 * side * side, 4096), nn.ReLU(True),
                                        nn.Linear(4096, 4096), nn.ReLU(True))
        self.final_layer = nn.Linear(4096, num_classes)
        self.features = None
        for m in self.modules():  # age/vgg.py:55-67
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def forward(self, x):
        h = self.feature_layers(x)
        h = self.classifier(h.view(h.size(0), -1))
        self.features = h
        return self.final_layer(h)


# ------------------------------------------------------------------------------------------- DenseNet crowd net
class DenseLayer(nn.Sequential):
    """BN-ReLU-conv1x1(4k)-BN-ReLU-conv3x3(k), output concatenated to the input
    (reference crowd/models.py:335-353)."""

    def __init__(self, in_features, growth_rate, bn_size):
        super().__init__()
        self.add_module('norm1', nn.BatchNorm2d(in_features))
        self.add_module('relu1', nn.ReLU(inplace=True))
        self.add_module('conv1', nn.Conv2d(in_features, bn_size * growth_rate, kernel_size=1, stride=1, bias=False))
        self.add_module('norm2', nn.BatchNorm2d(bn_size * growth_rate))
        self.add_module('relu2', nn.ReLU(inplace=True))
        self.add_module('conv2', nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1,
                                           bias=False))

    def forward(self, x):
        return torch.cat([x, super().forward(x)], 1)


def dense_block(num_layers, in_features, bn_size, growth_rate):
    """reference crowd/models.py:356-361."""
    block = nn.Sequential()
    for i in range(num_layers):
        block.add_module('denselayer%d' % (i + 1), DenseLayer(in_features + i * growth_rate, growth_rate, bn_size))
    return block


def transition(in_features, out_features):
    """BN-ReLU-conv1x1-avgpool2 (reference crowd/models.py:364-371)."""
    return nn.Sequential(OrderedDict([('norm', nn.BatchNorm2d(in_features)), ('relu', nn.ReLU(inplace=True)),
                                      ('conv', nn.Conv2d(in_features, out_features, kernel_size=1, stride=1,
                                                         bias=False)),
                                      ('pool', nn.AvgPool2d(kernel_size=2, stride=2))]))


class MapModule(nn.Module):
    """Transposed conv up to the label size, then a 3-conv + 2-"linear" count head, all leaky 0.01
    (reference crowd/models.py:763-786)."""

    def __init__(self, in_features, input_size, label_size):
        super().__init__()
        kernel = label_size // input_size
        self.map_transposed_conv_layer = nn.ConvTranspose2d(in_features, 1, kernel_size=kernel, stride=kernel)
        self.conv1 = nn.Conv2d(1, 8, kernel_size=2, stride=2)
        self.conv2 = nn.Conv2d(8, 16, kernel_size=2, stride=2)
        self.conv3 = nn.Conv2d(16, 32, kernel_size=2, stride=2)
        self.linear1 = nn.Conv2d(32, 20, kernel_size=label_size // 8)
        self.count_layer = nn.Conv2d(20, 1, kernel_size=1)

    def forward(self, x):
        map_ = F.leaky_relu(self.map_transposed_conv_layer(x))
        h = F.leaky_relu(self.conv1(map_))
        h = F.leaky_relu(self.conv2(h))
        h = F.leaky_relu(self.conv3(h))
        h = F.leaky_relu(self.linear1(h))
        return map_, self.count_layer(h), h


class KnnDenseNetCat(nn.Module):
    """DenseNet-201 trunk + three map heads + count head; features = cat[h1, h2, h3, count features]
    (reference crowd/models.py:1049-1166, ``pretrained=False`` branch).  ``image_size`` generalises the
    hard-coded 28/14/7 map inputs and the 7x7 final pool to S/8, S/16, S/32."""

    def __init__(self, growth_rate=32, block_config=(6, 12, 48, 32), num_init_features=64, bn_size=4,
                 image_size=224):
        super().__init__()
        self.label_patch_size = image_size
        self.dense_blocks = nn.ModuleList()
        self.transition_layers = nn.ModuleList()
        self.conv_layer1 = nn.Sequential(OrderedDict([
            ('conv0', nn.Conv2d(3, num_init_features, kernel_size=7, stride=2, padding=3, bias=False)),
            ('norm0', nn.BatchNorm2d(num_init_features)), ('relu0', nn.ReLU(inplace=True)),
            ('pool0', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
        features = num_init_features
        transition_widths = []
        for i, layers in enumerate(block_config):
            self.dense_blocks.add_module('denseblock%d' % (i + 1), dense_block(layers, features, bn_size, growth_rate))
            features += layers * growth_rate
            if i != len(block_config) - 1:
                self.transition_layers.add_module('transition%d' % (i + 1), transition(features, features // 2))
                features //= 2
                transition_widths.append(features)
        self.norm5 = nn.BatchNorm2d(features)
        for m in self.modules():  # crowd/models.py:1094-1101
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self.map_module1 = MapModule(transition_widths[0], image_size // 8, image_size)
        self.map_module2 = MapModule(transition_widths[1], image_size // 16, image_size)
        self.map_module3 = MapModule(transition_widths[2], image_size // 32, image_size)
        self.final_count_feature_layer = nn.Conv2d(features, 20, kernel_size=1)
        self.count_layer = nn.Conv2d(20, 1, kernel_size=1)
        self.final_pool_size = image_size // 32
        self.features = None

    def forward(self, x):
        batch = x.shape[0]
        h = self.conv_layer1(x)
        t1 = self.transition_layers.transition1(self.dense_blocks.denseblock1(h))
        t2 = self.transition_layers.transition2(self.dense_blocks.denseblock2(t1))
        t3 = self.transition_layers.transition3(self.dense_blocks.denseblock3(t2))
        h = F.relu(self.norm5(self.dense_blocks.denseblock4(t3)))
        pooled = F.avg_pool2d(h, kernel_size=self.final_pool_size, stride=1)
        count_features = F.leaky_relu(self.final_count_feature_layer(pooled))
        final_count = self.count_layer(count_features)
        map1, count1, h1 = self.map_module1(t1)
        map2, count2, h2 = self.map_module2(t2)
        map3, count3, h3 = self.map_module3(t3)
        self.features = torch.cat([t.view(batch, -1, 1, 1) for t in (h1, h2, h3, count_features)], dim=1)
        count = (count1 + count2 + count3 + final_count).view(batch)
        maps = torch.cat([map1, map2, map3], dim=1).view(batch, 3, self.label_patch_size, self.label_patch_size)
        density = torch.zeros([batch, self.label_patch_size, self.label_patch_size])
        return density, count, maps
