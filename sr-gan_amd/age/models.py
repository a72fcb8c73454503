"""DCGAN generator / discriminator on HIP kernels (surface of reference age/models.py:32-80, identical to
driving/models.py and, for the generator, crowd/models.py:127-147).  ``image_size`` may be an (H, W) pair
for the rectangular driving frames; a square int is the reference graph."""
from .. import functional as F
from .. import nn
from ..utility import seed_all

batch_norm = False


def _pair(value):
    return (value, value) if isinstance(value, int) else tuple(value)


def transpose_convolution(c_in, c_out, k_size, stride=2, pad=1, bn=batch_norm):
    layers = [nn.ConvTranspose2d(c_in, c_out, k_size, stride, pad)]
    if bn:
        layers.append(nn.BatchNorm2d(c_out))
    return nn.Sequential(*layers)


def convolution(c_in, c_out, k_size, stride=2, pad=1, bn=batch_norm):
    layers = [nn.Conv2d(c_in, c_out, k_size, stride, pad)]
    if bn:
        layers.append(nn.BatchNorm2d(c_out))
    return nn.Sequential(*layers)


class Generator(nn.Module):
    """z -> convT(k = S/16) -> 3x [convT k4 s2 p1, leaky 0.05] -> convT k4 s2 p1 -> tanh."""

    def __init__(self, z_dim=256, image_size=128, conv_dim=64):
        seed_all(0)
        super().__init__()
        height, width = _pair(image_size)
        self.fc = transpose_convolution(z_dim, conv_dim * 8, (int(height / 16), int(width / 16)), 1, 0, bn=False)
        self.layer1 = transpose_convolution(conv_dim * 8, conv_dim * 4, 4)
        self.layer2 = transpose_convolution(conv_dim * 4, conv_dim * 2, 4)
        self.layer3 = transpose_convolution(conv_dim * 2, conv_dim, 4)
        self.layer4 = transpose_convolution(conv_dim, 3, 4, bn=False)
        self.input_size = z_dim

    def forward(self, z):
        out = self.fc(F.view(z, (z.shape[0], z.shape[1], 1, 1)))
        out = F.leaky_relu(self.layer1(out), 0.05)
        out = F.leaky_relu(self.layer2(out), 0.05)
        out = F.leaky_relu(self.layer3(out), 0.05)
        return F.tanh(self.layer4(out))


class Discriminator(nn.Module):
    """4x [conv k4 s2 p1, leaky 0.05]; ``features`` = flattened activations; conv k = S/16 -> outputs."""

    def __init__(self, image_size=128, conv_dim=64, number_of_outputs=1):
        seed_all(0)
        super().__init__()
        height, width = _pair(image_size)
        self.number_of_outputs = number_of_outputs
        self.layer1 = convolution(3, conv_dim, 4, bn=False)
        self.layer2 = convolution(conv_dim, conv_dim * 2, 4)
        self.layer3 = convolution(conv_dim * 2, conv_dim * 4, 4)
        self.layer4 = convolution(conv_dim * 4, conv_dim * 8, 4)
        self.layer5 = convolution(conv_dim * 8, number_of_outputs, (int(height / 16), int(width / 16)), 1, 0, False)
        self.features = None

    def forward(self, x):
        out = F.leaky_relu(self.layer1(x), 0.05)
        out = F.leaky_relu(self.layer2(out), 0.05)
        out = F.leaky_relu(self.layer3(out), 0.05)
        out = F.leaky_relu(self.layer4(out), 0.05)
        self.features = F.flatten2d(out)
        out = self.layer5(out)
        if self.number_of_outputs == 1:
            return F.view(out, (-1,))
        return F.view(out, (-1, self.number_of_outputs))
