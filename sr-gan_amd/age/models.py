"""DCGAN generator / discriminator on HIP kernels (surface of reference age/models.py:32-80, identical to
driving/models.py and, for the generator, crowd/models.py:127-147): same module names, construction order (hence the same
initial weights from ``seed_all(0)``) and ``state_dict`` keys.  ``image_size`` may be an (H, W) pair for the rectangular
driving frames; a square int is the reference graph."""
from .. import functional as F
from .. import nn
from ..utility import seed_all

batch_norm = False            # the reference's module-level switch (age/models.py:13): no batch-norm anywhere
LEAK = 0.05


def _blocked_stack_ok(network):
    """The 16-bit path has the reference configuration: no batch-norm inside the stages (age/models.py:13)."""
    return all(len(stage) == 1 for name, stage in network.named_children())


def _pair(value):
    return (value, value) if isinstance(value, int) else tuple(value)


def _stage(layer_class, c_in, c_out, k_size, stride, pad, bn):
    """``Sequential(layer[, BatchNorm2d])`` -- the reference wraps every layer like this, so keys read ``layerN.0.weight``."""
    return nn.Sequential(*([layer_class(c_in, c_out, k_size, stride, pad)] + ([nn.BatchNorm2d(c_out)] if bn else [])))


def transpose_convolution(c_in, c_out, k_size, stride=2, pad=1, bn=batch_norm):
    return _stage(nn.ConvTranspose2d, c_in, c_out, k_size, stride, pad, bn)


def convolution(c_in, c_out, k_size, stride=2, pad=1, bn=batch_norm):
    return _stage(nn.Conv2d, c_in, c_out, k_size, stride, pad, bn)


def _seed_kernel(image_size):
    height, width = _pair(image_size)
    return int(height / 16), int(width / 16)      # four stride-2 stages between the 1x1 code and the image


class Generator(nn.Module):
    """z -> ``fc``: convT(k = S/16) -> ``layer1..3``: convT k4 s2 p1 + leaky 0.05 -> ``layer4``: convT k4 s2 p1 -> tanh."""

    def __init__(self, z_dim=256, image_size=128, conv_dim=64):
        seed_all(0)
        super().__init__()
        widths = (conv_dim * 8, conv_dim * 4, conv_dim * 2, conv_dim, 3)
        self.fc = transpose_convolution(z_dim, widths[0], _seed_kernel(image_size), 1, 0, bn=False)
        for index in range(1, 5):
            setattr(self, f'layer{index}', transpose_convolution(widths[index - 1], widths[index], 4,
                                                                 **({'bn': False} if index == 4 else {})))
        self.input_size = z_dim

    def forward(self, z):
        if F.STORAGE_DTYPE and z.meta is None and _blocked_stack_ok(self):
            from ..blocked16 import active_code
            return self._forward_blocked(z, active_code())
        out = self.fc(F.view(z, (z.shape[0], z.shape[1], 1, 1)))
        for stage in (self.layer1, self.layer2, self.layer3):
            out = F.leaky_relu(stage(out), LEAK)
        return F.tanh(self.layer4(out))

    def _forward_blocked(self, z, code):
        """The same graph on the 16-bit data path (``blocked16``): the code and every feature map bf16 / fp16 in the blocked
        layout, ``leaky_relu(convT(x))`` one kernel per stage, fp32 again for the images (tanh on the 3-channel result)."""
        from .. import blocked16 as B
        if code == 0:       # fp32 blocked: the seed layer on the fp32 kernels (one GEMM over its 26-137 M weights), then the stages
            h = B.pack(self.fc(F.view(z, (z.shape[0], z.shape[1], 1, 1))), 0)
        else:
            h = B.seed_conv_transpose(B.pack(F.view(z, (z.shape[0], z.shape[1])), code), self.fc[0])
        for stage in (self.layer1, self.layer2, self.layer3):
            h = B.conv_transpose4x4s2(h, stage[0], slope=LEAK)
        return F.tanh(B.unpack(B.conv_transpose4x4s2(h, self.layer4[0])))


class Discriminator(nn.Module):
    """``layer1..4``: conv k4 s2 p1 + leaky 0.05; ``features`` = the flattened result; ``layer5``: conv k = S/16 -> outputs."""

    def __init__(self, image_size=128, conv_dim=64, number_of_outputs=1):
        seed_all(0)
        super().__init__()
        self.number_of_outputs = number_of_outputs
        widths = (3, conv_dim, conv_dim * 2, conv_dim * 4, conv_dim * 8)
        for index in range(1, 5):
            setattr(self, f'layer{index}', convolution(widths[index - 1], widths[index], 4,
                                                       **({'bn': False} if index == 1 else {})))
        self.layer5 = convolution(widths[4], number_of_outputs, _seed_kernel(image_size), 1, 0, False)
        self.features = None

    def forward(self, x):
        if F.STORAGE_DTYPE and x.meta is None and _blocked_stack_ok(self):
            from ..blocked16 import active_code
            return self._forward_blocked(x, active_code())
        out = x
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            out = F.leaky_relu(stage(out), LEAK)
        self.features = F.flatten2d(out)
        scores = self.layer5(out)
        return F.view(scores, (-1,) if self.number_of_outputs == 1 else (-1, self.number_of_outputs))

    def _forward_blocked(self, x, code):
        """The same graph on the 16-bit data path (``blocked16``): every ``leaky_relu(conv(x))`` stage one kernel on bf16 / fp16
        tensors in the blocked layout; ``features`` (reference order: the NCHW flattening) and the scores leave as fp32."""
        from .. import blocked16 as B
        h = B.pack(x, code)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            h = B.conv4x4s2(h, stage[0], slope=LEAK)
        trunk = B.unpack(h)
        self.features = F.flatten2d(trunk)
        # the full-plane convolution = a linear map over the flattened trunk: 16-bit through the blocked order's shadow, fp32
        # blocked through the fp32 kernels on the unpacked trunk
        scores = self.layer5(trunk) if code == 0 else B.unpack(B.linear(B.flatten(h), self.layer5[0]))
        return F.view(scores, (-1,) if self.number_of_outputs == 1 else (-1, self.number_of_outputs))
