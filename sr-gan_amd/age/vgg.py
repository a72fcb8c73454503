"""VGG-16 discriminator on HIP kernels (surface of reference age/vgg.py:28-53,70-92,151-162).

``image_size`` generalises the classifier's ``512*7*7`` input to ``512*(S/32)**2``;
at 224 it is the reference graph.  The (B, 1) un-squeezed output is kept as in the reference (Appendix A.12).
Pretrained weights are never downloaded: ``pretrained`` takes a state dict / path (see ``vgg16``)."""
import math

from torch import nn as torch_nn

from .. import functional as F
from .. import nn

cfg = {
    'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'B': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'D': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'E': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
}


def make_layers(cfg_, batch_norm=False):
    layers, in_channels = [], 3
    for v in cfg_:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)
            layers += [conv2d, nn.BatchNorm2d(v), nn.ReLU(inplace=True)] if batch_norm else [conv2d,
                                                                                              nn.ReLU(inplace=True)]
            in_channels = v
    return nn.Sequential(*layers)


class VGG(nn.Module):
    def __init__(self, feature_layers, num_classes=1000, init_weights=True, image_size=224):
        super().__init__()
        self.feature_layers = feature_layers
        side = image_size // 32
        self.classifier = nn.Sequential(nn.Linear(512 * side * side, 4096), nn.ReLU(True),
                                        nn.Linear(4096, 4096), nn.ReLU(True))
        self.final_layer = nn.Linear(4096, num_classes)
        self.features = None
        if init_weights:
            self._initialize_weights()

    def forward(self, x):
        if F.STORAGE_DTYPE in (1, 2) and x.meta is None:           # (the 3x3 / pooling / linear kernels of the blocked path are 16-bit)
            return self._forward_blocked(x, F.STORAGE_DTYPE)
        h = self.feature_layers(x)
        h = self.classifier(F.flatten2d(h))
        self.features = h
        return self.final_layer(h)

    def _forward_blocked(self, x, code):
        """The same graph on the 16-bit data path (``blocked16``): bf16 / fp16 activations in the blocked layout, every
        ``conv -> ReLU`` / ``Linear -> ReLU`` pair one kernel, fp32 only at the two ends (the images; ``features`` and the
        prediction)."""
        from .. import blocked16 as B
        h = B.pack(x, code)
        layers = list(self.feature_layers)
        index = 0
        while index < len(layers):
            layer = layers[index]
            if isinstance(layer, torch_nn.Conv2d):
                fused = index + 1 < len(layers) and isinstance(layers[index + 1], nn.ReLU)
                h = B.conv3x3(h, layer, slope=0.0 if fused else None)
                index += 2 if fused else 1
            elif isinstance(layer, nn.MaxPool2d):
                if (layer.kernel_size, layer.stride, layer.padding) != (2, 2, 0):
                    raise NotImplementedError('16-bit path: 2x2 / stride 2 max-pooling')
                h = B.max_pool2(h)
                index += 1
            else:
                raise NotImplementedError(f'16-bit path: {type(layer).__name__} inside the VGG feature stack')
        h = B.flatten(h)
        first, _, second, _ = self.classifier
        h = B.linear(B.linear(h, first, slope=0.0), second, slope=0.0)
        self.features = B.unpack(h)
        return B.unpack(B.linear(h, self.final_layer))

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, torch_nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, torch_nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, torch_nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()


VGG16_WEIGHTS_ENV = 'SRGAN_VGG16_WEIGHTS'


def vgg16(pretrained=False, **kwargs):
    """reference age/vgg.py:151-162.  ``pretrained``: a torchvision vgg16 state dict, a path to one, or True = the path
    in SRGAN_VGG16_WEIGHTS (no download here).  As in the reference the dict is applied with ``strict=False`` to
    modules named ``feature_layers`` / ``classifier`` / ``final_layer``, so of torchvision's keys (``features.*``,
    ``classifier.{0,3,6}.*``) only ``classifier.0`` lands -- a reference quirk reproduced as is."""
    import os
    if pretrained:
        kwargs['init_weights'] = False
    model = VGG(make_layers(cfg['D']), **kwargs)
    if pretrained:
        source = pretrained
        if source is True:
            source = os.environ.get(VGG16_WEIGHTS_ENV)
            if not source:
                raise RuntimeError(f'pretrained=True needs the torchvision vgg16 state dict on disk: set {VGG16_WEIGHTS_ENV}')
        if isinstance(source, (str, os.PathLike)):
            import torch
            source = torch.load(source, map_location='cpu')
        model.load_state_dict(source, strict=False)
    return model
