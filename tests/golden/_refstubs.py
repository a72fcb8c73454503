"""Stub modules that let the upstream reference import in a container lacking its optional deps.

Test infrastructure only (golden generation). Registers fake ``tensorboardX``, ``imageio``,
``torchvision``, ``seaborn``, ``recordclass``, ``skimage``, ``patoolib`` and ``mtcnn`` modules in
``sys.modules`` (recipe: SURVEY.md Appendix B) and puts ``/root/reference`` on ``sys.path``.
Nothing here is shipped to, or used on, the GPU box.
"""
import sys
import types

REFERENCE_ROOT = '/root/reference'


class RecordingSummaryWriter:
    """Fake tensorboardX.SummaryWriter that records every scalar in a dict of lists."""

    def __init__(self, log_dir=None, comment='', **kwargs):
        self.log_dir = log_dir
        self.scalars = {}

    def add_scalar(self, tag, scalar_value, global_step=None, **kwargs):
        self.scalars.setdefault(tag, []).append((global_step, float(scalar_value)))

    def add_histogram(self, *args, **kwargs):
        pass

    def add_image(self, *args, **kwargs):
        pass


def _module(name, **attrs):
    module = types.ModuleType(name)
    for key, value in attrs.items():
        setattr(module, key, value)
    sys.modules[name] = module
    return module


class _Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, example):
        for transform in self.transforms:
            example = transform(example)
        return example


def install():
    """Install the stubs and make the reference importable."""
    if 'tensorboardX' not in sys.modules:
        _module('tensorboardX', SummaryWriter=RecordingSummaryWriter)
    for name in ('imageio', 'patoolib', 'mtcnn'):
        if name not in sys.modules:
            _module(name)
    if 'seaborn' not in sys.modules:
        _module('seaborn', set=lambda *a, **k: None, set_style=lambda *a, **k: None)
    if 'recordclass' not in sys.modules:
        _module('recordclass', RecordClass=type('RecordClass', (), {}))
    if 'skimage' not in sys.modules:
        skimage = _module('skimage')
        skimage.transform = _module('skimage.transform')
        skimage.color = _module('skimage.color')
    if 'torchvision' not in sys.modules:
        torchvision = _module('torchvision')
        torchvision.models = _module('torchvision.models')
        torchvision.models.densenet = _module('torchvision.models.densenet',
                                              model_urls={'densenet201': 'unavailable://no-network'})
        torchvision.utils = _module('torchvision.utils', make_grid=lambda *a, **k: None)
        torchvision.transforms = _module('torchvision.transforms', Compose=_Compose)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
