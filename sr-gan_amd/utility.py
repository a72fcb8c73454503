"""Utility surface of the reference's utility.py for the hot path: the device handle, seeding, the
mixture sampler, the bin helpers and the distance functions (as tape operations on HIP kernels).

Logging / plotting / download helpers of the reference file are out of scope."""
import os
import random
import re
import time

import numpy as np
import torch
from scipy.stats import rv_continuous

from . import functional as F
from .tape import Var


def current_device():
    """Per-rank device (the reference's module-global ``gpu`` is cuda:0, utility.py:18; one process per GPU
    here, so the rank's LOCAL_RANK selects it)."""
    if not torch.cuda.is_available():
        raise RuntimeError('srgan_amd needs a ROCm device: the training step has no CPU fallback')
    index = int(os.environ.get('LOCAL_RANK', '0'))
    if index >= torch.cuda.device_count():      # several ranks sharing one device (single-GPU tests over gloo)
        index = torch.cuda.current_device()
    return torch.device('cuda', index)


class _LazyDevice:
    """``utility.gpu`` stand-in that resolves on first use so importing the package needs no GPU."""

    def __getattr__(self, name):
        return getattr(current_device(), name)

    def __repr__(self):
        return repr(current_device())


gpu = _LazyDevice()


class SummaryWriter:
    """Scalar logger with the reference wrapper's interface (utility.py:21-49): implicit ``step``,
    ``summary_period``, ``is_summary_step``.  Records in memory; forwards to tensorboardX when installed."""

    def __init__(self, log_dir=None, comment='', summary_period=1, steps_to_run=-1, **kwargs):
        self.log_dir = log_dir
        self.step = 0
        self.summary_period = summary_period
        self.steps_to_run = steps_to_run
        self.scalars = {}
        self._backend = None
        if log_dir is not None:
            try:
                from tensorboardX import SummaryWriter as Backend
                self._backend = Backend(log_dir=log_dir, comment=comment, **kwargs)
            except ImportError:
                self._backend = None

    def add_scalar(self, tag, scalar_value, global_step=None, **kwargs):
        if global_step is None:
            global_step = self.step
        self.scalars.setdefault(tag, []).append((global_step, float(scalar_value)))
        if self._backend is not None:
            self._backend.add_scalar(tag, scalar_value, global_step, **kwargs)

    def add_histogram(self, *args, **kwargs):
        pass

    def add_image(self, *args, **kwargs):
        pass

    def is_summary_step(self):
        return self.step % self.summary_period == 0 or self.step == self.steps_to_run - 1


def clean_scientific_notation(string):
    """'1.000000e+02' -> '1e2' (reference utility.py:52-57)."""
    string = re.sub(r'\.?0*e([+\-])0*([0-9])', r'e\g<1>\g<2>', string)
    return re.sub(r'e\+', r'e', string)


def shuffled(list_):
    random.seed()
    random.shuffle(list_)
    return list_


class MixtureModel(rv_continuous):
    """Equal-weight mixture of frozen scipy.stats distributions (reference utility.py:89-107)."""

    def __init__(self, submodels, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.submodels = submodels

    def _pdf(self, x, **kwargs):
        return sum(submodel.pdf(x) for submodel in self.submodels) / len(self.submodels)

    def rvs(self, size):
        choices = np.random.randint(len(self.submodels), size=size)
        samples = [submodel.rvs(size=size) for submodel in self.submodels]
        return np.choose(choices, samples)


def seed_all(seed=None):
    """python, numpy, torch -- in the reference's order (utility.py:110-116)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(int(time.time()) if seed is None else seed)


def make_directory_name_unique(trial_directory):
    if os.path.exists(trial_directory):
        run_number = 1
        while os.path.exists(trial_directory + ' r{}'.format(run_number)):
            run_number += 1
        trial_directory += ' r{}'.format(run_number)
    return trial_directory


def to_normalized_range(tensor_):
    return (tensor_ / 127.5) - 1


def to_image_range(tensor_):
    return (tensor_ + 1) * 127.5


def real_numbers_to_bin_indexes(real_numbers, bins):
    """Index of the nearest bin (reference utility.py:141-144); host-side helper for evaluation code.  The
    training step uses the device one-hot form, ``functional.nearest_bin_onehot``."""
    real_numbers = real_numbers.data if isinstance(real_numbers, Var) else real_numbers
    bins = bins.data if isinstance(bins, Var) else bins
    return (real_numbers.reshape(-1, 1) - bins.reshape(1, -1)).abs().min(dim=1)[1]


def logits_to_bin_values(logits, bins):
    logits = logits.data if isinstance(logits, Var) else logits
    bins = bins.data if isinstance(bins, Var) else bins
    return bins[logits.max(dim=1)[1]]


def logsumexp(inputs, dim=None, keepdim=False):
    """Stable log-sum-exp over dim 1 of a [B, K] var (reference utility.py:161-182)."""
    if dim not in (1, -1) or len(inputs.shape) != 2:
        raise NotImplementedError('the training step only reduces [B, bins] logits over dim=1')
    out = F.logsumexp_rows(inputs)
    return F.view(out, (out.shape[0], 1)) if keepdim else out


# ---- distance functions (reference utility.py:201-243); each maps a feature-difference vector to a scalar ------
def abs_plus_one_square_root(tensor):
    return F.sqrt(F.add_scalar(F.abs_(tensor), 1.0))


def abs_plus_one_log_neg(tensor):
    """NB: returns a VECTOR, as in the reference (utility.py:206-208)."""
    return F.neg(F.log1p(F.abs_(tensor)))


def abs_plus_one_log_mean_neg(tensor):
    return F.neg(F.mean_all(F.log(F.add_scalar(F.abs_(tensor), 1.0))))


def abs_plus_one_sqrt_mean_neg(tensor):
    return F.neg(F.mean_all(F.sqrt(F.add_scalar(F.abs_(tensor), 1.0))))


def abs_mean_neg(tensor):
    return F.neg(F.mean_all(F.abs_(tensor)))


def abs_mean(tensor):
    return F.mean_all(F.abs_(tensor))


def norm_squared(tensor, axis=1):
    if axis != 1:
        raise NotImplementedError
    return F.row_sum(F.square(tensor))


def norm_mean(tensor):
    return F.sqrt(F.sum_all(F.square(tensor)))


def square_mean(tensor):
    return F.mean_all(F.square(tensor))
