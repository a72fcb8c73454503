"""Shared helpers for the test-suite (golden loading, tolerances)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def golden_state(golden, prefix):
    """The state dict stored under ``prefix/`` as torch tensors."""
    prefix = prefix.rstrip('/') + '/'
    return {k[len(prefix):]: torch.from_numpy(np.array(golden[k])) for k in golden.files if k.startswith(prefix)}


def golden_scalars(golden, step):
    prefix = f's{step}/'
    return {k[len(prefix):]: float(golden[k]) for k in golden.files
            if k.startswith(prefix) and golden[k].ndim == 0}


def make_settings(**overrides):
    """A bare settings object with the reference's defaults for the attributes the step reads
    (reference settings.py:14-67)."""
    from oracle import functional as OF
    values = dict(batch_size=1000, learning_rate=1e-4, weight_decay=0, labeled_loss_multiplier=1.0,
                  matching_loss_multiplier=1.0, contrasting_loss_multiplier=1.0, srgan_loss_multiplier=1.0,
                  gradient_penalty_multiplier=1e1, mean_offset=0, labeled_loss_order=2,
                  generator_training_step_period=1, normalize_feature_norm=False,
                  contrasting_distance_function=OF.abs_plus_one_sqrt_mean_neg,
                  matching_distance_function=OF.abs_mean, map_multiplier=1e-6, hidden_size=10, number_of_bins=10)
    values.update(overrides)
    return SimpleNamespace(**values)


def checksum(t):
    t64 = t.detach().double().reshape(-1).cpu()
    return np.array([t64.sum().item(), t64.abs().sum().item(), t64[0].item(), t64[-1].item()])


def assert_close(actual, expected, rtol=1e-3, atol=0.0, what=''):
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, f'{what}: shape {actual.shape} != {expected.shape}'
    scale = np.maximum(np.abs(expected), 1e-30)
    err = np.abs(actual - expected)
    bad = err > (atol + rtol * scale)
    assert not bad.any(), (f'{what}: {bad.sum()} of {bad.size} outside rtol={rtol} atol={atol}; '
                           f'max abs err {err.max():.3e}, max |expected| {np.abs(expected).max():.3e}')


def assert_close_norm(actual, expected, rtol=1e-3, what=''):
    """Relative error measured against the tensor's max magnitude (for tensors with near-zero entries)."""
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, f'{what}: shape {actual.shape} != {expected.shape}'
    denom = max(np.abs(expected).max(), 1e-30)
    err = np.abs(actual - expected).max() / denom
    assert err <= rtol, f'{what}: max err / max|expected| = {err:.3e} > {rtol}'
