"""Oracle loss math, RNG helpers and the toy-data generator (torch-CPU / numpy).  Test infrastructure."""
import random
import time

import numpy as np
import torch
from scipy.stats import norm, uniform


# ------------------------------------------------------------------------------------------- RNG
def seed_all(seed=None):
    """Seeds python, numpy and torch in that order (reference utility.py:110-116)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(int(time.time()) if seed is None else seed)


def mixture_rvs(submodels, size):
    """Equal-weight mixture sampling: one randint draw, one full draw per component, then select
    (reference utility.py:102-107).  ``submodels`` are frozen scipy.stats distributions."""
    which = np.random.randint(len(submodels), size=size)
    draws = [submodel.rvs(size=size) for submodel in submodels]
    return np.choose(which, draws)


def discriminator_noise(batch, width, mean_offset=0.0):
    """z for the D-side fake batch: float64 two-Gaussian mixture cast to float32 (srgan.py:286-289)."""
    return torch.tensor(mixture_rvs([norm(-mean_offset, 1), norm(mean_offset, 1)],
                                    size=[batch, width]).astype(np.float32))


# ------------------------------------------------------------------------------------------- toy data
IRRELEVANT_DATA_MULTIPLIER = 5
OBSERVATION_COUNT = 10


def polynomial_examples(count, observations=OBSERVATION_COUNT):
    """x + a2 x^2 + a3 x^3 + a4 x^4 + N(0, 0.1) on 5 irrelevant copies; label = first a3
    (reference coefficient/data.py:31-67)."""
    def double_uniform():
        return mixture_rvs([uniform(-2, 1), uniform(1, 1)],
                           size=[count, IRRELEVANT_DATA_MULTIPLIER, 1]).astype(np.float32)
    a2, a3, a4 = double_uniform(), double_uniform(), double_uniform()
    x = np.linspace(-1, 1, num=observations)
    examples = x + a2 * x ** 2 + a3 * x ** 3 + a4 * x ** 4
    examples = examples.reshape(count, observations * IRRELEVANT_DATA_MULTIPLIER).astype(np.float32)
    examples += np.random.normal(0, 0.1, examples.shape)
    return examples, a3[:, 0, 0]


def toy_dataset(count, seed):
    """ToyDataset arrays for sizes >= batch size (reference coefficient/data.py:13-29)."""
    seed_all(seed)
    return polynomial_examples(count)


# ------------------------------------------------------------------------------------------- distances
def abs_plus_one_log_neg(t):           # utility.py:206-208 (returns a VECTOR; no mean)
    return t.abs().log1p().neg()


def abs_plus_one_log_mean_neg(t):      # utility.py:211-213
    return (t.abs() + 1).log().mean().neg()


def abs_plus_one_sqrt_mean_neg(t):     # utility.py:216-218
    return (t.abs() + 1).sqrt().mean().neg()


def abs_mean_neg(t):                   # utility.py:221-223
    return t.abs().mean().neg()


def abs_mean(t):                       # utility.py:226-228
    return t.abs().mean()


def norm_mean(t):                      # utility.py:236-238
    return t.pow(2).sum().pow(0.5)


def square_mean(t):                    # utility.py:241-243
    return t.pow(2).mean()


def abs_plus_one_square_root(t):       # utility.py:201-203
    return (t.abs() + 1).sqrt()


DISTANCES = {f.__name__: f for f in (abs_plus_one_log_neg, abs_plus_one_log_mean_neg, abs_plus_one_sqrt_mean_neg,
                                     abs_mean_neg, abs_mean, norm_mean, square_mean)}


def batch_mean(features, dp=None):
    """Mean over the (global) batch.  With a data-parallel context the per-rank sums are all-reduced
    (SURVEY.md §8e): the backward of that is the identity on the local sum."""
    if dp is None or dp.world_size == 1:
        return features.mean(0)
    return dp.all_reduce_sum_autograd(features.sum(0)) / dp.global_batch(features.shape[0])


def feature_distance_loss(base, other, distance, normalize=False, dp=None):
    """distance(mean_b(base) - mean_b(other)) (reference srgan.py:438-449), including the reference's
    normalize_feature_norm branch as written (line 447 divides the un-meaned ``other``)."""
    base_mean, other_mean = batch_mean(base, dp), batch_mean(other, dp)
    if normalize:
        eps = 1e-5
        base_mean = base_mean / (base_mean.norm() + eps)
        other_mean = other / (other_mean.norm() + eps)
    return distance(base_mean - other_mean)


def labeled_loss(predicted, labels, order=2):
    """mean(|p - y| ** order) (reference srgan.py:414-417)."""
    return (predicted - labels).abs().pow(order).mean()


def crowd_labeled_loss(predicted, labels, order, map_multiplier):
    """count loss + map_multiplier * map loss (reference crowd/srgan.py:247-254)."""
    heads, knn_map = labels
    _, count, maps = predicted
    map_loss = (maps - knn_map.unsqueeze(1)).abs().mean(1).sum(1).sum(1).pow(order).mean()
    count_loss = (count - heads.sum(1).sum(1)).abs().pow(order).mean()
    return count_loss + map_loss * map_multiplier


# ------------------------------------------------------------------------------------------- SGAN math
def logsumexp(inputs, dim=None, keepdim=False):
    """s + log(sum(exp(x - s))) with s = max (reference utility.py:161-182)."""
    if dim is None:
        inputs, dim = inputs.reshape(-1), 0
    s = inputs.max(dim=dim, keepdim=True)[0]
    out = s + (inputs - s).exp().sum(dim=dim, keepdim=True).log()
    return out if keepdim else out.squeeze(dim)


def real_numbers_to_bin_indexes(reals, bins):
    """Index of the nearest bin centre (reference utility.py:141-144)."""
    return (reals.reshape(-1, 1) - bins.reshape(1, -1)).abs().min(dim=1)[1]


def logits_to_bin_values(logits, bins):
    """Bin centre of the arg-max logit (reference utility.py:147-151)."""
    return bins[logits.max(dim=1)[1]]
