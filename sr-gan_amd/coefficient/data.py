"""Polynomial-coefficient toy data (surface of reference coefficient/data.py:13-67): each example is 5 x 10
samples of x + a2 x^2 + a3 x^3 + a4 x^4 + N(0, 0.1) on linspace(-1, 1, 10); the label is the first a3."""
import numpy as np
from scipy.stats import uniform

from ..utility import MixtureModel, seed_all

irrelevant_data_multiplier = 5


class ToyDataset:
    """The polynomial estimation dataset (map-style: ``dataset[i] -> (example f32[50], label f32)``)."""

    def __init__(self, dataset_size, observation_count, settings, seed=None):
        seed_all(seed)
        self.examples, self.labels = generate_polynomial_examples(dataset_size, observation_count)
        if self.labels.shape[0] < settings.batch_size:
            repeats = int(settings.batch_size / self.labels.shape[0])   # the reference passes a float (NumPy 2 rejects it)
            self.examples = np.repeat(self.examples, repeats, axis=0)
            self.labels = np.repeat(self.labels, repeats, axis=0)
        self.length = self.labels.shape[0]

    def __getitem__(self, index):
        return self.examples[index], self.labels[index]

    def __len__(self):
        return self.length


def generate_polynomial_examples(number_of_examples, number_of_observations):
    a2, a3, a4 = generate_double_a2_a3_a4_coefficients(number_of_examples)
    examples = generate_examples_from_coefficients(a2, a3, a4, number_of_observations)
    examples += np.random.normal(0, 0.1, examples.shape)
    return examples, np.squeeze(a3[:, 0], axis=-1)


def _double_uniform(number_of_examples):
    distribution = MixtureModel([uniform(-2, 1), uniform(1, 1)])
    return distribution.rvs(size=[number_of_examples, irrelevant_data_multiplier, 1]).astype(dtype=np.float32)


def generate_double_a2_a3_a4_coefficients(number_of_examples):
    """a2, a3, a4 each from the two-interval uniform mixture on [-2,-1] u [1,2], drawn in that order."""
    return _double_uniform(number_of_examples), _double_uniform(number_of_examples), _double_uniform(number_of_examples)


def generate_examples_from_coefficients(a2, a3, a4, number_of_observations):
    x = np.linspace(-1, 1, num=number_of_observations)
    examples = x + (a2 * (x ** 2)) + (a3 * (x ** 3)) + (a4 * (x ** 4))
    return examples.reshape(examples.shape[0], number_of_observations * irrelevant_data_multiplier).astype(np.float32)
