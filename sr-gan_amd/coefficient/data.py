"""Polynomial-coefficient toy data (surface of reference coefficient/data.py:13-67).  An example is the concatenation of
``irrelevant_data_multiplier`` = 5 noisy polynomials x + a2 x^2 + a3 x^3 + a4 x^4 sampled at ``observation_count``
points of linspace(-1, 1); the label is the cubic coefficient of the FIRST polynomial.  The random draws are made in the
reference's order (all a2, then a3, then a4, then the noise), so a seed reproduces its dataset."""
import numpy as np
from scipy.stats import uniform

from ..utility import MixtureModel, seed_all

irrelevant_data_multiplier = 5
NOISE_STANDARD_DEVIATION = 0.1


def _coefficients(number_of_examples):
    """One coefficient per (example, polynomial) from the two-interval mixture U[-2, -1] u U[1, 2]."""
    mixture = MixtureModel([uniform(-2, 1), uniform(1, 1)])
    return mixture.rvs(size=[number_of_examples, irrelevant_data_multiplier, 1]).astype(dtype=np.float32)


def generate_double_a2_a3_a4_coefficients(number_of_examples):
    return tuple(_coefficients(number_of_examples) for _ in range(3))


def generate_examples_from_coefficients(a2, a3, a4, number_of_observations):
    x = np.linspace(-1, 1, num=number_of_observations)
    polynomials = x + (a2 * (x ** 2)) + (a3 * (x ** 3)) + (a4 * (x ** 4))              # [examples, 5, observations]
    return polynomials.reshape(polynomials.shape[0], -1).astype(np.float32)


def generate_polynomial_examples(number_of_examples, number_of_observations):
    a2, a3, a4 = generate_double_a2_a3_a4_coefficients(number_of_examples)
    examples = generate_examples_from_coefficients(a2, a3, a4, number_of_observations)
    examples += np.random.normal(0, NOISE_STANDARD_DEVIATION, examples.shape)
    return examples, a3[:, 0, 0]


class ToyDataset:
    """Map-style dataset: ``dataset[i] -> (example f32[5 * observation_count], label f32)``.  A dataset smaller than
    one batch is tiled up to the batch size (the reference passes NumPy a float repeat count there, which NumPy 2
    rejects; the integer part is what it meant)."""

    def __init__(self, dataset_size, observation_count, settings, seed=None):
        seed_all(seed)
        self.examples, self.labels = generate_polynomial_examples(dataset_size, observation_count)
        if len(self.labels) < settings.batch_size:
            repeats = int(settings.batch_size / len(self.labels))
            self.examples, self.labels = (np.repeat(array, repeats, axis=0) for array in (self.examples, self.labels))
        self.length = len(self.labels)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        return self.examples[index], self.labels[index]
