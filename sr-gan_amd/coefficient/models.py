"""Coefficient-task networks on HIP kernels (surface of reference coefficient/models.py:12-93)."""
from .. import functional as F
from .. import nn
from ..utility import seed_all

observation_count = 10
irrelevant_data_multiplier = 5


class Generator(nn.Module):
    """10 -> h -> h -> h -> 50, leaky_relu 0.01 (reference coefficient/models.py:12-28; no reseed)."""

    def __init__(self, hidden_size=10):
        super().__init__()
        self.input_size = 10
        self.linear1 = nn.Linear(self.input_size, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, hidden_size)
        self.linear4 = nn.Linear(hidden_size, observation_count * irrelevant_data_multiplier)

    def forward(self, z, add_noise=False):
        h = F.leaky_relu(self.linear1(z))
        h = F.leaky_relu(self.linear2(h))
        h = F.leaky_relu(self.linear3(h))
        return self.linear4(h)


class MLP(nn.Module):
    """50 -> h -> h -> h -> 1; ``features`` is the third hidden activation (reference
    coefficient/models.py:31-50)."""

    def __init__(self, hidden_size=10, outputs=1, tap_features=True):
        super().__init__()
        seed_all(0)
        self.linear1 = nn.Linear(observation_count * irrelevant_data_multiplier, hidden_size)
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
        out = self.linear4(h)
        return F.view(out, (out.shape[0],)) if out.shape[1] == 1 else out


class DgganMLP(MLP):
    """50 -> h -> h -> h -> 2: (predicted value, real/fake score) per example (reference coefficient/models.py:53-72)."""

    def __init__(self, hidden_size=10):
        super().__init__(hidden_size, outputs=2)

    def forward(self, x):
        out = super().forward(x)                                   # [B, 2]
        columns = F.view(out, (out.shape[0], 2, 1, 1))
        return (F.view(F.slice_channels(columns, 0, 1), (out.shape[0],)),
                F.view(F.slice_channels(columns, 1, 2), (out.shape[0],)))


class SganMLP(MLP):
    """50 -> 100 -> 100 -> 100 -> bins, no feature tap (reference coefficient/models.py:75-93)."""

    def __init__(self, number_of_bins=10):
        super().__init__(hidden_size=100, outputs=number_of_bins, tap_features=False)
