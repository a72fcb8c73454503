"""Coefficient-task networks on HIP kernels (surface of reference coefficient/models.py:12-93): four-layer perceptrons
whose module names (``linear1`` .. ``linear4``), construction order and ``state_dict`` keys are the reference's."""
from .. import functional as F
from .. import nn
from ..utility import seed_all

observation_count = 10
irrelevant_data_multiplier = 5
EXAMPLE_WIDTH = observation_count * irrelevant_data_multiplier          # 50 numbers per example


class _Perceptron(nn.Module):
    """width_in -> h -> h -> h -> width_out with leaky_relu(0.01) between; ``hidden(x)`` is the third hidden layer."""

    def _build(self, width_in, hidden_size, width_out):
        widths = (width_in, hidden_size, hidden_size, hidden_size, width_out)
        for index in range(4):
            setattr(self, f'linear{index + 1}', nn.Linear(widths[index], widths[index + 1]))

    def hidden(self, x):
        for layer in (self.linear1, self.linear2, self.linear3):
            x = F.leaky_relu(layer(x))
        return x


class Generator(_Perceptron):
    """z[10] -> example[50] (reference coefficient/models.py:12-28; the only network that does not reseed)."""

    def __init__(self, hidden_size=10):
        super().__init__()
        self.input_size = 10
        self._build(self.input_size, hidden_size, EXAMPLE_WIDTH)

    def forward(self, z, add_noise=False):
        return self.linear4(self.hidden(z))


class MLP(_Perceptron):
    """example[50] -> value; ``features`` = the third hidden activation (reference coefficient/models.py:31-50).
    A single output is squeezed to shape (B)."""

    def __init__(self, hidden_size=10, outputs=1, tap_features=True):
        super().__init__()
        seed_all(0)
        self._build(EXAMPLE_WIDTH, hidden_size, outputs)
        self.tap_features = tap_features
        self.features = None

    def forward(self, x):
        h = self.hidden(x)
        if self.tap_features:
            self.features = h
        out = self.linear4(h)
        return F.view(out, (out.shape[0],)) if out.shape[1] == 1 else out


class DgganMLP(MLP):
    """example[50] -> (value, real/fake score), each of shape (B) (reference coefficient/models.py:53-72)."""

    def __init__(self, hidden_size=10):
        super().__init__(hidden_size, outputs=2)

    def forward(self, x):
        out = super().forward(x)                                   # [B, 2]
        batch = out.shape[0]
        columns = F.view(out, (batch, 2, 1, 1))
        return tuple(F.view(F.slice_channels(columns, column, column + 1), (batch,)) for column in (0, 1))


class SganMLP(MLP):
    """example[50] -> bin logits through 100-wide layers, no feature tap (reference coefficient/models.py:75-93)."""

    def __init__(self, number_of_bins=10):
        super().__init__(hidden_size=100, outputs=number_of_bins, tap_features=False)
