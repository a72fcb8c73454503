"""SGAN for the age application (surface of reference age/sgan.py:10-26): the discriminators classify the age into
``settings.number_of_bins`` bins spread evenly over [10, 95] years."""
import torch

from ..sgan import SganExperiment
from .models import Generator, Discriminator
from .srgan import AgeExperiment

AGE_RANGE = (10, 95)
BIN_LOGITS = 10                       # the reference builds its discriminators with 10 outputs whatever the bin count


class AgeSganExperiment(SganExperiment, AgeExperiment):
    def __init__(self, settings):
        super().__init__(settings)
        self.bins = torch.linspace(*AGE_RANGE, settings.number_of_bins)

    def model_setup(self):
        size = self._size()
        self.G = Generator(image_size=size)
        self.D, self.DNN = (Discriminator(image_size=size, number_of_outputs=BIN_LOGITS) for _ in range(2))
