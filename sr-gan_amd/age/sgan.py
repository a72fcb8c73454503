"""SGAN for the age application: 10 bins on [10, 95] (surface of reference age/sgan.py:10-26)."""
import torch

from ..sgan import SganExperiment
from .models import Generator, Discriminator
from .srgan import AgeExperiment


class AgeSganExperiment(SganExperiment, AgeExperiment):
    def __init__(self, settings):
        super().__init__(settings)
        self.bins = torch.linspace(10, 95, settings.number_of_bins)

    def model_setup(self):
        size = self._size()
        self.G = Generator(image_size=size)
        self.D = Discriminator(image_size=size, number_of_outputs=10)
        self.DNN = Discriminator(image_size=size, number_of_outputs=10)
