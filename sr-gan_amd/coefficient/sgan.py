"""SGAN for the coefficient application (surface of reference coefficient/sgan.py:11-23)."""
import torch

from ..sgan import SganExperiment
from .models import SganMLP, Generator
from .srgan import CoefficientExperiment


class CoefficientSganExperiment(SganExperiment, CoefficientExperiment):
    def __init__(self, settings):
        super().__init__(settings)
        self.bins = torch.linspace(-3, 3, self.settings.number_of_bins)

    def model_setup(self):
        self.DNN = SganMLP(self.settings.number_of_bins)
        self.D = SganMLP(self.settings.number_of_bins)
        self.G = Generator()

    def validation_summaries(self, step):
        pass
