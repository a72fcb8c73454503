"""DNN-only crowd experiment (reference crowd/dnn.py:20): ``DnnExperiment`` mixed into the crowd application."""
from ..dnn import DnnExperiment
from .models import KnnDenseNetCat
from .srgan import CrowdExperiment


class CrowdDnnExperiment(DnnExperiment, CrowdExperiment):
    """The DNN-only version of the crowd application."""

    def model_setup(self):
        self.DNN = KnnDenseNetCat(image_size=self.settings.image_patch_size)
