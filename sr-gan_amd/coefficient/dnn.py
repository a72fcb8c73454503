"""DNN-only coefficient experiment: ``DnnExperiment`` mixed into the coefficient application (additive -- the
reference wires the DNN method for the crowd application only, run.py:56; the mix-in pattern is the same)."""
from ..dnn import DnnExperiment
from .models import MLP
from .srgan import CoefficientExperiment


class CoefficientDnnExperiment(DnnExperiment, CoefficientExperiment):
    def model_setup(self):
        self.DNN = MLP(self.settings.hidden_size)

    def validation_summaries(self, step):
        for dataset, name in ((self.train_dataset, '2 Train Error'), (self.validation_dataset, '1 Validation Error')):
            self.evaluation_epoch(self.DNN, dataset, self.dnn_summary_writer, name)
