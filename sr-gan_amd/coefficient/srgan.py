"""Coefficient application (surface of reference coefficient/srgan.py:17-38): toy datasets, MLP D / DNN and
generator.  Validation plots (seaborn frames) are out of scope; MAE / MSE summaries are kept."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from .. import functional as F
from ..srgan import Experiment, as_var
from ..tape import no_grad
from .data import ToyDataset
from .models import Generator, MLP, observation_count


class CoefficientExperiment(Experiment):
    """The coefficient application."""

    def dataset_setup(self):
        settings = self.settings
        self.train_dataset = ToyDataset(dataset_size=settings.labeled_dataset_size, observation_count=observation_count,
                                        settings=settings, seed=settings.labeled_dataset_seed)
        self.train_dataset_loader = DataLoader(self.train_dataset, batch_size=settings.batch_size, shuffle=True,
                                               pin_memory=settings.pin_memory)
        self.unlabeled_dataset = ToyDataset(dataset_size=settings.unlabeled_dataset_size,
                                            observation_count=observation_count, settings=settings, seed=100)
        self.unlabeled_dataset_loader = DataLoader(self.unlabeled_dataset, batch_size=settings.batch_size,
                                                   shuffle=True, pin_memory=settings.pin_memory)
        self.validation_dataset = ToyDataset(settings.validation_dataset_size, observation_count, seed=101,
                                             settings=settings)

    def model_setup(self):
        self.DNN = MLP(self.settings.hidden_size)
        self.D = MLP(self.settings.hidden_size)
        self.G = Generator(self.settings.hidden_size)

    def validation_summaries(self, step):
        """MAE / MSE / RMSE of DNN and D on the train and validation sets, and the GAN/DNN ratios
        (reference coefficient/srgan.py:40-101 without the plotting)."""
        dnn_validation = None
        for network, writer in ((self.DNN, self.dnn_summary_writer), (self.D, self.gan_summary_writer)):
            for dataset, name in ((self.train_dataset, '2 Train Error'), (self.validation_dataset, '1 Validation Error')):
                values = self.evaluation_epoch(network, dataset, writer, name,
                                               dnn_validation if (network is self.D and name.startswith('1')) else None)
                if network is self.DNN and name.startswith('1'):
                    dnn_validation = values

    def evaluation_epoch(self, network, dataset, summary_writer, summary_name, comparison_values=None):
        with no_grad():
            predicted = network(as_var(torch.from_numpy(dataset.examples.astype(np.float32)))).cpu().numpy()
        mae = float(np.mean(np.abs(predicted - dataset.labels)))
        mse = float(np.mean(np.power(predicted - dataset.labels, 2)))
        rmse = mse ** 0.5
        summary_writer.add_scalar(f'{summary_name}/MAE', mae)
        summary_writer.add_scalar(f'{summary_name}/MSE', mse)
        summary_writer.add_scalar(f'{summary_name}/RMSE', rmse)
        if comparison_values:
            summary_writer.add_scalar(f'{summary_name}/Ratio MAE GAN DNN', mae / comparison_values['mae'])
            summary_writer.add_scalar(f'{summary_name}/Ratio MSE GAN DNN', mae / comparison_values['mse'])
            summary_writer.add_scalar(f'{summary_name}/Ratio RMSE GAN DNN', rmse / comparison_values['rmse'])
        return dict(mae=mae, mse=mse, rmse=rmse, predicted_labels=predicted)
