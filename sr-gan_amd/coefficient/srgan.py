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
        """Labeled / unlabeled / validation toy datasets with the reference's seeds (labeled: the settings' seed, unlabeled
        100, validation 101; reference coefficient/srgan.py:20-33); the two training sets are shuffled each epoch."""
        settings = self.settings

        def toy(size, seed):
            return ToyDataset(dataset_size=size, observation_count=observation_count, settings=settings, seed=seed)

        def shuffled(dataset):
            return DataLoader(dataset, batch_size=settings.batch_size, shuffle=True, pin_memory=settings.pin_memory)
        self.train_dataset = toy(settings.labeled_dataset_size, settings.labeled_dataset_seed)
        self.train_dataset_loader = shuffled(self.train_dataset)
        self.unlabeled_dataset = toy(settings.unlabeled_dataset_size, 100)
        self.unlabeled_dataset_loader = shuffled(self.unlabeled_dataset)
        self.validation_dataset = toy(settings.validation_dataset_size, 101)

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
        errors = predicted - dataset.labels
        values = dict(mae=float(np.mean(np.abs(errors))), mse=float(np.mean(np.power(errors, 2))))
        values['rmse'] = values['mse'] ** 0.5
        for key, value in values.items():
            summary_writer.add_scalar(f'{summary_name}/{key.upper()}', value)
        if comparison_values:
            # (the reference's "Ratio MSE" divides the MAE by the DNN's MSE, coefficient/srgan.py:96; kept as it logs it)
            for key, numerator in (('mae', 'mae'), ('mse', 'mae'), ('rmse', 'rmse')):
                summary_writer.add_scalar(f'{summary_name}/Ratio {key.upper()} GAN DNN',
                                          values[numerator] / comparison_values[key])
        values['predicted_labels'] = predicted
        return values
