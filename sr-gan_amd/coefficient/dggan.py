"""Dual-goal regression GAN on the coefficient task (surface of reference coefficient/dggan.py:13-64):
the discriminator predicts the value AND a real/fake score; the unsupervised terms are binary cross-entropies on the
score (unlabeled examples against 0, generated ones against 1 -- the reference's convention), the gradient penalty is
taken on the per-example scores of the interpolates, and the generator maximises the discriminator's confusion."""
from .. import functional as F
from .. import nn
from ..sgan import bce_with_logits
from .models import DgganMLP, Generator
from .srgan import CoefficientExperiment


class CoefficientDgganExperiment(CoefficientExperiment):
    """A dual goal regression GAN experiment."""

    def model_setup(self):
        self.DNN = DgganMLP(self.settings.hidden_size)
        self.D = DgganMLP(self.settings.hidden_size)
        self.G = Generator(self.settings.hidden_size)

    def dnn_loss_calculation(self, labeled_examples, labels):
        predicted_labels, _ = self.DNN(labeled_examples)
        loss = self.labeled_loss_function(predicted_labels, labels, order=self.settings.labeled_loss_order)
        return F.scale(loss, self.settings.labeled_loss_multiplier)

    def labeled_loss_calculation(self, labeled_examples, labels):
        predicted_labels, _ = self.D(labeled_examples)
        loss = self.labeled_loss_function(predicted_labels, labels, order=self.settings.labeled_loss_order)
        return F.scale(loss, self.settings.labeled_loss_multiplier)

    def unlabeled_loss_calculation(self, labeled_examples, unlabeled_examples):
        _, scores = self.D(unlabeled_examples)
        loss = bce_with_logits(scores, 0.0, self.batch_mean_of_examples)
        return F.scale(loss, self.settings.matching_loss_multiplier * self.settings.dggan_loss_multiplier)

    def fake_loss_calculation(self, unlabeled_examples, fake_examples):
        _, scores = self.D(fake_examples.detach())
        loss = bce_with_logits(scores, 1.0, self.batch_mean_of_examples)
        return F.scale(loss, self.settings.contrasting_loss_multiplier * self.settings.dggan_loss_multiplier)

    def discriminator_losses_shared_forwards(self, labeled_examples, labels, unlabeled_examples, fake_examples):
        # one forward per batch already: the reference order is the shared one
        return (self.labeled_loss_calculation(labeled_examples, labels),
                self.unlabeled_loss_calculation(labeled_examples, unlabeled_examples),
                self.fake_loss_calculation(unlabeled_examples, fake_examples))

    def interpolate_loss_calculation(self, interpolates):
        _, scores = self.D(interpolates)
        return scores

    def generator_loss_calculation(self, fake_examples, _):
        with nn.frozen_parameters(self.D):
            _, scores = self.D(fake_examples)
        return bce_with_logits(scores, 0.0, self.batch_mean_of_examples)

    def validation_summaries(self, step):
        """MAE / MSE of the value heads (the distribution plots of the reference are out of scope)."""
        dnn_validation = None
        for network, writer in ((self.DNN, self.dnn_summary_writer), (self.D, self.gan_summary_writer)):
            value_head = lambda examples, network=network: network(examples)[0]
            for dataset, name in ((self.train_dataset, '2 Train Error'), (self.validation_dataset, '1 Validation Error')):
                values = self.evaluation_epoch(value_head, dataset, writer, name,
                                               dnn_validation if (network is self.D and name.startswith('1')) else None)
                if network is self.DNN and name.startswith('1'):
                    dnn_validation = values
