"""Dual-goal regression GAN on the crowd task (surface of reference crowd/dggan.py:9-49): the
discriminator (``KnnDenseNetCatDggan``) produces the count, the maps AND a real/fake score per image; the unsupervised
terms are binary cross-entropies on the score (unlabeled examples against 0, generated ones against 1 -- the
reference's convention), the gradient penalty is taken on the per-example scores of the interpolates, and the generator
drives its samples' scores towards 0."""
from .. import functional as F
from .. import nn
from ..sgan import bce_with_logits
from .models import DCGenerator, KnnDenseNetCatDggan
from .srgan import CrowdExperiment


class CrowdDgganExperiment(CrowdExperiment):
    """The DGGAN crowd experiment."""

    def model_setup(self):
        size = self.settings.image_patch_size
        self.G = DCGenerator(image_size=size)
        self.D = KnnDenseNetCatDggan(image_size=size)
        self.DNN = KnnDenseNetCatDggan(image_size=size)

    def _scores(self, examples):
        _ = self.D(examples)
        return self.D.real_label

    def unlabeled_loss_calculation(self, labeled_examples, unlabeled_examples):
        loss = bce_with_logits(self._scores(unlabeled_examples), 0.0, self.batch_mean_of_examples)
        return F.scale(loss, self.settings.matching_loss_multiplier * self.settings.dggan_loss_multiplier)

    def fake_loss_calculation(self, unlabeled_examples, fake_examples):
        loss = bce_with_logits(self._scores(fake_examples.detach()), 1.0, self.batch_mean_of_examples)
        return F.scale(loss, self.settings.contrasting_loss_multiplier * self.settings.dggan_loss_multiplier)

    def discriminator_losses_shared_forwards(self, labeled_examples, labels, unlabeled_examples, fake_examples):
        # one forward per batch already: the reference order is the shared one
        return (self.labeled_loss_calculation(labeled_examples, labels),
                self.unlabeled_loss_calculation(labeled_examples, unlabeled_examples),
                self.fake_loss_calculation(unlabeled_examples, fake_examples))

    def interpolate_loss_calculation(self, interpolates):
        return self._scores(interpolates)

    def generator_loss_calculation(self, fake_examples, _):
        with nn.frozen_parameters(self.D):
            scores = self._scores(fake_examples)
        return bce_with_logits(scores, 0.0, self.batch_mean_of_examples)
