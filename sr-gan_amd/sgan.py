"""Classification-GAN variant of the step (surface of reference sgan.py:10-67) on the HIP tape.

Cross-entropy over binned labels, logsumexp -> BCE-with-logits real/fake losses, and the gradient penalty on
the (scalar) BCE of the interpolates -- with the reference's double application of the penalty multiplier
(sgan.py:58 and srgan.py:374, Appendix A.7)."""
from abc import ABC

from . import functional as F
from .srgan import Experiment, as_var
from .tape import backward, no_grad
from .utility import logsumexp


def cross_entropy_with_bins(logits, labels, bins, batch_mean):
    """CrossEntropyLoss(logits, nearest_bin(labels)): mean_b(logsumexp(logits_b) - logits_b[index_b])
    (reference sgan.py:18-32, utility.py:141-144)."""
    onehot = F.nearest_bin_onehot(labels, bins)
    picked = F.row_dot(logits, onehot)
    return batch_mean(F.sub(logsumexp(logits, dim=1), picked))


def bce_with_logits(logits, target, batch_mean):
    """BCEWithLogitsLoss against a constant target: mean(softplus(x) - target * x)."""
    per_example = F.softplus(logits)
    if target != 0.0:
        per_example = F.sub(per_example, F.scale(logits, target))
    return batch_mean(per_example)


class SganExperiment(Experiment, ABC):
    """An SGAN experiment (reference sgan.py:10-67).  Subclasses set ``self.bins`` (a device Var / tensor)."""

    def __init__(self, settings):
        super().__init__(settings)
        self.bins = None

    def _bins(self):
        self.bins = as_var(self.bins)
        return self.bins

    def class_logits(self, network, examples):
        """The (B, bins) class logits of ``network`` (applications whose networks return more than the logits override
        this, e.g. the crowd SGAN's (density, logits) pair)."""
        return network(examples)

    def dnn_loss_calculation(self, labeled_examples, labels):
        loss = cross_entropy_with_bins(self.class_logits(self.DNN, labeled_examples), labels, self._bins(),
                                       self.batch_mean_of_examples)
        return F.scale(loss, self.settings.labeled_loss_multiplier)

    def labeled_loss_calculation(self, labeled_examples, labels):
        loss = cross_entropy_with_bins(self.class_logits(self.D, labeled_examples), labels, self._bins(),
                                       self.batch_mean_of_examples)
        return F.scale(loss, self.settings.labeled_loss_multiplier)

    def _binary_loss(self, examples, target):
        return bce_with_logits(logsumexp(self.class_logits(self.D, examples), dim=1), target, self.batch_mean_of_examples)

    def unlabeled_loss_calculation(self, labeled_examples, unlabeled_examples):
        return F.scale(self._binary_loss(unlabeled_examples, 1.0), self.settings.matching_loss_multiplier)

    def fake_loss_calculation(self, unlabeled_examples, fake_examples):
        return F.scale(self._binary_loss(fake_examples.detach(), 0.0), self.settings.matching_loss_multiplier)

    def discriminator_losses_shared_forwards(self, labeled_examples, labels, unlabeled_examples, fake_examples):
        # The SGAN losses never recompute a forward, so the reference order already is the shared one.
        return (self.labeled_loss_calculation(labeled_examples, labels),
                self.unlabeled_loss_calculation(labeled_examples, unlabeled_examples),
                self.fake_loss_calculation(unlabeled_examples, fake_examples))

    def interpolate_loss_calculation(self, interpolates):
        loss = self._binary_loss(interpolates, 0.0)        # differentiated twice
        return F.scale(loss, self.settings.gradient_penalty_multiplier)

    def generator_loss_calculation(self, fake_examples, unlabeled_examples):
        from . import nn
        with nn.frozen_parameters(self.D):
            return F.neg(self._binary_loss(fake_examples, 0.0))
