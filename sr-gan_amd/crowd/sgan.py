"""SGAN for the crowd application (surface of reference crowd/sgan.py:10-87): ``JointDCDiscriminator`` pairs classify
the head count into ``settings.number_of_bins`` bins over [0, 300] and regress a quarter-resolution density map; the
real / fake losses are the SGAN ones on the log-sum-exp of the class logits.

Upstream this experiment is stale (its ``model_setup`` pairs a 224-pixel generator with a 128-pixel discriminator, and
``CrowdExperiment`` no longer feeds quarter-resolution density labels); here all three networks are built at
``settings.image_patch_size`` and the labels of a batch are ONE tensor: the density label (B, S/4, S/4).  The loss
methods are pinned against the reference's own (golden g14)."""
import torch

from .. import functional as F
from ..sgan import SganExperiment, cross_entropy_with_bins
from ..utility import logits_to_bin_values
from .models import DCGenerator, JointDCDiscriminator
from .srgan import CrowdExperiment

COUNT_RANGE = (0, 300)
DENSITY_LOSS_WEIGHT = 10          # reference crowd/sgan.py:37,50


class CrowdSganExperiment(SganExperiment, CrowdExperiment):
    def __init__(self, settings):
        super().__init__(settings)
        self.bins = torch.linspace(*COUNT_RANGE, settings.number_of_bins)

    def model_setup(self):
        size = self.settings.image_patch_size
        bins = self.settings.number_of_bins
        self.G = DCGenerator(image_size=size)
        self.D = JointDCDiscriminator(image_size=size, number_of_outputs=bins)
        self.DNN = JointDCDiscriminator(image_size=size, number_of_outputs=bins)

    def dataset_setup(self):
        """Synthetic (image, density label) batches of the batch contract above (the reference's loaders for this
        experiment no longer exist upstream)."""
        from ..synthetic import SyntheticLoader
        settings = self.settings
        self.train_dataset_loader = SyntheticLoader.crowd_density(settings.batch_size, settings.image_patch_size,
                                                                  seed=settings.labeled_dataset_seed, dp=self.dp)
        self.unlabeled_dataset_loader = SyntheticLoader.crowd_density(settings.batch_size, settings.image_patch_size,
                                                                      seed=100, dp=self.dp)

    def validation_summaries(self, step):
        pass

    def class_logits(self, network, examples):
        return network(examples)[1]

    def images_to_predicted_labels(self, network, images):
        """(density maps, count = centre of the arg-max bin) (reference crowd/sgan.py:22-26)."""
        predicted_densities, predicted_count_logits = network(images)
        return predicted_densities, logits_to_bin_values(predicted_count_logits, self._bins())

    def _labeled_loss(self, network, labeled_examples, density_labels):
        """cross-entropy of the binned count + 10 * mean_b sum_hw |density - label|^2 (reference crowd/sgan.py:28-52)."""
        predicted_density_labels, predicted_count_logits = network(labeled_examples)
        difference = F.sub(predicted_density_labels, F.view(density_labels, predicted_density_labels.shape))
        density_loss = self.batch_mean_of_examples(F.row_sum(F.square(difference)))
        count_labels = F.row_sum(density_labels)
        count_loss = cross_entropy_with_bins(predicted_count_logits, count_labels, self._bins(), self.batch_mean_of_examples)
        labeled_loss = F.add(count_loss, F.scale(density_loss, DENSITY_LOSS_WEIGHT))
        return F.scale(labeled_loss, self.settings.labeled_loss_multiplier)

    def dnn_loss_calculation(self, labeled_examples, labels):
        return self._labeled_loss(self.DNN, labeled_examples, labels)

    def labeled_loss_calculation(self, labeled_examples, labels):
        return self._labeled_loss(self.D, labeled_examples, labels)
