"""Crowd application glue (surface of reference crowd/srgan.py): model setup and the crowd labeled loss on HIP
kernels.  Dataset loading / evaluation / sliding-window inference of the reference are out of scope for the
hot path; ``dataset_setup`` provides synthetic ShanghaiTech-shaped batches."""
import torch

from .. import functional as F
from ..srgan import Experiment
from .models import DCGenerator, KnnDenseNetCat
from ..synthetic import SyntheticLoader


class CrowdExperiment(Experiment):
    """The crowd application."""

    def dataset_setup(self):
        """Synthetic (image, head-count label, ikNN map) batches of the reference's batch contract
        (crowd/shanghai_tech_data.py:99-104): image f32[3,S,S] in [-1,1], label f32[S,S], map f32[S,S]."""
        settings = self.settings
        size = settings.image_patch_size
        self.train_dataset_loader = SyntheticLoader.crowd(settings.batch_size, size, seed=settings.labeled_dataset_seed,
                                                          dp=self.dp)
        self.unlabeled_dataset_loader = SyntheticLoader.crowd(settings.batch_size, size, seed=100, dp=self.dp)

    def model_setup(self):
        """reference crowd/srgan.py:92-96 (``pretrained=True`` there downloads torchvision weights; offline the
        networks are freshly initialised and a checkpoint can be loaded instead)."""
        size = self.settings.image_patch_size
        self.G = DCGenerator(image_size=size)
        self.D = KnnDenseNetCat(image_size=size)
        self.DNN = KnnDenseNetCat(image_size=size)

    def validation_summaries(self, step):
        pass

    def labeled_loss_function(self, predicted_labels, labels, order=2):
        """count loss + map_multiplier * map loss (reference crowd/srgan.py:247-254)."""
        head_labels, map_labels = labels
        _, predicted_count_labels, predicted_maps = predicted_labels
        map_rows = F.crowd_map_l1(predicted_maps, map_labels)
        map_loss = self.batch_mean_of_examples(F.pow_scalar(map_rows, order))
        head_counts = F.row_sum(head_labels)
        count_loss = self.batch_mean_of_examples(F.pow_scalar(F.abs_(F.sub(predicted_count_labels, head_counts)), order))
        return F.add(count_loss, F.scale(map_loss, self.settings.map_multiplier))
