"""Crowd application glue (surface of reference crowd/srgan.py): model setup, the crowd labeled loss on HIP kernels
and the sliding-window inference of a full image.  Dataset loading and the evaluation plots of the
reference are out of scope for the hot path; ``dataset_setup`` provides synthetic
ShanghaiTech-shaped batches."""
import numpy as np
import torch

from .. import functional as F
from ..srgan import Experiment, as_var
from ..tape import no_grad
from .data import ImageSlidingWindowDataset
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

    def predict_full_example(self, full_example, network):
        """Count and density prediction of one full crowd image: every sliding-window patch goes through ``network``
        (batches of ``settings.batch_size``), each patch's density and uniformly spread count are accumulated at its
        position (clipped at the image borders) and divided by the number of patches covering each pixel
        (reference crowd/srgan.py:332-395).  Returns ``(count, density[H, W])`` like the reference.

        The reference resizes every predicted density patch to the patch size with ``scipy.misc.imresize`` (removed
        from SciPy); ``KnnDenseNetCat`` already predicts at the patch size, where that resize is the identity, and
        only that case is supported."""
        settings = self.settings
        patch_size = settings.image_patch_size
        half = patch_size // 2
        height, width = full_example.label.shape[0], full_example.label.shape[1]
        sum_density = np.zeros((height, width), dtype=np.float32)
        sum_count = np.zeros((height, width), dtype=np.float32)
        hits = np.zeros((height, width), dtype=np.int32)
        dataset = ImageSlidingWindowDataset(full_example, patch_size, settings.test_sliding_window_size)
        self.join_dnn_stream()
        for start in range(0, len(dataset), settings.batch_size):
            items = [dataset[i] for i in range(start, min(start + settings.batch_size, len(dataset)))]
            images = torch.stack([item[0] for item in items])
            with no_grad():
                densities, counts, _ = network(as_var(images))
            densities, counts = densities.cpu().numpy(), counts.cpu().numpy().reshape(-1)
            for (_, x, y), density, count in zip(items, densities, counts):
                if density.shape != (patch_size, patch_size):
                    raise NotImplementedError('density predictions at another resolution than the patch need '
                                              'scipy.misc.imresize, which SciPy removed')
                count_array = np.full(density.shape, count / density.size, dtype=np.float32)
                y_start = half - y if y - half < 0 else 0
                y_end = y + half - height if y + half > height else 0
                x_start = half - x if x - half < 0 else 0
                x_end = x + half - width if x + half > width else 0
                target = (slice(y - half + y_start, y + half - y_end), slice(x - half + x_start, x + half - x_end))
                source = (slice(y_start, density.shape[0] - y_end), slice(x_start, density.shape[1] - x_end))
                sum_density[target] += density[source]
                sum_count[target] += count_array[source]
                hits[target] += 1
        hits[hits == 0] = 1
        full_density = sum_density / hits.astype(np.float32)
        full_count = np.sum(sum_count / hits.astype(np.float32))
        return full_count, full_density
