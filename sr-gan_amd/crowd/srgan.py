"""Crowd application glue (surface of reference crowd/srgan.py): model setup, the crowd labeled loss on HIP kernels
and the sliding-window inference of a full image.  Dataset loading and the evaluation plots of the
reference are out of scope for the hot path; ``dataset_setup`` provides synthetic
ShanghaiTech-shaped batches."""
import numpy as np
import torch

from .. import functional as F
from ..srgan import Experiment, as_var
from ..tape import no_grad
from .data import CrowdExample, ImageSlidingWindowDataset
from .models import DCGenerator, KnnDenseNetCat
from ..synthetic import SyntheticLoader


class CrowdExperiment(Experiment):
    """The crowd application."""

    DATABASE_ENV = 'SRGAN_CROWD_DATABASE'       # e.g. .../ShanghaiTech ; SRGAN_CROWD_DATABASE_PART e.g. part_A

    def dataset_setup(self):
        """Without a database on disk: synthetic (image, head-count label, ikNN map) batches of the reference's batch
        contract (crowd/shanghai_tech_data.py:99-104): image f32[3,S,S] in [-1,1], label f32[S,S], map f32[S,S].
        With ``SRGAN_CROWD_DATABASE`` pointing at a database preprocessed by the reference: the
        ShanghaiTech branch of reference crowd/srgan.py:54-69 -- labeled / unlabeled scenes resident in HBM and cut into
        random patches on the device, the test split behind ``dataset_class`` for the full-image summaries."""
        import os
        settings = self.settings
        size = settings.image_patch_size
        directory = os.environ.get(self.DATABASE_ENV)
        if directory:
            from .data import DeviceCrowdPatchLoader, PreprocessedCrowdDataset
            part = os.environ.get(self.DATABASE_ENV + '_PART') or None
            maps = settings.map_directory_name

            def scenes(count):
                return PreprocessedCrowdDataset(directory, 'train', part, number_of_examples=count,
                                                map_directory_name=maps).examples()
            self.train_dataset_loader = DeviceCrowdPatchLoader(scenes(settings.labeled_dataset_size), settings.batch_size,
                                                               size, seed=settings.labeled_dataset_seed, dp=self.dp)
            self.unlabeled_dataset_loader = DeviceCrowdPatchLoader(scenes(settings.unlabeled_dataset_size),
                                                                   settings.batch_size, size, seed=100, dp=self.dp)
            self.dataset_class = lambda dataset, map_directory_name: PreprocessedCrowdDataset(
                directory, dataset, part, map_directory_name=map_directory_name)
            return
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
        """The scalar summaries of reference crowd/srgan.py:98-147 when evaluation datasets are attached
        (``train_dataset`` / ``validation_dataset``: indexable, items ``(image, label, map)``; ``dataset_class`` for the
        full-image test summaries); the image grids of the reference are out of scope."""
        train_dataset = getattr(self, 'train_dataset', None)
        validation_dataset = getattr(self, 'validation_dataset', None)
        if train_dataset is not None and validation_dataset is not None:
            settings = self.settings
            self.evaluation_epoch(settings, self.DNN, train_dataset, self.dnn_summary_writer, '2 Train Error', shuffle=False)
            dnn_validation_count_mae = self.evaluation_epoch(settings, self.DNN, validation_dataset,
                                                             self.dnn_summary_writer, '1 Validation Error', shuffle=False)
            self.evaluation_epoch(settings, self.D, train_dataset, self.gan_summary_writer, '2 Train Error', shuffle=False)
            self.evaluation_epoch(settings, self.D, validation_dataset, self.gan_summary_writer, '1 Validation Error',
                                  comparison_value=dnn_validation_count_mae, shuffle=False)
        if getattr(self, 'dataset_class', None) is not None:
            self.test_summaries()

    def images_to_predicted_labels(self, network, images):
        """reference crowd/srgan.py:256-259"""
        predicted_densities, predicted_counts, predicted_maps = network(images)
        return predicted_densities, predicted_counts, predicted_maps

    def evaluation_epoch(self, settings, network, dataset, summary_writer, summary_name, comparison_value=None,
                         shuffle=True):
        """Count ME / MAE / MSE and kNN-map MAE / MSE of ``network`` over batches of ``dataset`` (reference
        crowd/srgan.py:149-191), stopping once more than 100 examples have been seen.  As in the reference, the map
        errors cover the FIRST batch only (its concatenation of the maps sits inside the "still empty" branch)."""
        self.join_dnn_stream()
        order = np.random.permutation(len(dataset)) if shuffle else np.arange(len(dataset))
        predicted_counts, label_counts = [], []
        maps = predicted_maps = None
        for index, start in enumerate(range(0, len(order), settings.batch_size)):
            items = [dataset[int(i)] for i in order[start:start + settings.batch_size]]
            images = torch.stack([torch.as_tensor(item[0]) for item in items])
            labels = torch.stack([torch.as_tensor(item[1]) for item in items])
            with no_grad():
                _, batch_predicted_counts, batch_predicted_maps = self.images_to_predicted_labels(network, as_var(images))
            predicted_counts.append(batch_predicted_counts.cpu().numpy().reshape(-1).astype(np.float64))
            label_counts.append(labels.cpu().numpy().astype(np.float64).sum(axis=(1, 2)))
            if maps is None:
                maps = np.stack([np.asarray(torch.as_tensor(item[2]).cpu(), dtype=np.float64) for item in items])
                predicted_maps = batch_predicted_maps.cpu().numpy().astype(np.float64)
            if index * settings.batch_size >= 100:
                break
        predicted_counts, label_counts = np.concatenate(predicted_counts), np.concatenate(label_counts)
        maps = np.expand_dims(maps, axis=1)
        count_mae = np.abs(predicted_counts - label_counts).mean()
        summary_writer.add_scalar('{}/ME'.format(summary_name), (predicted_counts - label_counts).mean())
        summary_writer.add_scalar('{}/MAE'.format(summary_name), count_mae)
        summary_writer.add_scalar('{}/kNN MAE'.format(summary_name), np.abs(predicted_maps - maps).mean())
        summary_writer.add_scalar('{}/MSE'.format(summary_name), (np.abs(predicted_counts - label_counts) ** 2).mean())
        summary_writer.add_scalar('{}/kNN MSE'.format(summary_name), (np.abs(predicted_maps - maps) ** 2).mean())
        if comparison_value is not None:
            summary_writer.add_scalar('{}/Ratio MAE GAN DNN'.format(summary_name), count_mae / comparison_value)
        return count_mae

    def test_summaries(self):
        """Full-image test errors of both networks through ``predict_full_example`` (reference crowd/srgan.py:261-300):
        NAE / MAE / RMSE of the count, MAE / RMSE of the density sum, and the GAN-to-DNN ratios."""
        import random
        test_dataset = self.dataset_class(dataset='test', map_directory_name=self.settings.map_directory_name)
        if self.settings.test_summary_size is not None:
            indexes = random.sample(range(test_dataset.length), self.settings.test_summary_size)
        else:
            indexes = range(test_dataset.length)
        dnn_mae_count = dnn_rmse_count = None
        for network in (self.DNN, self.D):
            totals = dict.fromkeys(('Count error', 'NAE', 'Density sum error', 'SE count', 'SE density'), 0.0)
            for index in indexes:
                full_image, full_label, _ = test_dataset[index]
                full_example = CrowdExample(image=full_image, label=full_label)
                predicted_count, predicted_label = self.predict_full_example(full_example, network)
                true_count = full_example.label.sum()
                totals['Count error'] += np.abs(predicted_count - true_count)
                totals['NAE'] += np.abs(predicted_count - true_count) / true_count
                totals['Density sum error'] += np.abs(predicted_label.sum() - true_count)
                totals['SE count'] += (predicted_count - true_count) ** 2
                totals['SE density'] += (predicted_label.sum() - true_count) ** 2
            summary_writer = self.dnn_summary_writer if network is self.DNN else self.gan_summary_writer
            mae_count = totals['Count error'] / len(indexes)
            rmse_count = (totals['SE count'] / len(indexes)) ** 0.5
            summary_writer.add_scalar('0 Test Error/NAE count', totals['NAE'] / len(indexes))
            summary_writer.add_scalar('0 Test Error/MAE count', mae_count)
            summary_writer.add_scalar('0 Test Error/MAE density', totals['Density sum error'] / len(indexes))
            summary_writer.add_scalar('0 Test Error/RMSE count', rmse_count)
            summary_writer.add_scalar('0 Test Error/RMSE density', (totals['SE density'] / len(indexes)) ** 0.5)
            if network is self.DNN:
                dnn_mae_count, dnn_rmse_count = mae_count, rmse_count
            else:
                summary_writer.add_scalar('0 Test Error/Ratio MAE GAN DNN', mae_count / dnn_mae_count)
                summary_writer.add_scalar('0 Test Error/Ratio RMSE GAN DNN', rmse_count / dnn_rmse_count)

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
