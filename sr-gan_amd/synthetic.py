"""Synthetic, seeded stand-ins for the reference's datasets (the real ones need downloads; SURVEY.md §8a-D1
fixes only the batch contract).  Batches are generated on the host with torch's CPU generator and moved to the
device, like the reference's DataLoader + ``.to(gpu)`` (srgan.py:107-117); under data parallelism every rank
draws the same global batch and keeps its own shard."""
import math

import torch

from .utility import current_device


class SyntheticLoader:
    """An endless iterable of pre-generated device batches (``pool`` distinct batches, cycled)."""
    # The batches exist -- complete, in HBM -- before the first step and are never freed: a consumer on another stream
    # needs no ordering against the stream that "produced" them (see Experiment.dnn_training_step).
    resident = True

    def __init__(self, make_batch, pool=4):
        self.batches = [make_batch(i) for i in range(pool)]

    def __iter__(self):
        return iter(self.batches)

    @staticmethod
    def _shard(tensor, dp):
        if dp is None or dp.world_size == 1:
            return tensor
        return dp.shard(tensor)

    @classmethod
    def crowd(cls, batch_size, size, seed=0, dp=None, pool=2):
        generator = torch.Generator().manual_seed(seed)
        device = current_device()

        def make(_):
            image = torch.rand(batch_size, 3, size, size, generator=generator) * 2 - 1
            heads = (torch.rand(batch_size, size, size, generator=generator) < 0.002).float()
            knn_map = torch.rand(batch_size, size, size, generator=generator)
            return tuple(cls._shard(t, dp).to(device) for t in (image, heads, knn_map))
        return cls(make, pool)

    @classmethod
    def crowd_density(cls, batch_size, size, seed=0, dp=None, pool=2):
        """(image, quarter-resolution density label f32[S/4, S/4]) batches: the crowd SGAN's batch contract (reference
        crowd/sgan.py:28-38: ``labels`` is the density map whose sum is the head count)."""
        generator = torch.Generator().manual_seed(seed)
        device = current_device()

        def make(_):
            image = torch.rand(batch_size, 3, size, size, generator=generator) * 2 - 1
            density = (torch.rand(batch_size, size // 4, size // 4, generator=generator) < 0.1).float() * \
                torch.rand(batch_size, 1, 1, generator=generator) * 8
            return tuple(cls._shard(t, dp).to(device) for t in (image, density))
        return cls(make, pool)

    @classmethod
    def images(cls, batch_size, size, label_range=(10.0, 95.0), seed=0, dp=None, pool=2):
        """(image f32[3,H,W] in [-1,1], scalar label) batches: age (10..95 years) / driving (angle) contract
        (age/data.py:52-60)."""
        height, width = (size, size) if isinstance(size, int) else size
        generator = torch.Generator().manual_seed(seed)
        device = current_device()
        low, high = label_range

        def make(_):
            image = torch.rand(batch_size, 3, height, width, generator=generator) * 2 - 1
            label = torch.rand(batch_size, generator=generator) * (high - low) + low
            return tuple(cls._shard(t, dp).to(device) for t in (image, label))
        return cls(make, pool)
