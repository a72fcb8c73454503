"""Oracle restatement of the SRGAN / SGAN training step in torch-CPU.  Test infrastructure.

Follows reference ``srgan.py:259-391`` (and ``sgan.py:10-67`` for the classification variant) operation
for operation, including the quirks listed in SURVEY.md Appendix A.  Random draws (z for the D step,
z for the G step, the interpolation alpha) can be injected so results are reproducible against goldens.

An optional data-parallel context ``dp`` (duck-typed: ``world_size``, ``global_batch(local)``,
``all_reduce_sum_autograd(t)``, ``all_reduce_sum_(t)``) turns the step into the sharded algorithm of
SURVEY.md §8e; with ``dp=None`` it is the single-process reference algorithm.
"""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn
from torch.optim import Adam

from . import functional as OF


@dataclass
class Draws:
    z_d: Optional[torch.Tensor] = None
    z_g: Optional[torch.Tensor] = None
    alpha: Optional[torch.Tensor] = None


def freeze_batch_norm(module):
    """Every BatchNorm in eval mode: running statistics used, never updated (reference srgan.py:538-542)."""
    if isinstance(module, nn.modules.batchnorm._BatchNorm):
        module.eval()


class OracleExperiment:
    """SRGAN step.  ``settings`` is any object with the reference's attribute names (settings.py:14-67)."""

    def __init__(self, settings, D, DNN, G, labeled_loss_function=None, dp=None):
        self.settings, self.D, self.DNN, self.G, self.dp = settings, D, DNN, G, dp
        self.labeled_loss_function = labeled_loss_function or OF.labeled_loss
        s = settings  # reference srgan.py:131-138
        self.d_optimizer = Adam(D.parameters(), lr=s.learning_rate, weight_decay=s.weight_decay)
        self.g_optimizer = Adam(G.parameters(), lr=s.learning_rate)
        self.dnn_optimizer = Adam(DNN.parameters(), lr=s.learning_rate, weight_decay=s.weight_decay)
        for m in (D, DNN, G):
            m.train()
        self.labeled_features = self.unlabeled_features = self.fake_features = None
        self.interpolates_features = self.gradient_norm = None
        self.scalars = {}

    # --- data-parallel helpers -----------------------------------------------------------------
    def _batch_mean(self, per_example):
        """Mean over the global batch of a per-example vector."""
        if self.dp is None or self.dp.world_size == 1:
            return per_example.mean()
        return self.dp.all_reduce_sum_autograd(per_example.sum()) / self.dp.global_batch(per_example.shape[0])

    def _sync_gradients(self, module):
        if self.dp is not None and self.dp.world_size > 1:
            for p in module.parameters():
                if p.grad is not None:
                    self.dp.all_reduce_sum_(p.grad)

    def _distance(self, base, other, distance=None):
        s = self.settings
        return OF.feature_distance_loss(base, other, distance or s.matching_distance_function,
                                        normalize=s.normalize_feature_norm, dp=self.dp)

    # --- loss pieces ---------------------------------------------------------------------------
    def _labeled(self, network, x, y):
        s = self.settings
        predicted = network(x)
        if self.dp is not None and self.dp.world_size > 1:
            loss = self._labeled_sharded(predicted, y)
        else:
            loss = self.labeled_loss_function(predicted, y, order=s.labeled_loss_order)
        return loss * s.labeled_loss_multiplier

    def _labeled_sharded(self, predicted, y):
        # Only the generic loss is a plain batch mean of per-example terms; shard it explicitly.
        per_example = (predicted - y).abs().pow(self.settings.labeled_loss_order)
        return self._batch_mean(per_example)

    def dnn_training_step(self, x, y):
        """reference srgan.py:259-271 + :322-327."""
        self.DNN.apply(freeze_batch_norm)
        self.dnn_optimizer.zero_grad()
        loss = self._labeled(self.DNN, x, y)
        loss.backward()
        self._sync_gradients(self.DNN)
        self.dnn_optimizer.step()
        self.scalars['dnn_loss'] = loss.item()
        return loss

    def interpolate_loss(self, interpolates):
        """reference srgan.py:377-381."""
        self.D(interpolates)
        self.interpolates_features = self.D.features
        return self.interpolates_features.norm(dim=1)

    def gradient_penalty(self, fake, u, alpha=None):
        """reference srgan.py:360-375 (alpha's leading extent is settings.batch_size, Appendix A.1)."""
        s = self.settings
        if alpha is None:
            shape = [1] * u.dim()
            shape[0] = s.batch_size if self.dp is None else u.size(0)
            alpha = torch.rand(shape)
        interpolates = alpha * u.detach().requires_grad_() + (1 - alpha) * fake.detach().requires_grad_()
        f = self.interpolate_loss(interpolates)
        gradients = torch.autograd.grad(outputs=f, inputs=interpolates, grad_outputs=torch.ones_like(f),
                                        create_graph=True)[0]
        self.gradient_norm = gradients.view(u.size(0), -1).norm(dim=1)
        excess = torch.max(self.gradient_norm - 1, torch.zeros_like(self.gradient_norm))
        return self._batch_mean(excess ** 2) * s.gradient_penalty_multiplier

    def d_losses(self, x, y, u, fake, alpha):
        """The four discriminator losses in the reference's order, each back-propagated on its own."""
        s = self.settings
        labeled = self._labeled(self.D, x, y)                       # :329-335
        self.labeled_features = self.D.features
        labeled.backward()
        self.D(x)                                                   # :337-346 (x recomputed)
        self.labeled_features = self.D.features
        self.D(u)
        self.unlabeled_features = self.D.features
        unlabeled = self._distance(self.unlabeled_features, self.labeled_features)
        unlabeled = unlabeled * s.matching_loss_multiplier * s.srgan_loss_multiplier
        unlabeled.backward()
        self.D(u)                                                   # :348-358 (u recomputed)
        self.unlabeled_features = self.D.features
        self.D(fake.detach())
        self.fake_features = self.D.features
        fake_loss = self._distance(self.unlabeled_features, self.fake_features, s.contrasting_distance_function)
        fake_loss = fake_loss * s.contrasting_loss_multiplier * s.srgan_loss_multiplier
        fake_loss.backward()
        penalty = self.gradient_penalty(fake, u, alpha)             # :294-295
        penalty.backward()
        return labeled, unlabeled, fake_loss, penalty

    def g_loss(self, fake, u):
        """reference srgan.py:383-391 (no srgan_loss_multiplier, Appendix A.4)."""
        self.D(fake)
        self.fake_features = self.D.features
        self.D(u)
        detached = self.D.features.detach()
        return self._distance(detached, self.fake_features) * self.settings.matching_loss_multiplier

    def gan_training_step(self, x, y, u, step=0, draws: Draws = None):
        """reference srgan.py:273-320."""
        s, draws = self.settings, draws or Draws()
        self.D.apply(freeze_batch_norm)
        self.d_optimizer.zero_grad()
        z = draws.z_d if draws.z_d is not None else OF.discriminator_noise(u.size(0), self.G.input_size,
                                                                            s.mean_offset)
        # NB: the reference draws z after the unlabeled backward; position in the RNG stream only.
        fake = self.G(z)
        labeled, unlabeled, fake_loss, penalty = self.d_losses(x, y, u, fake, draws.alpha)
        self.d_grads = {n: p.grad.detach().clone() for n, p in self.D.named_parameters() if p.grad is not None}
        self._sync_gradients(self.D)
        self.d_optimizer.step()
        result = {'labeled_loss': labeled.item(), 'unlabeled_loss': unlabeled.item(),
                  'fake_loss': fake_loss.item(), 'gradient_penalty': penalty.item(),
                  'gradient_norm_mean': self.gradient_norm.mean().item()}
        if step % s.generator_training_step_period == 0:
            self.g_optimizer.zero_grad()
            z = draws.z_g if draws.z_g is not None else torch.randn(u.size(0), self.G.input_size)
            generator_loss = self.g_loss(self.G(z), u)
            generator_loss.backward()
            self._sync_gradients(self.G)
            self.g_optimizer.step()
            result['generator_loss'] = generator_loss.item()
        self.scalars.update(result)
        return result


class OracleSganExperiment(OracleExperiment):
    """Classification-GAN losses (reference sgan.py:10-67)."""

    def __init__(self, settings, D, DNN, G, bins, dp=None):
        super().__init__(settings, D, DNN, G, dp=dp)
        self.bins = bins
        self.cross_entropy = nn.CrossEntropyLoss()
        self.bce = nn.BCEWithLogitsLoss()

    def _labeled(self, network, x, y):
        indexes = OF.real_numbers_to_bin_indexes(y, self.bins)
        return self.cross_entropy(network(x), indexes) * self.settings.labeled_loss_multiplier

    def _binary(self, examples, target):
        logits = OF.logsumexp(self.D(examples), dim=1)
        return self.bce(logits, torch.full_like(logits, target))

    def interpolate_loss(self, interpolates):
        # sgan.py:52-59: a SCALAR, already multiplied by the GP multiplier (applied again at srgan.py:374).
        return self._binary(interpolates, 0.0) * self.settings.gradient_penalty_multiplier

    def d_losses(self, x, y, u, fake, alpha):
        s = self.settings
        labeled = self._labeled(self.D, x, y)
        labeled.backward()
        unlabeled = self._binary(u, 1.0) * s.matching_loss_multiplier
        unlabeled.backward()
        fake_loss = self._binary(fake.detach(), 0.0) * s.matching_loss_multiplier
        fake_loss.backward()
        penalty = self.gradient_penalty(fake, u, alpha)
        penalty.backward()
        return labeled, unlabeled, fake_loss, penalty

    def g_loss(self, fake, u):
        return self._binary(fake, 0.0).neg()
