"""Semi-supervised regression GAN training on MI355X: the ``Experiment`` surface of reference srgan.py with
the training step (dnn_training_step / gan_training_step and the loss calculations, srgan.py:259-391)
running on the HIP tape.

Differences from the reference that are deliberate (each keeps results identical to 1e-3):

* ``settings.reference_schedule = False`` (default) shares the discriminator forward of the labeled batch
  between the labeled and unlabeled losses and that of the unlabeled batch between the unlabeled and fake
  losses, and back-propagates their sum once: the reference recomputes those forwards (Appendix A.2).
  ``True`` replays the reference's exact forward / backward order.
* the discriminator's weight gradients are not computed during the generator update (the reference computes
  and discards them, srgan.py:304 then :278).
* one process per GPU: an optional data-parallel context shards the three batches and exchanges feature
  sums and gradients over RCCL (``parallel.py``).  The gradient all-reduces are asynchronous: each starts during the
  last backward pass into its network's arena (reference srgan.py:264 for the DNN, :295 for D, :304 for G are where those
  gradients complete) and the optimizer update that needs it is applied when the weights are next used
  (``finish_update``): the DNN's exchange runs under the whole GAN step, D's under the generator forward, G's under
  the next iteration's DNN step.  The arithmetic is unchanged.
"""
import datetime
import os
import re
import select
import sys
from abc import ABC, abstractmethod

import numpy as np
import torch
from scipy.stats import norm

from . import _lib
from . import functional as F
from . import nn
from .optim import Adam
from .settings import Settings
from .tape import Var, backward, no_grad
from .utility import SummaryWriter, MixtureModel, current_device, make_directory_name_unique, seed_all


def as_var(value, staged=False):
    """Device tensors from the data pipeline enter the tape as constants.  ``staged``: a small host tensor goes through a
    pinned staging buffer (the three random draws of an iteration ask for it; nothing else does)."""
    if isinstance(value, Var) or value is None:
        return value
    if isinstance(value, (tuple, list)):
        return tuple(as_var(v) for v in value)
    if isinstance(value, np.ndarray):
        value = torch.from_numpy(value)
    if staged and value.device.type == 'cpu' and torch.cuda.is_available() and value.dtype == torch.float32 and \
            0 < value.numel() <= _STAGING_LIMIT and not torch.cuda.is_current_stream_capturing():
        return F.constant(_through_pinned_staging(value))
    return F.constant(value.to(current_device(), non_blocking=True))


# The three random draws of an iteration (srgan.py:286-289,301,364) reach the device through PINNED staging buffers: a copy
# from pageable memory is staged by the runtime and its blit kernel sat 0.55 ms on the stream per draw (rocprofv3, round 3:
# three __amd_rocclr_copyBuffer launches of 555 us per iteration, the first in front of the generator's forward pass).  A ring
# of buffers per size; a buffer is reused only after the copy that read it has finished.  Only the draw call sites stage
# (`as_var(..., staged=True)`), never while a stream is capturing (a captured copy would re-read whatever the reusable slot
# holds at replay time), and the rings of at most _STAGING_SIZES distinct sizes are kept (least recently used dropped), so
# pinned memory stays bounded when batch sizes vary.
_STAGING_LIMIT = 1 << 20
_STAGING_RING = 8
_STAGING_SIZES = 6
_staging = {}


def _through_pinned_staging(value):
    device = current_device()
    key = (value.numel(), str(device))
    ring = _staging.pop(key, None)
    if ring is None:
        while len(_staging) >= _STAGING_SIZES:
            oldest = _staging.pop(next(iter(_staging)))
            for _, event in oldest['slots']:
                if event is not None:
                    event.synchronize()             # its copies are done before the pinned pages go away
        ring = {'next': 0, 'slots': [[torch.empty(value.numel(), dtype=torch.float32).pin_memory(), None]
                                     for _ in range(_STAGING_RING)]}
    _staging[key] = ring                            # (re-inserted last: the dict's order is the recency order)
    slot = ring['slots'][ring['next']]
    ring['next'] = (ring['next'] + 1) % _STAGING_RING
    if slot[1] is not None:
        slot[1].synchronize()                     # (eight copies ago: long done)
    slot[0].copy_(value.reshape(-1))
    out = torch.empty(value.shape, dtype=torch.float32, device=device)
    out.view(-1).copy_(slot[0], non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record()
    return out


def examples_on_gpu():
    return torch.cuda.is_available()


class Experiment(ABC):
    """Manages one experimental trial (reference srgan.py:24-469)."""

    def __init__(self, settings: Settings):
        self.settings = settings
        self.trial_directory = None
        self.dnn_summary_writer = None
        self.gan_summary_writer = None
        self.dataset_class = None
        self.train_dataset = None
        self.train_dataset_loader = None
        self.unlabeled_dataset = None
        self.unlabeled_dataset_loader = None
        self.validation_dataset = None
        self.DNN = None
        self.dnn_optimizer = None
        self.D = None
        self.d_optimizer = None
        self.G = None
        self.g_optimizer = None
        self.signal_quit = False
        self.starting_step = 0

        self.labeled_features = None
        self.unlabeled_features = None
        self.fake_features = None
        self.interpolates_features = None
        self.gradient_norm = None

        self.dp = None                 # optional parallel.DataParallel context
        self._pending_updates = {}     # network name -> (gradient exchange in flight, optimizer to step after it)
        self.injected_draws = None     # tests: dict with 'z_d', 'z_g', 'alpha' device/CPU tensors, used once
        self.last_losses = {}          # device scalars of the latest step (no host sync)

    @property
    def parallel(self):
        """True when the data-parallel exchanges run: more than one rank, or one rank with ``dp.force`` (the whole
        collective path -- feature sums, bucketed gradient exchange, broadcasts -- on a world of one: how the RCCL
        backend is exercised on a one-GPU box, ``bench.py --force-dp``)."""
        return self.dp is not None and self.dp.active

    # ------------------------------------------------------------------------------------------ lifecycle
    def _trial_decision(self):
        """What reference srgan.py:54-70 decides from the file system before a trial starts: skip it, its (unique)
        directory, and where to load a model from.  Taken on rank 0 only and broadcast, so that every rank of a
        data-parallel run trains the same trial from the same checkpoint."""
        settings = self.settings
        trial_directory = os.path.join(settings.logs_directory, settings.trial_name)
        decision = dict(skip=False, trial_directory=trial_directory, load_model_path=settings.load_model_path,
                        continue_existing_experiments=settings.continue_existing_experiments)
        if (settings.skip_completed_experiment and os.path.exists(trial_directory) and
                '/check' not in trial_directory and not settings.continue_existing_experiments):
            decision['skip'] = True
            return decision
        if not settings.continue_existing_experiments:
            decision['trial_directory'] = make_directory_name_unique(trial_directory)
        else:
            if os.path.exists(trial_directory) and settings.load_model_path is not None:
                raise ValueError('Cannot load from path and continue existing at the same time.')
            elif settings.load_model_path is None:
                decision['load_model_path'] = trial_directory
            elif not os.path.exists(trial_directory):
                decision['continue_existing_experiments'] = False
        os.makedirs(os.path.join(decision['trial_directory'], settings.temporary_directory), exist_ok=True)
        return decision

    def train(self):
        """Run the SRGAN training for the experiment (reference srgan.py:52-86)."""
        settings = self.settings
        parallel = self.parallel
        if parallel and settings.batch_size % self.dp.world_size:
            raise ValueError(f'batch_size {settings.batch_size} is the GLOBAL batch and must be divisible by the '
                             f'{self.dp.world_size} data-parallel ranks')
        decision = self._trial_decision() if not parallel or self.dp.rank == 0 else None
        if parallel:
            decision = self.dp.broadcast_object(decision)
        self.trial_directory = decision['trial_directory']
        if decision['skip']:
            print('`{}` experiment already exists. Skipping...'.format(self.trial_directory))
            return
        settings.load_model_path = decision['load_model_path']
        settings.continue_existing_experiments = decision['continue_existing_experiments']
        print(self.trial_directory)
        self.prepare_summary_writers()
        seed_all(0)

        self.dataset_setup()
        self.model_setup()
        self.gpu_mode()              # parameters move into flat device arenas before the optimizers bind
        self.prepare_optimizers()
        self.load_models()
        if parallel:                 # every rank starts from rank 0's weights (and the checkpoint it may have loaded)
            for module in (self.D, self.DNN, self.G):
                if module is not None and getattr(module, '_srgan_arena', None) is not None:
                    self.dp.broadcast_parameters(module._srgan_arena)
        self.train_mode()

        self.training_loop()

        print('Completed {}'.format(self.trial_directory))
        if settings.should_save_models:
            self.save_models(step=settings.steps_to_run)
        self.close()

    # (network attribute, optimizer attribute): also the keys of the checkpoint dictionary (reference srgan.py:88-97)
    CHECKPOINT_PARTS = (('DNN', 'dnn_optimizer'), ('D', 'd_optimizer'), ('G', 'g_optimizer'))

    def save_models(self, step):
        """One torch.save dict with the reference's keys (srgan.py:88-97); written by rank 0 only."""
        self.join_dnn_stream()         # also applies optimizer updates still waiting for their gradient exchange
        if self.dp is not None and self.dp.rank != 0:
            return
        model = {'step': step}
        for network, optimizer in self.CHECKPOINT_PARTS:
            model[network] = getattr(self, network).state_dict()
            model[optimizer] = getattr(self, optimizer).state_dict()
        torch.save(model, os.path.join(self.trial_directory, f'model_{step}.pth'))

    @staticmethod
    def unpack_labeled(samples):
        """(examples, labels) of a labeled batch; a third element is a secondary label (the crowd maps)."""
        if len(samples) == 2:
            return as_var(samples[0]), as_var(samples[1])
        examples, primary_labels, secondary_labels = samples
        return as_var(examples), as_var((primary_labels, secondary_labels))

    def end_of_step(self, step, summary_writer, step_time_start):
        """Periodic validation summaries, the stdin commands and periodic checkpoints (reference srgan.py:120-129);
        returns the time stamp the next progress line is measured from."""
        if summary_writer.is_summary_step() or step == self.settings.steps_to_run - 1:
            print('\rStep {}, {}...'.format(step, datetime.datetime.now() - step_time_start), end='')
            step_time_start = datetime.datetime.now()
            self.join_dnn_stream()
            self.eval_mode()
            with no_grad():
                self.validation_summaries(step)
            self.train_mode()
        self.handle_user_input(step)
        if self.settings.save_step_period and step % self.settings.save_step_period == 0 and step != 0:
            self.save_models(step=step)
        return step_time_start

    def training_loop(self):
        """The per-step hot loop (reference srgan.py:99-129)."""
        train_dataset_generator = self.infinite_iter(self.train_dataset_loader)
        unlabeled_dataset_generator = self.infinite_iter(self.unlabeled_dataset_loader)
        step_time_start = datetime.datetime.now()
        for step in range(self.starting_step, self.settings.steps_to_run):
            self.adjust_learning_rate(step)
            labeled_examples, labels = self.unpack_labeled(next(train_dataset_generator))
            unlabeled_examples = as_var(next(unlabeled_dataset_generator)[0])
            self.training_iteration(labeled_examples, labels, unlabeled_examples, step)
            step_time_start = self.end_of_step(step, self.gan_summary_writer, step_time_start)

    def training_iteration(self, labeled_examples, labels, unlabeled_examples, step):
        """The body of the reference's loop (srgan.py:104-118): the DNN step, then the GAN step.  With
        ``settings.step_graph`` on a single device the iteration is captured once as a HIP graph and replayed
        (``graph.CapturedIteration``); summary steps and the first ``settings.step_graph_warmup`` iterations run eagerly."""
        if getattr(self.settings, 'step_graph', False) and examples_on_gpu() and not getattr(self.settings, 'storage_dtype', None) \
                and self._exchanges_are_capturable():
            if getattr(self, '_captured_iteration', None) is None:
                from .graph import CapturedIteration
                self._captured_iteration = CapturedIteration(self)
            return self._captured_iteration.run(labeled_examples, labels, unlabeled_examples, step)
        self.dnn_training_step(labeled_examples, labels, step)
        self.gan_training_step(labeled_examples, labels, unlabeled_examples, step)

    def _stream_setting(self, name):
        """A side-stream switch of ``settings`` as THIS iteration sees it: off while the iteration is captured / replayed as a
        HIP graph under data parallelism (``_single_compute_stream``), without touching ``settings`` itself."""
        if getattr(self, '_single_compute_stream', False):
            return False
        return getattr(self.settings, name, False)

    def _exchanges_are_capturable(self):
        """True when this run's iterations can be captured as HIP graphs.  Single device: always.  Data parallel: only when the
        caller OPTED IN with ``settings.step_graph_collectives = 'abi'`` -- the exchanges then run through the C ABI's RCCL entry
        points on a stream this process owns (``DataParallel.use_abi_collectives``, the default device transport; destroyed by
        ``close()``) and are captured with the iteration; collectives that go through a host-side process group
        (gloo, torch's own NCCL work queue) keep the run eager.  No setting is modified: the compute side streams are switched
        off for such a run through ``_stream_setting`` (hipStreamEndCapture crashes -- ROCm 7.0 runtime, a segmentation fault
        inside capture_end, round 5; re-tested in round 6 with the process on four streams: gpurun_out/r6u -- when a capture holds
        the compute side streams AND the communication stream)."""
        if not self.parallel:
            return True
        if getattr(self.settings, 'step_graph_collectives', None) != 'abi':
            if not getattr(self, '_graph_note', False):
                print('[srgan_amd] step_graph under data parallelism needs settings.step_graph_collectives = "abi" '
                      '(the C ABI\'s RCCL entry points); running eagerly')
                self._graph_note = True
            return False
        if getattr(self.dp, 'abi', None) is None and getattr(self.dp, 'device_backend', None) == 'nccl':
            self.dp.use_abi_collectives()
        if getattr(self.dp, 'abi', None) is None:
            return False
        if not getattr(self, '_single_compute_stream', False):
            self.join_dnn_stream()                  # an eager iteration may have left work on a side stream
            self._single_compute_stream = True
            print('[srgan_amd] step_graph under data parallelism: one compute stream + RCCL\'s (side streams off for this run)')
        return True

    def close(self):
        """Releases what the experiment created outside torch: the C ABI's RCCL communicator (before the process group goes)."""
        self.join_dnn_stream()
        abi = getattr(self.dp, 'abi', None) if self.dp is not None else None
        if abi is not None:
            torch.cuda.synchronize()
            abi.close()
            self.dp.abi = None

    def prepare_optimizers(self):
        """Adam for D (with coupled L2), G and DNN (reference srgan.py:131-138) on the flat arenas."""
        if getattr(self.D, '_srgan_arena', None) is None:
            self.gpu_mode()
        d_lr = self.settings.learning_rate
        weight_decay = self.settings.weight_decay
        self.d_optimizer = Adam(self.D._srgan_arena, lr=d_lr, weight_decay=weight_decay)
        self.g_optimizer = Adam(self.G._srgan_arena, lr=d_lr)
        self.dnn_optimizer = Adam(self.DNN._srgan_arena, lr=d_lr, weight_decay=weight_decay)

    def summary_directory(self, name):
        """Rank 0 writes the trial's event files; the other ranks of a data-parallel run log in memory only."""
        if self.dp is not None and self.dp.rank != 0:
            return None
        return os.path.join(self.trial_directory, name)

    def prepare_summary_writers(self):
        self.dnn_summary_writer = SummaryWriter(self.summary_directory('DNN'))
        self.gan_summary_writer = SummaryWriter(self.summary_directory('GAN'))
        for writer in (self.dnn_summary_writer, self.gan_summary_writer):
            writer.summary_period = self.settings.summary_step_period
            writer.steps_to_run = self.settings.steps_to_run

    USER_INPUT_EXCHANGE_PERIOD = 50     # data-parallel runs agree on typed commands every this many steps

    def _poll_stdin(self):
        """'save' / 'quit' typed on stdin since the last poll, without blocking (reference srgan.py:149-163)."""
        commands = set()
        try:
            ready = sys.stdin in select.select([sys.stdin], [], [], 0)[0]
        except (ValueError, OSError):
            return commands
        while ready:
            line = sys.stdin.readline()
            if not line:
                break
            commands.update(word for word in ('save', 'quit') if word in line)
            ready = sys.stdin in select.select([sys.stdin], [], [], 0)[0]
        return commands

    def handle_user_input(self, step):
        """Acts on 'save' / 'quit' (reference srgan.py:149-163).  Under data parallelism rank 0 reads stdin and the ranks
        exchange what it saw every ``USER_INPUT_EXCHANGE_PERIOD`` steps, so that all of them save / stop together
        (one rank leaving the loop alone would leave the others hanging in their next collective)."""
        parallel = self.parallel
        if not parallel:
            commands = self._poll_stdin()
        else:
            if self.dp.rank == 0:
                self._typed_commands = getattr(self, '_typed_commands', set()) | self._poll_stdin()
            if step % self.USER_INPUT_EXCHANGE_PERIOD != 0 and step != self.settings.steps_to_run - 1:
                return
            commands = self.dp.broadcast_object(sorted(getattr(self, '_typed_commands', set())))
            self._typed_commands = set()
        if 'save' in commands:
            self.save_models(step)
            print('\rSaved model for step {}...'.format(step))
        if 'quit' in commands:
            self.signal_quit = True
            print('\rQuit requested after current experiment...')

    def train_mode(self):
        for module in (self.D, self.DNN, self.G):
            module.train()

    def eval_mode(self):
        for module in (self.D, self.DNN, self.G):
            module.eval()

    def gpu_mode(self):
        """Moves each network into its flat parameter / gradient arena on this rank's device."""
        device = current_device()
        for module in (self.D, self.DNN, self.G):
            if getattr(module, '_srgan_arena', None) is None:
                nn.flatten_parameters(module, device)

    def cpu_mode(self):
        raise RuntimeError('the MI355X training step has no CPU mode')

    @staticmethod
    def compare_model_path_for_latest(model_path1, model_path2):
        """Later step wins; a file without a step number wins over all (reference srgan.py:197-219)."""
        if model_path1 is None:
            return model_path2
        if model_path1.group(1) is None:
            return model_path1
        if model_path2.group(1) is None:
            return model_path2
        return model_path1 if int(model_path1.group(1)) > int(model_path2.group(1)) else model_path2

    def latest_checkpoint_path(self):
        """The ``model[_<step>].pth`` with the highest step in ``settings.load_model_path`` (None if there is none)."""
        latest_model = None
        for file_name in os.listdir(self.settings.load_model_path):
            match = re.search(r'model_?(\d+)?\.pth', file_name)
            if match:
                latest_model = self.compare_model_path_for_latest(latest_model, match)
        return os.path.join(self.settings.load_model_path, latest_model.group(0)) if latest_model else None

    def load_models(self, with_optimizers=True):
        """Loads the latest checkpoint of ``settings.load_model_path`` (reference srgan.py:221-251)."""
        if not self.settings.load_model_path:
            return
        model_path = self.latest_checkpoint_path()
        if model_path is None:
            return
        loaded_model = torch.load(model_path, map_location='cpu')
        for network, optimizer in self.CHECKPOINT_PARTS:
            getattr(self, network).load_state_dict(loaded_model[network])
        if with_optimizers:
            for network, optimizer in self.CHECKPOINT_PARTS:
                getattr(self, optimizer).load_state_dict(loaded_model[optimizer])
        print('Model loaded from `{}`.'.format(model_path))
        if self.settings.continue_existing_experiments:
            self.starting_step = loaded_model['step'] + 1
            print(f'Continuing from step {self.starting_step}')

    # ------------------------------------------------------------------------------------------ random draws
    def _take_draw(self, key):
        if self.injected_draws is not None and self.injected_draws.get(key) is not None:
            value = self.injected_draws[key]
            self.injected_draws[key] = None
            return value
        return None

    def _global_draw(self, local_batch, draw):
        """Under data parallelism every rank draws the tensor for the GLOBAL batch from its (identically seeded) host
        stream and keeps its own shard: the ranks' examples differ, and together they are exactly what the
        single-device reference draws at the global batch size."""
        if not self.parallel:
            return draw(local_batch)
        return self.dp.shard(draw(self.dp.global_batch(local_batch)))

    def draw_discriminator_noise(self, batch_size):
        """Host tensor: float64 two-Gaussian mixture from NumPy's global stream cast to float32 (srgan.py:286-289)."""
        offset = self.settings.mean_offset
        return self._global_draw(batch_size, lambda count: torch.tensor(
            MixtureModel([norm(-offset, 1), norm(offset, 1)]).rvs(size=[count, self.G.input_size]).astype(np.float32)))

    def draw_generator_noise(self, batch_size):
        """Host tensor: N(0, 1) from torch's CPU stream (srgan.py:301)."""
        return self._global_draw(batch_size, lambda count: torch.randn(count, self.G.input_size))

    def draw_interpolation_alpha(self, batch_size):
        """Host tensor: U[0, 1) per example.  The reference draws on the device (srgan.py:364), which is not
        reproducible across devices; here it comes from torch's CPU stream and is copied over."""
        return self._global_draw(batch_size, lambda count: torch.rand(count))

    def sample_discriminator_noise(self, batch_size):
        z = self._take_draw('z_d')
        return as_var(self.draw_discriminator_noise(batch_size) if z is None else z, staged=True)

    def sample_generator_noise(self, batch_size):
        z = self._take_draw('z_g')
        return as_var(self.draw_generator_noise(batch_size) if z is None else z, staged=True)

    def sample_interpolation_alpha(self, batch_size):
        alpha = self._take_draw('alpha')
        return as_var((self.draw_interpolation_alpha(batch_size) if alpha is None else alpha).reshape(-1), staged=True)

    # ------------------------------------------------------------------------------------------ batch reductions
    def _global_batch(self, local):
        return local if self.dp is None else self.dp.global_batch(local)

    def batch_mean_of_features(self, features, rank_specific_use=False):
        """Mean over the (global) batch -> shape features.shape[1:] (the ``mean(0)`` of srgan.py:442-443).
        ``rank_specific_use``: what is computed from the mean differs between the ranks (see ``all_reduce_sum_var``)."""
        sums = F.col_sum(features)
        if self.parallel:
            sums = self.dp.all_reduce_sum_var(sums, reduce_backward=rank_specific_use)
        return F.scale(sums, 1.0 / self._global_batch(features.shape[0]))

    def batch_mean_of_examples(self, per_example):
        """Mean over the (global) batch of per-example scalars.  Under data parallelism this is the rank's
        partial sum / global batch: gradients add up over ranks, and logged values are summed over ranks."""
        return F.scale(F.sum_all(per_example), 1.0 / self._global_batch(per_example.shape[0]))

    # ------------------------------------------------------------------------------------------ the hot path
    def dnn_training_step(self, examples, labels, step):
        """One round of DNN training (reference srgan.py:259-271).

        The DNN baseline and the GAN networks never exchange anything inside an iteration, so with
        ``settings.overlap_dnn_step = True`` this step is only ENQUEUED here, on a second HIP stream: its ~3000
        launches -- mostly small kernels on 32x32 / 16x16 planes that cannot fill 256 CUs on their own -- then run
        concurrently with the discriminator / generator step on the main stream (measured: 316.6 -> 306.1 ms per
        step, +3.4 % images/s).  ``gan_training_step`` joins the two streams before it returns; anything else that
        touches the DNN first calls ``join_dnn_stream()``.  Off by default: with two streams in flight a kernel's
        duration no longer measures the kernel (bench.py's per-kernel roofline would read 34 % instead of 40 %)."""
        examples, labels = as_var(examples), as_var(labels)
        self._apply_stream_settings()
        side = self._dnn_side_stream()
        if side is None:
            return self._dnn_training_step(examples, labels, step)
        if not self._batches_are_resident() or torch.cuda.is_current_stream_capturing():
            # the batch was produced on the main stream; while a HIP graph is being captured the wait is also what
            # FORKS the side stream into the capture (the batches then are the graph's static input copies, and the
            # chain is joined again before the capture ends, gan_training_step)
            side.wait_stream(torch.cuda.current_stream())
        # (resident batches, eager: the DNN step of iteration i + 1 may start while iteration i's generator step still runs
        # -- the DNN shares nothing with the GAN networks, so its stream only ever waits for its own previous step)
        with torch.cuda.stream(side):
            self._dnn_training_step(examples, labels, step)

    def _dnn_side_stream(self):
        if not self._stream_setting('overlap_dnn_step') or not examples_on_gpu():
            return None
        if getattr(self, '_dnn_stream', None) is None:
            self._dnn_stream = torch.cuda.Stream()
        return self._dnn_stream

    def _auxiliary_stream(self):
        """A second stream for forward passes that nothing differentiates (``settings.overlap_generator_forwards``)."""
        if not self._stream_setting('overlap_generator_forwards') or not examples_on_gpu() or self.parallel:
            # (under data parallelism the chains run on THREE streams -- main, gradient penalty, DNN step -- so that RCCL's
            # communication stream gets the fourth hardware queue of the HIP runtime to itself: a fifth stream aliases onto a
            # queue and couples two chains)
            return None
        if getattr(self, '_aux_stream', None) is None:
            self._aux_stream = torch.cuda.Stream()
        return self._aux_stream

    def _penalty_stream(self):
        """A stream of its own for the gradient-penalty chain (``settings.overlap_gradient_penalty``; shared-forwards
        schedule; while a HIP graph is captured the fork / join below become edges of the graph; under data parallelism D's gradient exchange then starts after the
        two chains have joined and runs under the generator forward instead of under the penalty's backward): D(interpolates), the recorded gradient w.r.t. them, the double
        backward and the backward through the forward graph (reference srgan.py:294-295) are a chain of batch-sized kernels
        that depends on nothing of the stacked pass over [x, u, fake] (srgan.py:279-292) but the generated images -- the two
        chains run next to each other, each into its own gradient buffer of D's arena."""
        if not self._stream_setting('overlap_gradient_penalty') or not examples_on_gpu() or \
                getattr(self.D, '_srgan_arena', None) is None:
            return None
        if getattr(self, '_gp_stream', None) is None:
            # (SRGAN_GP_STREAM_PRIORITY=-1: the runtime's high priority for this, the longest chain -- an experiment that is NOT the
            # default: +0.7 % on one GPU in the eager four-stream schedule, but a high-priority stream is one more hardware queue:
            # -25 % under data parallelism next to RCCL's stream and -18 % / -47 % when the chains are replayed as a HIP graph,
            # profiles/r05_stream_priority.txt)
            self._gp_stream = torch.cuda.Stream(priority=int(os.environ.get('SRGAN_GP_STREAM_PRIORITY', '0')))
        return self._gp_stream

    def _gradient_penalty_on_its_own_stream(self, stream, fake_examples, unlabeled_examples):
        main = torch.cuda.current_stream()
        stream.wait_stream(main)                          # the generated images, the batch, last step's weight update
        arena = self.D._srgan_arena
        with torch.cuda.stream(stream):
            with arena.gradients_into_alternate() as alternate:
                F.fill_(alternate, 0.0)
                with self.precision('penalty'):
                    gradient_penalty = self.gradient_penalty_calculation(fake_examples, unlabeled_examples)
                    self.scaled_backward(gradient_penalty)
        for var in (gradient_penalty, self.gradient_norm, self.interpolates_features):
            if var is not None:
                var.data.record_stream(main)              # read on the main stream after the join (summaries)
        return gradient_penalty

    def _apply_stream_settings(self):
        """``settings.wgrad_stream`` (None: leave the module default / SRGAN_WGRAD_STREAM) -> ``fused.WGRAD_STREAM``."""
        wanted = getattr(self.settings, 'wgrad_stream', None)
        if getattr(self, '_single_compute_stream', False):
            wanted = False
        if wanted is not None:
            from . import fused
            fused.WGRAD_STREAM = bool(wanted)

    def _batches_are_resident(self):
        return bool(getattr(self.train_dataset_loader, 'resident', False))

    def _join_side_stream(self):
        stream = getattr(self, '_dnn_stream', None)
        if stream is not None:
            torch.cuda.current_stream().wait_stream(stream)

    def join_dnn_stream(self):
        """Settle the networks before anything outside the training step reads them: the current stream waits for an
        enqueued DNN step, and optimizer updates still waiting for their gradient exchange are applied."""
        self.finish_update()
        self._join_side_stream()

    # ---- data-parallel gradient exchange, overlapped with the step (no-ops on one device) ----------------------
    def gradient_exchange(self, module):
        """The asynchronous all-reduce of ``module``'s gradient arena for the LAST backward pass into it (handed to
        ``tape.backward(grad_ready=...)``); None on a single device."""
        if not self.parallel:
            return None
        # settings.gradient_wire_dtype ('f32' | 'bf16': buckets travel as bf16, the master gradients stay fp32) and
        # settings.gradient_exchange_form ('all_reduce' | 'reduce_scatter'), opt-in per configuration
        return self.dp.gradient_exchange(module._srgan_arena, wire=getattr(self.settings, 'gradient_wire_dtype', None),
                                         form=getattr(self.settings, 'gradient_exchange_form', None))

    # ---- mixed precision (BASELINE.json configs 2 and 5; the reference is fp32-only) --------------------------------
    def precision(self, phase='step'):
        """The MFMA operand type of a phase as a context manager: ``settings.compute_dtype`` ('f32' default, 'bf16',
        'f16') for the step, ``settings.gradient_penalty_dtype`` (default 'f32': "fp16 with fp32 GP") for the
        gradient-penalty chain -- its forward, the recorded inner gradient and the penalty's own backward."""
        name = getattr(self.settings, 'compute_dtype', 'f32')
        if phase == 'penalty':
            name = getattr(self.settings, 'gradient_penalty_dtype', 'f32')
        # settings.storage_dtype ('bf16' / 'f16' / None): the networks with a 16-bit data path (blocked16) keep activations
        # and gradients in that type while the phase computes in it; a phase in another type (an fp32 penalty chain) does not
        storage = getattr(self.settings, 'storage_dtype', None)
        if storage and F.COMPUTE_DTYPES[storage] == F.COMPUTE_DTYPES[name]:
            return _Contexts(F.compute_dtype(name), F.storage_dtype(storage))
        # settings.blocked_fp32: a phase that computes in fp32 runs the DCGAN stacks' 4x4 / stride 2 stages on fp32 tensors in
        # the blocked layout (exact arithmetic on the LDS-DMA kernels of csrc/blocked16_k4s2.hip)
        if F.COMPUTE_DTYPES[name] == 0 and getattr(self.settings, 'blocked_fp32', False):
            return _Contexts(F.compute_dtype(name), F.storage_dtype('f32b'))
        return _Contexts(F.compute_dtype(name), F.storage_dtype(None))

    def scaled_backward(self, root, **arguments):
        """``backward(root)`` with the root gradient set to ``settings.loss_scale`` (static loss scaling for the fp16 mode:
        gradients of 1e-6 would otherwise reach the fp16 operands as subnormals); ``apply_update`` divides it out."""
        scale = float(getattr(self.settings, 'loss_scale', 1.0))
        return backward(root, grad=None if scale == 1.0 else F.full_like(root, scale), **arguments)

    def apply_update(self, optimizer):
        scale = float(getattr(self.settings, 'loss_scale', 1.0))
        if scale != 1.0:
            F._unary_raw(F.U_AFFINE, optimizer.arena.grad, 1.0 / scale, 0.0, out=optimizer.arena.grad)
        optimizer.step()

    def start_update(self, name, optimizer, exchange):
        """The optimizer step of network ``name``: at once on a single device; under data parallelism after its gradient
        exchange, i.e. in ``finish_update`` -- the rest of the arena goes out now and the step continues meanwhile."""
        if exchange is None:
            self.apply_update(optimizer)
            return
        exchange.finish()
        if not getattr(self.settings, 'overlap_gradient_exchange', True):
            exchange.wait()
            self.apply_update(optimizer)
            return
        self._pending_updates[name] = (exchange, optimizer)

    def finish_update(self, *names):
        """Wait for the gradient exchange of the named networks (default: all) and apply their optimizer updates."""
        for name in (names or tuple(self._pending_updates)):
            pending = self._pending_updates.pop(name, None)
            if pending is None:
                continue
            exchange, optimizer = pending
            side = self._dnn_side_stream() if name == 'DNN' else None
            if side is not None:                    # the DNN's step and exchange live on the side stream
                with torch.cuda.stream(side):
                    exchange.wait()
                    self.apply_update(optimizer)
            else:
                exchange.wait()
                self.apply_update(optimizer)

    def _dnn_training_step(self, examples, labels, step):
        self.DNN.apply(disable_batch_norm_updates)
        self.dnn_summary_writer.step = step
        self.finish_update('DNN')
        self.dnn_optimizer.zero_grad()
        exchange = self.gradient_exchange(self.DNN)
        with self.precision():
            dnn_loss = self.dnn_loss_calculation(examples, labels)
            self.scaled_backward(dnn_loss, grad_ready=exchange)
        self.start_update('DNN', self.dnn_optimizer, exchange)       # finished at the end of gan_training_step
        self.last_losses['dnn_loss'] = dnn_loss
        if self.dnn_summary_writer.is_summary_step():
            self.dnn_summary_writer.add_scalar('Discriminator/Labeled Loss', self.loss_value(dnn_loss, partial=True))
            if getattr(self.DNN, 'features', None) is not None:
                norms = F.row_norm(F.flatten2d(self.DNN.features.detach()))
                self.dnn_summary_writer.add_scalar('Feature Norm/Labeled', F.mean_all(norms).item())

    def gan_training_step(self, labeled_examples, labels, unlabeled_examples, step):
        """One round of GAN training (reference srgan.py:273-320)."""
        settings = self.settings
        labeled_examples, labels = as_var(labeled_examples), as_var(labels)
        unlabeled_examples = as_var(unlabeled_examples)
        self._apply_stream_settings()
        self.D.apply(disable_batch_norm_updates)
        self.gan_summary_writer.step = step
        self.finish_update('G', 'D')         # the previous iteration's generator update (its exchange ran under the DNN step)
        self.d_optimizer.zero_grad()
        batch_size = unlabeled_examples.shape[0]
        penalty_stream = None
        with self.precision():
            if getattr(settings, 'reference_schedule', False):
                labeled_loss = self.labeled_loss_calculation(labeled_examples, labels)
                self.scaled_backward(labeled_loss)
                unlabeled_loss = self.unlabeled_loss_calculation(labeled_examples, unlabeled_examples)
                self.scaled_backward(unlabeled_loss)
                z = self.sample_discriminator_noise(batch_size)
                with no_grad():
                    fake_examples = self.G(z)
                fake_loss = self.fake_loss_calculation(unlabeled_examples, fake_examples)
                self.scaled_backward(fake_loss)
            else:
                z = self.sample_discriminator_noise(batch_size)
                with no_grad():
                    fake_examples = self.G(z)
                penalty_stream = self._penalty_stream()
                if penalty_stream is not None:
                    gradient_penalty = self._gradient_penalty_on_its_own_stream(penalty_stream, fake_examples,
                                                                                unlabeled_examples)
                labeled_loss, unlabeled_loss, fake_loss = self.discriminator_losses_shared_forwards(
                    labeled_examples, labels, unlabeled_examples, fake_examples)
                self.scaled_backward(F.add(F.add(labeled_loss, unlabeled_loss), fake_loss))
        generator_phase = step % settings.generator_training_step_period == 0
        generator_fake = None
        if generator_phase and penalty_stream is not None:
            # The generator's forward pass of the generator step (srgan.py:300-302) depends on nothing the discriminator step
            # produces -- only D(fake) needs the updated discriminator -- so it is enqueued HERE, behind the stacked pass's
            # backward: the main stream would otherwise idle until the (longer) penalty chain has joined (measured: the
            # stacked chain ends 16 ms before the penalty chain).  The random draws keep the reference's order (z_D, alpha, z_G).
            self.g_optimizer.zero_grad()
            with self.precision():
                generator_fake = self.G(self.sample_generator_noise(batch_size))
        exchange = self.gradient_exchange(self.D)
        if penalty_stream is not None:
            # the penalty chain accumulated into the arena's second gradient buffer on its own stream: join and add
            torch.cuda.current_stream().wait_stream(penalty_stream)
            arena = self.D._srgan_arena
            F._binary_raw(F.B_ADD, arena.grad, arena._alternate, out=arena.grad)
        else:
            with self.precision('penalty'):
                gradient_penalty = self.gradient_penalty_calculation(fake_examples, unlabeled_examples)
                self.scaled_backward(gradient_penalty, grad_ready=exchange)   # the last backward pass into D's arena (srgan.py:295)
        self.start_update('D', self.d_optimizer, exchange)
        generator_loss = None
        if generator_phase:
            with self.precision():
                if generator_fake is None:
                    self.g_optimizer.zero_grad()
                    generator_fake = self.G(self.sample_generator_noise(batch_size))   # runs while D's gradients are still being exchanged
                self.finish_update('D')
                generator_loss = self.generator_loss_calculation(generator_fake, unlabeled_examples)
                exchange = self.gradient_exchange(self.G)
                self.scaled_backward(generator_loss, grad_ready=exchange)
            self.start_update('G', self.g_optimizer, exchange)      # finished when G is next used (next iteration)
        self.finish_update('D', 'DNN')
        self.last_losses.update(labeled_loss=labeled_loss, unlabeled_loss=unlabeled_loss, fake_loss=fake_loss,
                                gradient_penalty=gradient_penalty, generator_loss=generator_loss)
        if not self._batches_are_resident() or self.gan_summary_writer.is_summary_step() or \
                self.dnn_summary_writer.is_summary_step() or torch.cuda.is_current_stream_capturing():
            # (a batch that is freed after the iteration must outlive its DNN step; resident batches let the DNN stream
            # run on into the next iteration -- everything that reads the DNN joins it first, join_dnn_stream(); a HIP
            # graph capture must end with every forked stream joined)
            self._join_side_stream()
        if self.gan_summary_writer.is_summary_step():
            writer = self.gan_summary_writer
            if generator_loss is not None:
                writer.add_scalar('Generator/Loss', self.loss_value(generator_loss))
            writer.add_scalar('Discriminator/Labeled Loss', self.loss_value(labeled_loss, partial=True))
            writer.add_scalar('Discriminator/Unlabeled Loss', self.loss_value(unlabeled_loss))
            writer.add_scalar('Discriminator/Fake Loss', self.loss_value(fake_loss))
            writer.add_scalar('Discriminator/Gradient Penalty', self.loss_value(gradient_penalty, partial=True))
            writer.add_scalar('Discriminator/Gradient Norm', self.loss_value(
                self.batch_mean_of_examples(self.gradient_norm.detach()), partial=True))
            if self.labeled_features is not None and self.unlabeled_features is not None:
                with no_grad():
                    for tag, features in (('Labeled', self.labeled_features), ('Unlabeled', self.unlabeled_features)):
                        mean = self.batch_mean_of_features(features.detach())
                        writer.add_scalar('Feature Norm/' + tag, F.sqrt(F.sum_all(F.square(mean))).item())

    def _distance_over_the_ranks_rows(self, distance_function, difference):
        from . import utility
        mean_type = (utility.abs_mean, utility.abs_mean_neg, utility.abs_plus_one_log_mean_neg,
                     utility.abs_plus_one_sqrt_mean_neg, utility.square_mean)
        if distance_function in mean_type:
            return F.scale(self.dp.all_reduce_sum_var(distance_function(difference)), 1.0 / self.dp.world_size)
        if distance_function is utility.norm_mean:
            return F.sqrt(self.dp.all_reduce_sum_var(F.sum_all(F.square(difference))))
        raise NotImplementedError(f'normalize_feature_norm under data parallelism: no rule to assemble '
                                  f'{getattr(distance_function, "__name__", distance_function)} from the ranks\' rows')

    def loss_value(self, loss, partial=False):
        """Host value of a device scalar; ``partial`` scalars are per-rank partial sums under data parallelism."""
        value = float(loss.item())
        if partial and self.parallel:
            value = self.dp.all_reduce_sum_float(value)
        return value

    def synchronize_gradients(self, module):
        """Blocking form of the gradient exchange (methods that do not overlap it: the DNN-only experiment)."""
        if self.parallel:
            self.dp.all_reduce_gradients(module._srgan_arena)

    def discriminator_losses_shared_forwards(self, labeled_examples, labels, unlabeled_examples, fake_examples):
        """Labeled + unlabeled + fake losses from ONE discriminator pass over the three batches stacked along the
        batch dimension; same mathematics as srgan.py:329-358, which runs five separate forwards (the discriminator
        acts per example: its batch-norm layers are frozen, srgan.py:276).  One pass over 3B examples instead of
        three over B triples the work per kernel launch, which is what the 16x16 / 32x32 dense blocks lack."""
        settings = self.settings
        sizes = (labeled_examples.shape[0], unlabeled_examples.shape[0], fake_examples.shape[0])
        stackable = getattr(settings, 'batched_discriminator', True) and not getattr(self, '_stack_too_large', False) and \
            tuple(labeled_examples.shape[1:]) == tuple(unlabeled_examples.shape[1:]) == tuple(fake_examples.shape[1:])
        if stackable:
            # Stacking triples the batch of every activation; one of them may then pass the 2^31-element limit of the
            # kernels' 32-bit offsets although the three batches fit on their own (e.g. VGG-16 at batch 600: 1800 x 64 x
            # 128 x 128).  The library reports that as SRGAN_ERANGE; the three separate passes are the same mathematics.
            try:
                stacked = F.cat_rows([labeled_examples, unlabeled_examples, fake_examples.detach()])
                predicted = self.D(stacked)
                features = self.D.features
            except _lib.HipLibraryError as error:
                if error.status != _lib.ERANGE:
                    raise
                print(f'stacked discriminator pass too large ({error}); using three separate passes')
                self._stack_too_large = True
                stackable = False
        if stackable:
            take = lambda value, first, count: (tuple(F.narrow_rows(v, first, count) for v in value)
                                                if isinstance(value, (tuple, list)) else F.narrow_rows(value, first, count))
            predicted_labels = take(predicted, 0, sizes[0])
            self.labeled_features = F.narrow_rows(features, 0, sizes[0])
            self.unlabeled_features = F.narrow_rows(features, sizes[0], sizes[1])
            self.fake_features = F.narrow_rows(features, sizes[0] + sizes[1], sizes[2])
        else:
            predicted_labels = self.D(labeled_examples)
            self.labeled_features = self.D.features
            _ = self.D(unlabeled_examples)
            self.unlabeled_features = self.D.features
            _ = self.D(fake_examples.detach())
            self.fake_features = self.D.features
        labeled_loss = self.labeled_loss_function(predicted_labels, labels, order=settings.labeled_loss_order)
        labeled_loss = F.scale(labeled_loss, settings.labeled_loss_multiplier)
        unlabeled_loss = self.feature_distance_loss(self.unlabeled_features, self.labeled_features)
        unlabeled_loss = F.scale(unlabeled_loss, settings.matching_loss_multiplier * settings.srgan_loss_multiplier)
        fake_loss = self.feature_distance_loss(self.unlabeled_features, self.fake_features,
                                               distance_function=settings.contrasting_distance_function)
        fake_loss = F.scale(fake_loss, settings.contrasting_loss_multiplier * settings.srgan_loss_multiplier)
        return labeled_loss, unlabeled_loss, fake_loss

    def dnn_loss_calculation(self, labeled_examples, labels):
        """reference srgan.py:322-327."""
        predicted_labels = self.DNN(labeled_examples)
        labeled_loss = self.labeled_loss_function(predicted_labels, labels, order=self.settings.labeled_loss_order)
        return F.scale(labeled_loss, self.settings.labeled_loss_multiplier)

    def labeled_loss_calculation(self, labeled_examples, labels):
        """reference srgan.py:329-335."""
        predicted_labels = self.D(labeled_examples)
        self.labeled_features = self.D.features
        labeled_loss = self.labeled_loss_function(predicted_labels, labels, order=self.settings.labeled_loss_order)
        return F.scale(labeled_loss, self.settings.labeled_loss_multiplier)

    def unlabeled_loss_calculation(self, labeled_examples, unlabeled_examples):
        """reference srgan.py:337-346."""
        _ = self.D(labeled_examples)
        self.labeled_features = self.D.features
        _ = self.D(unlabeled_examples)
        self.unlabeled_features = self.D.features
        unlabeled_loss = self.feature_distance_loss(self.unlabeled_features, self.labeled_features)
        return F.scale(unlabeled_loss, self.settings.matching_loss_multiplier * self.settings.srgan_loss_multiplier)

    def fake_loss_calculation(self, unlabeled_examples, fake_examples):
        """reference srgan.py:348-358."""
        _ = self.D(unlabeled_examples)
        self.unlabeled_features = self.D.features
        _ = self.D(fake_examples.detach())
        self.fake_features = self.D.features
        fake_loss = self.feature_distance_loss(self.unlabeled_features, self.fake_features,
                                               distance_function=self.settings.contrasting_distance_function)
        return F.scale(fake_loss, self.settings.contrasting_loss_multiplier * self.settings.srgan_loss_multiplier)

    def gradient_penalty_calculation(self, fake_examples, unlabeled_examples):
        """Interpolated-sample gradient penalty with its double backward (reference srgan.py:360-375).
        alpha has ``settings.batch_size`` entries as in the reference (:363), i.e. the unlabeled batch must
        have exactly that many examples (per rank under data parallelism)."""
        settings = self.settings
        expected = settings.batch_size if self.dp is None else self.dp.local_batch(settings.batch_size)
        alpha = self.sample_interpolation_alpha(expected)
        interpolates = F.gp_interpolate(unlabeled_examples.detach(), fake_examples.detach(), alpha)
        interpolates_loss = self.interpolate_loss_calculation(interpolates)
        gradients, = backward(interpolates_loss, grad=F.full_like(interpolates_loss, 1.0), inputs=[interpolates],
                              create_graph=True)
        gradient_norm = F.row_norm(F.flatten2d(gradients))
        self.gradient_norm = gradient_norm
        norm_excesses = F.relu(F.add_scalar(gradient_norm, -1.0))
        penalty = self.batch_mean_of_examples(F.square(norm_excesses))
        return F.scale(penalty, settings.gradient_penalty_multiplier)

    def interpolate_loss_calculation(self, interpolates):
        """Per-example feature norm of the interpolates (reference srgan.py:377-381)."""
        _ = self.D(interpolates)            # differentiated twice: every node's backward is itself recorded
        self.interpolates_features = self.D.features
        return F.row_norm(F.flatten2d(self.interpolates_features))

    def generator_loss_calculation(self, fake_examples, unlabeled_examples):
        """reference srgan.py:383-391 (no srgan_loss_multiplier, Appendix A.4)."""
        with nn.frozen_parameters(self.D):
            side = self._auxiliary_stream()
            if side is None:
                _ = self.D(fake_examples)
                self.fake_features = self.D.features
                with no_grad():
                    _ = self.D(unlabeled_examples)
                    detached_unlabeled_features = self.D.features.detach()
            else:
                # D(u) needs no gradient and shares nothing with D(fake) but the (frozen) weights: its batch-sized kernels
                # run on a second stream next to D(fake)'s, which on their own cannot fill the GPU on the small planes
                main = torch.cuda.current_stream()
                side.wait_stream(main)                   # the discriminator update and the batch are enqueued on main
                with torch.cuda.stream(side), no_grad():
                    _ = self.D(unlabeled_examples)
                    detached_unlabeled_features = self.D.features.detach()
                detached_unlabeled_features.data.record_stream(main)
                _ = self.D(fake_examples)
                self.fake_features = self.D.features
                main.wait_stream(side)
        generator_loss = self.feature_distance_loss(detached_unlabeled_features, self.fake_features)
        return F.scale(generator_loss, self.settings.matching_loss_multiplier)

    # ------------------------------------------------------------------------------------------ plug-in hooks
    @abstractmethod
    def dataset_setup(self):
        """Prepares the datasets and loaders (reference srgan.py:393-400)."""

    @abstractmethod
    def model_setup(self):
        """Assigns self.DNN, self.D and self.G (reference srgan.py:402-407)."""

    @abstractmethod
    def validation_summaries(self, step: int):
        """Per-application evaluation summaries (reference srgan.py:409-412)."""

    def labeled_loss_function(self, predicted_labels, labels, order=2):
        """mean(|p - y| ** order) (reference srgan.py:414-417).  A (B, 1) prediction against (B) labels
        broadcasts to (B, B) exactly as in the reference's VGG path (Appendix A.12)."""
        if len(predicted_labels.shape) == 2 and predicted_labels.shape[1] == 1 and len(labels.shape) == 1:
            batch = predicted_labels.shape[0]
            rows = F.row_broadcast(F.view(predicted_labels, (batch,)), (batch, batch))
            columns = F.col_broadcast(labels, batch)
            difference = F.sub(rows, columns)
            return F.scale(F.sum_all(F.pow_scalar(F.abs_(difference), order)),
                           1.0 / (batch * self._global_batch(batch)))
        difference = F.sub(predicted_labels, F.view(labels, predicted_labels.shape))
        return self.batch_mean_of_examples(F.pow_scalar(F.abs_(difference), order))

    def evaluate(self):
        self.model_setup()
        self.gpu_mode()
        self.load_models(with_optimizers=False)
        self.eval_mode()

    def regression_evaluation_epoch(self, network, dataset, summary_writer, summary_name, comparison_value=None,
                                    normalized=False):
        """MAE / MSE (and NMAE = MAE / label range with ``normalized``) of ``network`` over ``dataset`` -- any
        iterable of ``(examples, labels)`` batches -- as summary scalars; ratio to ``comparison_value`` (the DNN's MAE)
        when given.  Forward passes only (reference age/srgan.py:92-107, driving/srgan.py:87-104)."""
        predictions, labels = [], []
        self.join_dnn_stream()
        with no_grad():
            for examples, batch_labels in dataset:
                predicted = network(as_var(examples))
                predictions.append(predicted.cpu().numpy().reshape(-1).astype(np.float64))
                labels.append(np.asarray(batch_labels.cpu() if hasattr(batch_labels, 'cpu') else batch_labels,
                                         dtype=np.float64).reshape(-1))
        predictions, labels = np.concatenate(predictions), np.concatenate(labels)
        mae = float(np.abs(predictions - labels).mean())
        summary_writer.add_scalar('{}/MAE'.format(summary_name), mae)
        if normalized:
            summary_writer.add_scalar('{}/NMAE'.format(summary_name), mae / float(labels.max() - labels.min()))
        summary_writer.add_scalar('{}/MSE'.format(summary_name), float((np.abs(predictions - labels) ** 2).mean()))
        if comparison_value is not None:
            summary_writer.add_scalar('{}/Ratio MAE GAN DNN'.format(summary_name), mae / comparison_value)
        return mae

    def regression_validation_summaries(self, normalized=False):
        """DNN and D on the train and validation batches, D's validation MAE relative to the DNN's (reference
        age/srgan.py:52-71, driving/srgan.py:48-67; the image grids are out of scope)."""
        train, validation = self.train_dataset_loader, self.validation_dataset_loader
        self.regression_evaluation_epoch(self.DNN, train, self.dnn_summary_writer, '2 Train Error', normalized=normalized)
        dnn_mae = self.regression_evaluation_epoch(self.DNN, validation, self.dnn_summary_writer, '1 Validation Error',
                                                   normalized=normalized)
        self.regression_evaluation_epoch(self.D, train, self.gan_summary_writer, '2 Train Error', normalized=normalized)
        self.regression_evaluation_epoch(self.D, validation, self.gan_summary_writer, '1 Validation Error',
                                         comparison_value=dnn_mae, normalized=normalized)

    @staticmethod
    def infinite_iter(dataset):
        while True:
            for examples in dataset:
                yield examples

    def adjust_learning_rate(self, step):
        """lr * 0.1 ** (step // 100000), applied to the DNN optimizer only (reference srgan.py:432-436)."""
        lr = self.settings.learning_rate * (0.1 ** (step // 100000))
        for param_group in self.dnn_optimizer.param_groups:
            param_group['lr'] = lr

    def feature_distance_loss(self, base_features, other_features, distance_function=None):
        """distance(mean_b(base) - mean_b(other)) (reference srgan.py:438-449)."""
        if distance_function is None:
            distance_function = self.settings.matching_distance_function
        normalize = self.settings.normalize_feature_norm
        base_mean_features = self.batch_mean_of_features(base_features, rank_specific_use=normalize)
        other_mean_features = self.batch_mean_of_features(other_features, rank_specific_use=normalize)
        if normalize:
            # The reference's branch AS WRITTEN (srgan.py:444-447): the base mean is divided by its norm, but line 447
            # divides the un-meaned ``other_features`` (B, F) by the norm of their mean, so the difference broadcasts to
            # (B, F) and the distance function averages over examples as well.  Off by default (settings.py:39).
            # Under data parallelism a rank holds its own rows of that (B, F) difference: the distance is assembled from the
            # ranks' parts (a mean over all elements = the mean of the equally sized shards' means; a norm = the root of the
            # summed squares), and the two batch means pass their gradients through an all-reduce of their own.
            epsilon = 1e-5

            def inverse_norm(vector):
                norm = F.sqrt(F.sum_all(F.square(vector)))
                return F.div(F.full_like(norm, 1.0), F.add_scalar(norm, epsilon))
            base_normalized = F.scalar_mul(base_mean_features, inverse_norm(base_mean_features))
            other_rows = F.flatten2d(other_features)
            other_normalized = F.scalar_mul(other_rows, inverse_norm(other_mean_features))
            difference = F.sub(F.col_broadcast(base_normalized, other_rows.shape[0]), other_normalized)
            if self.parallel:
                return self._distance_over_the_ranks_rows(distance_function, difference)
            return distance_function(difference)
        return distance_function(F.sub(base_mean_features, other_mean_features))

    @property
    def inference_network(self):
        return self.D

    def inference_setup(self):
        self.model_setup()
        self.gpu_mode()
        self.load_models(with_optimizers=False)
        self.eval_mode()

    def inference(self, input_):
        raise NotImplementedError


class _Contexts:
    """Several context managers entered / left together."""

    def __init__(self, *managers):
        self.managers = managers

    def __enter__(self):
        for manager in self.managers:
            manager.__enter__()
        return self

    def __exit__(self, *exc):
        for manager in reversed(self.managers):
            manager.__exit__(*exc)


def disable_batch_norm_updates(module):
    """Every batch-norm layer in eval mode (reference srgan.py:538-542).  The HIP batch-norm always uses the
    running statistics, so this only keeps ``module.training`` flags consistent with the reference."""
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
        module.eval()


def enable_batch_norm_updates(module):
    """reference srgan.py:545-549 (never used on the hot path: the re-enable is commented out upstream)."""
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
        module.train()
