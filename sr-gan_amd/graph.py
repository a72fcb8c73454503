"""One training iteration captured ONCE as a HIP graph and replayed (``settings.step_graph = True``).

An iteration of reference srgan.py:104-118 is ~5000 kernel launches whose arguments do not depend on the step: the
shapes are fixed, the parameters / gradient arenas / Adam moments live at fixed addresses, and the tape makes the same
decisions every time.  At 512x512 the GPU is the bottleneck and the Python tape hides behind it; at the reference's own
224x224 the host needs ~94 ms to enqueue ~75 ms of GPU work.  Capturing the iteration on a HIP stream
(``hipStreamBeginCapture`` through ``torch.cuda.graph``: every launch of libsrgan_hip.so goes to torch's current stream,
which is the capturing one) and replaying the instantiated graph removes the host from the loop.

What changes between iterations enters through fixed device buffers that are refreshed before each replay:
* the labeled / unlabeled batches (device-to-device copies into the captured input tensors);
* the three random draws, made on the HOST from the same streams and in the same order as the eager step (z for the
  discriminator step, alpha, z for the generator step), so a replayed run consumes the generators exactly as an eager one;
* Adam's update count, kept on the device (``Adam.count_on_device`` / ``srgan_adam_step_counted``).

The side streams of the eager schedule (``settings.overlap_dnn_step`` / ``overlap_gradient_penalty`` /
``overlap_generator_forwards``) are captured too: a ``wait_stream`` on the capturing stream forks a chain into the capture,
the joins at the end of the step close it, and the instantiated graph holds the four chains as parallel branches -- they start
together on replay instead of in the order Python reaches them.

Iterations whose host-side behaviour differs run eagerly: summary steps (they read losses back), and a graph is keyed
by everything the captured launches baked in (input shapes, whether the generator trains this step, learning rates).
Data-parallel runs (round 5): the exchanges are captured too when they are launches on streams this process owns -- the C
ABI's RCCL entry points on the communicator's stream (``parallel.AbiCommunicator``; forked into the capture by the event
wait in front of a collective, joined by the ``wait()`` of its work object) -- so that eight Python ranks on a 16-CPU
quota replay instead of enqueueing ~4000 launches each; every captured iteration ends with ALL pending optimizer updates
applied (the generator's exchange, which the eager loop hides under the next iteration's DNN step, completes inside the
graph).  Over gloo (host collectives) a data-parallel run stays eager.
"""
import torch

from . import functional as F
from .tape import Var

# attributes a training step leaves on the experiment for summaries / tests; after a replay they must again refer to
# the captured tensors (an eager summary step in between rebinds them)
MAX_RECORDS = 4          # captured graphs kept per experiment (they share one memory pool)
STEP_OUTPUTS = ('last_losses', 'gradient_norm', 'labeled_features', 'unlabeled_features', 'fake_features',
                'interpolates_features')


def _copied_out(value):
    if isinstance(value, dict):
        return {key: _copied_out(item) for key, item in value.items()}
    if isinstance(value, Var):
        return Var(F._unary_raw(F.U_COPY, value.data))
    return value


def _flatten(value, out):
    if isinstance(value, (tuple, list)):
        for item in value:
            _flatten(item, out)
    elif value is not None:
        out.append(value)
    return out


def _rebuild(value, replacements):
    if isinstance(value, (tuple, list)):
        return tuple(_rebuild(item, replacements) for item in value)
    return None if value is None else next(replacements)


class CapturedIteration:
    def __init__(self, experiment):
        self.experiment = experiment
        self.records = {}
        self.eager_iterations = 0
        self.replays = 0

    # -------------------------------------------------------------------------------------------------------------
    def optimizers(self):
        e = self.experiment
        return e.dnn_optimizer, e.d_optimizer, e.g_optimizer

    def order_side_streams(self):
        """Replays run on the current stream, eager iterations in between put the DNN step on its side stream without
        joining it (resident batches): at every switch between the two the streams wait for each other, so that an eager
        DNN step never runs next to a replay's and the other way round."""
        side = getattr(self.experiment, '_dnn_stream', None)
        if side is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            side.wait_stream(main)

    def eager(self, labeled_examples, labels, unlabeled_examples, step):
        e = self.experiment
        self.order_side_streams()
        e.dnn_training_step(labeled_examples, labels, step)
        e.gan_training_step(labeled_examples, labels, unlabeled_examples, step)
        self.eager_iterations += 1

    def run(self, labeled_examples, labels, unlabeled_examples, step):
        from .srgan import as_var
        e = self.experiment
        settings = e.settings
        inputs = as_var((labeled_examples, labels, unlabeled_examples))
        for writer in (e.dnn_summary_writer, e.gan_summary_writer):
            writer.step = step
        summary = e.dnn_summary_writer.is_summary_step() or e.gan_summary_writer.is_summary_step()
        if summary or self.eager_iterations < int(getattr(settings, 'step_graph_warmup', 2)) or \
                getattr(settings, 'reference_schedule', False):
            return self.eager(*inputs, step)
        flat = _flatten(inputs, [])
        generator_phase = step % settings.generator_training_step_period == 0
        key = (tuple(tuple(v.shape) for v in flat), generator_phase, F.COMPUTE_DTYPE,
               tuple(o.param_groups[0]['lr'] for o in self.optimizers()))
        batch = inputs[2].shape[0]
        # host draws of THIS iteration, in the order the eager step makes them (srgan.py:286, :364, :301)
        # (a draw injected by a test is used once, exactly as the eager step would use it)
        def drawn(name, draw):
            injected = e._take_draw(name)
            return draw() if injected is None else injected
        host = {'z_d': drawn('z_d', lambda: e.draw_discriminator_noise(batch)),
                'alpha': drawn('alpha', lambda: e.draw_interpolation_alpha(settings.batch_size)).reshape(-1)}
        if generator_phase:
            host['z_g'] = drawn('z_g', lambda: e.draw_generator_noise(batch))
        record = self.records.get(key)
        if record is None:
            record = self.records[key] = self.capture(inputs, flat, host, step)
        for target, source in zip(record['inputs'], flat):
            F._unary_raw(F.U_COPY, source.data, out=target.data)
        for name, value in host.items():
            record['draws'][name].copy_(value, non_blocking=True)
        self.order_side_streams()
        # An eager iteration in between (a summary step, a warm-up) may have left an optimizer update waiting for its
        # gradient exchange (``_pending_updates``); the captured ``finish_update`` was a no-op when it was recorded, so it is
        # settled here, before the replay overwrites those gradients.
        e.finish_update()
        record['graph'].replay()
        for optimizer, advanced in zip(self.optimizers(), record['advanced']):
            optimizer.step_count += advanced
        # The step outputs (losses, features, gradient norms: a few hundred floats) are COPIED out of the graphs' shared
        # memory pool: a later replay -- of this graph or of another one that reuses the pool as scratch -- may overwrite
        # the static tensors, and a caller may keep an iteration's losses beyond the next one.
        for name, value in record['outputs'].items():
            setattr(e, name, _copied_out(value))
        self.replays += 1

    def capture(self, inputs, flat, host, step):
        e = self.experiment
        device = flat[0].data.device
        static = [Var(torch.empty_like(v.data)) for v in flat]
        draws = {name: torch.empty(value.shape, dtype=torch.float32, device=device) for name, value in host.items()}
        optimizers = self.optimizers()
        for optimizer in optimizers:
            optimizer.count_on_device()
        before = [optimizer.step_count for optimizer in optimizers]
        static_inputs = _rebuild(inputs, iter(static))
        injected, e.injected_draws = e.injected_draws, dict(draws)
        graph = torch.cuda.CUDAGraph()
        # ONE memory pool for every captured graph of this experiment: replays are strictly sequential and the step outputs
        # are read before the next replay, so the graphs (generator / no-generator phase, other learning rates, a ragged last
        # batch) can share their activation memory instead of holding a private copy each.
        pool = next(iter(self.records.values()))['graph'].pool() if self.records else None
        if len(self.records) >= MAX_RECORDS:                  # bounded: the oldest capture (and its tensors) goes
            self.records.pop(next(iter(self.records)))
        advanced = None
        e.join_dnn_stream()                                   # nothing eager in flight on a side stream the capture forks
        try:
            with torch.cuda.graph(graph, pool=pool):
                e.dnn_training_step(static_inputs[0], static_inputs[1], step)
                e.gan_training_step(static_inputs[0], static_inputs[1], static_inputs[2], step)
                # data parallel: a replay is a closed unit -- the generator's gradient exchange and update, which the
                # eager loop finishes when G is next used (under the next iteration's DNN step), end inside the graph
                e.finish_update()
            advanced = [optimizer.step_count - count for optimizer, count in zip(optimizers, before)]
        finally:
            e.injected_draws = injected
            for optimizer, count in zip(optimizers, before):   # capturing recorded the launches, it did not run them
                optimizer.step_count = count                   # (also when the capture raised)
        outputs = {}
        for name in STEP_OUTPUTS:
            value = getattr(e, name, None)
            outputs[name] = dict(value) if isinstance(value, dict) else value
        return dict(graph=graph, inputs=static, draws=draws, advanced=advanced, outputs=outputs)
