"""settings.step_graph: one training iteration captured as a HIP graph and replayed (srgan_amd/graph.py) must leave the
same weights, Adam state and losses as the eager Python tape on the same inputs and host random streams."""
import numpy as np
import pytest
import torch

from test_steps_gpu import make_experiment, finish_setup, crowd_inputs

pytestmark = pytest.mark.gpu


def _run(step_graph, iterations, size=64, batch=2, summary_period=10 ** 9, d_scale=1.27):
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.models import DCGenerator, KnnDenseNetCat
    from srgan_amd.utility import seed_all
    experiment = make_experiment(
        lambda: (DCGenerator(image_size=size), KnnDenseNetCat(image_size=size), KnnDenseNetCat(image_size=size)),
        dict(batch_size=batch, matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2, gradient_penalty_multiplier=1e2,
             map_multiplier=1e-3, step_graph=step_graph, step_graph_warmup=1, steps_to_run=10 ** 9), crowd=True)
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                module.weight.mul_(d_scale)                    # gradient penalty active
    finish_setup(experiment)
    for optimizer in (experiment.d_optimizer, experiment.g_optimizer, experiment.dnn_optimizer):
        optimizer.count_on_device()        # both runs through the device-counted Adam entry point (a captured update needs it)
    for writer in (experiment.dnn_summary_writer, experiment.gan_summary_writer):
        writer.summary_period, writer.steps_to_run = summary_period, 10 ** 9
    seed_all(5)                                                # the host streams the draws come from
    generator = torch.Generator().manual_seed(11)
    losses = []
    for step in range(1, iterations + 1):
        x, labels, u = crowd_inputs(generator, batch, size)
        experiment.training_iteration(x.cuda(), tuple(t.cuda() for t in labels), u.cuda(), step)
        losses.append({name: float(value.item()) for name, value in experiment.last_losses.items() if value is not None})
    torch.cuda.synchronize()
    return experiment, losses


def _compare(eager, replayed, iterations):
    """Every sum that several workgroups share finishes in a fixed order (csrc/split_finish.h, round 5) and both runs use one
    stream, so a replay executes exactly the eager run's arithmetic: weights and both Adam moments of the three networks
    are BIT-identical after any number of iterations.  A replay that dropped one layer's gradient contribution, consumed
    the wrong batch / draw / update count or re-ordered an accumulation cannot pass."""
    for name in ('D', 'DNN', 'G'):
        a, b = getattr(eager, name)._srgan_arena.data, getattr(replayed, name)._srgan_arena.data
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    for a, b in ((eager.d_optimizer, replayed.d_optimizer), (eager.g_optimizer, replayed.g_optimizer),
                 (eager.dnn_optimizer, replayed.dnn_optimizer)):
        assert a.step_count == b.step_count
        assert int(b.device_state[0]) == b.step_count
        assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)


def test_replayed_iterations_match_the_eager_tape():
    iterations = 5
    eager, eager_losses = _run(False, iterations)
    replayed, replayed_losses = _run(True, iterations)
    captured = replayed._captured_iteration
    assert captured.eager_iterations == 1 and captured.replays == iterations - 1 and len(captured.records) == 1
    assert getattr(eager, '_captured_iteration', None) is None
    for step, (a, b) in enumerate(zip(eager_losses, replayed_losses)):
        assert a == b, f'step {step}: {a} vs {b}'                # all six losses of every iteration, bit for bit
    assert eager_losses[-1]['gradient_penalty'] > 0.0 and all(np.isfinite(v) for v in eager_losses[-1].values())
    assert eager_losses[-1] != eager_losses[-2]                 # the replays really consumed new batches and draws
    _compare(eager, replayed, iterations)


def test_summary_steps_run_eagerly_between_replays():
    """A summary step reads losses back on the host: it runs through the eager tape, with the same Adam counter (kept on
    the device) and the same host random streams as the replays around it."""
    eager, _ = _run(False, 6, summary_period=3)
    replayed, _ = _run(True, 6, summary_period=3)
    captured = replayed._captured_iteration
    assert captured.eager_iterations == 3 and captured.replays == 3      # steps 1 (warm-up), 3 and 6 are eager
    _compare(eager, replayed, 6)


def test_counted_adam_matches_the_host_counted_update():
    from srgan_amd.optim import Adam

    class Arena:
        pass
    arenas = []
    for _ in range(2):
        arena = Arena()
        generator = torch.Generator().manual_seed(3)
        arena.data = torch.randn(10007, generator=generator).cuda()
        arena.grad = torch.randn(10007, generator=generator).cuda()
        arena.numel, arena.parameters = 10007, []
        arenas.append(arena)
    host, device = Adam(arenas[0], lr=1e-3, weight_decay=1e-2), Adam(arenas[1], lr=1e-3, weight_decay=1e-2).count_on_device()
    for _ in range(7):
        host.step()
        device.step()
    assert device.step_count == 7 and int(device.device_state[0]) == 7
    torch.testing.assert_close(arenas[0].data, arenas[1].data, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(host.exp_avg_sq, device.exp_avg_sq, rtol=0, atol=0)
