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
    """Two runs of the SAME eager code already differ by fp32-atomic summation order, and Adam's first updates turn a
    rounding-level gradient difference into a +-lr step of that element (cf. test_steps_gpu): the bulk must agree, a
    few elements may be a learning rate per update apart.  A replay that consumed the wrong batch, draw or update count
    is off by orders of magnitude more."""
    lr = 1e-4
    for name in ('D', 'DNN', 'G'):
        a, b = getattr(eager, name)._srgan_arena.data, getattr(replayed, name)._srgan_arena.data
        difference = (a - b).abs()
        assert float(difference.max()) <= 2.2 * lr * iterations, name
        assert float(difference.mean()) <= 0.5 * lr, name
    for a, b in ((eager.d_optimizer, replayed.d_optimizer), (eager.g_optimizer, replayed.g_optimizer),
                 (eager.dnn_optimizer, replayed.dnn_optimizer)):
        assert a.step_count == b.step_count
        assert int(b.device_state[0]) == b.step_count
        assert bool(torch.isfinite(b.exp_avg).all()) and bool(torch.isfinite(b.exp_avg_sq).all())
        scale = float(a.exp_avg.abs().max())
        assert float((a.exp_avg - b.exp_avg).abs().max()) <= 0.25 * scale       # (first moments follow the drifting gradients)


def test_replayed_iterations_match_the_eager_tape():
    iterations = 5
    eager, eager_losses = _run(False, iterations)
    replayed, replayed_losses = _run(True, iterations)
    captured = replayed._captured_iteration
    assert captured.eager_iterations == 1 and captured.replays == iterations - 1 and len(captured.records) == 1
    assert getattr(eager, '_captured_iteration', None) is None
    for step, (a, b) in enumerate(zip(eager_losses, replayed_losses)):
        assert a.keys() == b.keys()
        # (the runs drift apart with every Adam update -- see _compare; the gradient penalty amplifies it most)
        rtol = 5e-3 if step <= 1 else 0.2
        for name in a:
            assert np.isclose(a[name], b[name], rtol=rtol, atol=1e-6), f'step {step} {name}: {a[name]} vs {b[name]}'
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
