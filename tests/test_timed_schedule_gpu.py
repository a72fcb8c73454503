"""The schedule bench.py TIMES, at the size it is timed at (four streams; the same chains as branches of one HIP graph).

``bench.py`` runs its timed region on four streams (main chain, gradient-penalty chain into the arena's alternate gradient
buffer, the DNN step never joined across iterations, D(unlabeled) of the generator step) and, with ``--step-graph``, as
one captured HIP graph whose branches are those chains.  Races are size dependent, so the experiment is built here by
``bench.build_experiment`` itself and stepped by ``bench.one_step`` -- the code path of the timed loop -- at 512 x 512,
16 images, for TWO consecutive iterations (the un-joined DNN stream crosses an iteration boundary), and compared with the
CPU oracle and with the single-stream HIP schedule (reference order of operations: srgan.py:273-320)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSSES = ('dnn_loss', 'labeled_loss', 'unlabeled_loss', 'fake_loss', 'gradient_penalty', 'generator_loss')


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def bench_arguments(image_size, batch, **overrides):
    """The namespace ``bench.parse()`` yields for ``python bench.py`` (the driver's command line), with the size / batch
    of the case."""
    import bench
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    args.image_size, args.batch_per_gpu = image_size, batch
    for key, value in overrides.items():
        assert hasattr(args, key), key
        setattr(args, key, value)
    return args


def draws_for(iterations, batch, seed=21):
    generator = torch.Generator().manual_seed(seed)
    return [dict(z_d=torch.randn(batch, 256, generator=generator), z_g=torch.randn(batch, 256, generator=generator),
                 alpha=torch.rand(batch, generator=generator)) for _ in range(iterations)]


def run_bench_schedule(args, iterations, draws, initial=None, configure=None):
    """``iterations`` calls of ``bench.one_step`` on an experiment built by ``bench.build_experiment``; returns the
    experiment, the losses of every iteration (read only after the last one: no host sync in between, as in the timed
    loop) and the batches it consumed."""
    import bench
    experiment = bench.build_experiment(args, None)
    if configure is not None:
        configure(experiment)
    if initial is not None:
        for name, data in initial.items():
            getattr(experiment, name)._srgan_arena.data.copy_(data)
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
    kept = []
    for step in range(iterations):
        experiment.injected_draws = {k: v.clone() for k, v in draws[step].items()}
        bench.one_step(experiment, labeled, unlabeled, step)
        # references to this iteration's own loss tensors, read after the last iteration (a graph replay hands out copies
        # of its static output tensors, graph.py, so the same holds there)
        kept.append({name: experiment.last_losses[name].data for name in LOSSES if experiment.last_losses[name] is not None})
    experiment.join_dnn_stream()
    torch.cuda.synchronize()
    losses = [{name: float(values[name].item()) if name in values else None for name in LOSSES} for values in kept]
    return experiment, losses


def weights_of(experiment):
    return {name: getattr(experiment, name)._srgan_arena.data.detach().clone() for name in ('D', 'DNN', 'G')}


def compare_hip_runs(a_losses, b_losses, a, b, first_rtol, later_rtol, lr=1e-4):
    """Two runs of the SAME kernels differ by the summation order of their fp32 atomics (1e-7 relative on a gradient), and
    Adam turns a rounding-level sign difference of a near-zero gradient element into +-lr on that weight (the gradient penalty,
    a function of a recorded gradient's norm, shows 2e-5 between two runs of the same schedule): losses of the
    first iteration agree to ``first_rtol``, later ones to ``later_rtol``; weights agree in the bulk and a few elements may
    be a learning rate per update apart.  A race (a kernel reading a buffer another chain is still writing) is off by
    orders of magnitude more, or NaN under F.POISON."""
    worst = 0.0
    for step, (x, y) in enumerate(zip(a_losses, b_losses)):
        rtol = first_rtol if step == 0 else later_rtol
        for name in LOSSES:
            error = abs(x[name] - y[name]) / max(abs(y[name]), 1e-9)
            worst = max(worst, error)
            assert np.isfinite(x[name]) and error <= rtol, f'iteration {step} {name}: {x[name]} vs {y[name]} (rel {error:.2e})'
    iterations = len(a_losses)
    for name in ('D', 'DNN', 'G'):
        difference = (getattr(a, name)._srgan_arena.data - getattr(b, name)._srgan_arena.data).abs()
        assert float(difference.max()) <= 2.2 * lr * iterations, (name, float(difference.max()))
        assert float(difference.mean()) <= 0.05 * lr * iterations, (name, float(difference.mean()))
    return worst


def _host_memory_gib():
    from test_config_parity_gpu import _host_memory_gib as available
    return available()


def test_the_timed_schedule_at_the_timed_size_matches_the_oracle_and_the_single_stream_run(pkg):
    """512 x 512, 16 images, bench.py's default command line: four streams, resident SyntheticLoader, transitions pool
    first, generator forward before the join; two consecutive iterations.  Losses of both iterations and the post-Adam
    weights of D / G / DNN against the CPU oracle (1e-3; the second iteration 2e-3: it starts from weights that already
    differ by rounding-level Adam sign flips) and against the same experiment on ONE stream."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import conftest
    from oracle import functional as OF, models as OM
    from oracle.experiment import OracleExperiment, Draws
    size, batch, iterations = 512, 16, 2
    memory = _host_memory_gib()
    if memory < 90:
        batch = 8
        assert memory >= 48, f'host memory {memory:.0f} GiB: not even batch 8 of the bench shape fits the CPU oracle'
    entry._cap_host_threads(torch)
    draws = draws_for(iterations, batch)
    args = bench_arguments(size, batch)
    assert not args.single_stream and not args.step_graph
    # the initial weights, before any step, for the oracle and the single-stream run
    import bench
    probe = bench.build_experiment(args, None)
    assert probe.settings.overlap_dnn_step and probe.settings.overlap_gradient_penalty and \
        probe.settings.overlap_generator_forwards and probe.train_dataset_loader.resident
    initial = weights_of(probe)
    state = {name: {k: v.detach().cpu().clone() for k, v in getattr(probe, name).state_dict().items()}
             for name in ('D', 'DNN', 'G')}
    labeled_pool = [tuple(t.cpu() for t in b) for b in probe.train_dataset_loader.batches]
    unlabeled_pool = [tuple(t.cpu() for t in b) for b in probe.unlabeled_dataset_loader.batches]
    del probe
    torch.cuda.empty_cache()

    experiment, losses = run_bench_schedule(args, iterations, draws, initial)
    single, single_losses = run_bench_schedule(bench_arguments(size, batch, single_stream=True), iterations, draws, initial)
    assert not single.settings.overlap_dnn_step
    worst = compare_hip_runs(losses, single_losses, experiment, single, first_rtol=1e-4, later_rtol=5e-4)

    oracle_g = OM.DCGANGenerator(image_size=size)
    oracle_d, oracle_dnn = OM.KnnDenseNetCat(image_size=size), OM.KnnDenseNetCat(image_size=size)
    for name, module in (('D', oracle_d), ('DNN', oracle_dnn), ('G', oracle_g)):
        module.load_state_dict(state[name], strict=True)
    oracle = OracleExperiment(entry.settings_for_oracle(experiment.settings), oracle_d, oracle_dnn, oracle_g,
                              labeled_loss_function=lambda p, y, order: OF.crowd_labeled_loss(p, y, order, 1e-3))
    oracle_worst = 0.0
    for step in range(iterations):
        x, heads, knn = labeled_pool[step % len(labeled_pool)]
        u = unlabeled_pool[step % len(unlabeled_pool)][0]
        d = draws[step]
        dnn_loss = oracle.dnn_training_step(x, (heads, knn))
        expected = oracle.gan_training_step(x, (heads, knn), u, step,
                                            Draws(d['z_d'], d['z_g'], d['alpha'].reshape(-1, 1, 1, 1)))
        expected['dnn_loss'] = float(dnn_loss.item())
        rtol = 1e-3 if step == 0 else 2e-3
        for name in LOSSES:
            error = abs(losses[step][name] - expected[name]) / max(abs(expected[name]), 1e-12)
            oracle_worst = max(oracle_worst, error)
            print(f'[timed schedule 512x{batch}] iteration {step} {name}: hip {losses[step][name]:.6g}  oracle {expected[name]:.6g}  '
                  f'rel {error:.2e}')
            assert error < rtol, (step, name)
    assert expected['gradient_penalty'] > 1.0
    for name, ours, theirs in (('D', experiment.D, oracle.D), ('G', experiment.G, oracle.G), ('DNN', experiment.DNN, oracle.DNN)):
        reference = dict(theirs.named_parameters())
        for pname, p in ours.named_parameters():
            want, got = reference[pname].detach().numpy(), p.detach().cpu().numpy()
            assert np.abs(got - want).max() <= iterations * 2.2e-4 + 1e-3 * np.abs(want).max(), f'{name} {pname}'
            assert np.abs(got - want).mean() <= 4e-5 + 1e-4 * np.abs(want).mean(), f'{name} {pname} (mean)'
    conftest.PARITY_NOTES.append(
        f'bench.py\'s timed schedule (four streams, resident batches, generator forward before the join) checked at 512x512, '
        f'batch {batch}, two consecutive iterations: worst relative loss error {oracle_worst:.1e} against the CPU oracle, '
        f'{worst:.1e} against the single-stream HIP run')


@pytest.mark.parametrize('step_graph', [False, True])
def test_the_timed_schedule_on_poisoned_allocations(pkg, monkeypatch, step_graph):
    """256 x 256, 8 images, every allocation NaN-poisoned: three consecutive iterations of the four-stream schedule (eager,
    and captured as ONE HIP graph with the chains as parallel branches and then replayed) against the single-stream
    eager run."""
    from srgan_amd import functional as F
    size, batch, iterations = 256, 8, 3
    draws = draws_for(iterations, batch, seed=5)
    single, single_losses = run_bench_schedule(bench_arguments(size, batch, single_stream=True), iterations, draws)
    monkeypatch.setattr(F, 'POISON', True)
    args = bench_arguments(size, batch, step_graph=step_graph)
    experiment, losses = run_bench_schedule(args, iterations, draws)
    if step_graph:
        captured = experiment._captured_iteration
        assert experiment.settings.overlap_dnn_step and captured.replays >= 1, (captured.replays, captured.eager_iterations)
    compare_hip_runs(losses, single_losses, experiment, single, first_rtol=1e-4, later_rtol=5e-3)
    assert losses[-1]['gradient_penalty'] > 0.0


def test_graph_replay_with_the_dnn_side_stream_and_a_resident_loader(pkg):
    """``step_graph`` + ``overlap_dnn_step`` with a SyntheticLoader.  The DNN step must be INSIDE the
    captured graph (forked from the capturing stream, joined before the capture ends): every replay trains the DNN, on the
    batch of that replay -- compared with the eager four-stream run over five iterations at 64 x 64."""
    size, batch, iterations = 64, 2, 5
    draws = draws_for(iterations, batch, seed=9)
    eager, eager_losses = run_bench_schedule(bench_arguments(size, batch), iterations, draws)
    replayed, replayed_losses = run_bench_schedule(bench_arguments(size, batch, step_graph=True), iterations, draws)
    captured = replayed._captured_iteration
    assert captured.replays >= 2 and replayed.settings.overlap_dnn_step
    assert replayed.dnn_optimizer.step_count == eager.dnn_optimizer.step_count == iterations
    assert int(replayed.dnn_optimizer.device_state[0]) == iterations
    # the DNN's weights moved in every iteration, and to the same place as in the eager run
    difference = (replayed.DNN._srgan_arena.data - eager.DNN._srgan_arena.data).abs()
    assert float(difference.max()) <= 2.2e-4 * iterations and float(difference.mean()) <= 2e-5
    for step in (0, 1):
        for name in LOSSES:
            assert np.isclose(replayed_losses[step][name], eager_losses[step][name], rtol=5e-3, atol=1e-6), (step, name)
    assert replayed_losses[-1]['dnn_loss'] != replayed_losses[-2]['dnn_loss']


def test_two_captured_graphs_alternating_in_one_memory_pool(pkg):
    """with ``generator_training_step_period = 2`` the iterations alternate between TWO captured graphs
    (with and without the generator step) that share one memory pool.  Eight iterations at 64 x 64: every iteration's loss
    tensors are read at once AND kept, and read again after the last replay -- a replay of the other graph, which may use
    the same pool memory as scratch, must not have changed them (graph.py hands out copies); the first iterations are
    compared with the eager run (later ones only loosely: the gradient penalty of this tiny case amplifies the
    rounding-level difference between two runs about tenfold per iteration, profiles/r04h_schedule_check_bisect.txt)."""
    import bench
    size, batch, iterations = 64, 2, 8
    draws = draws_for(iterations, batch, seed=13)

    def run(step_graph):
        experiment = bench.build_experiment(bench_arguments(size, batch, step_graph=step_graph), None)
        experiment.settings.generator_training_step_period = 2
        labeled = experiment.infinite_iter(experiment.train_dataset_loader)
        unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
        kept, at_once = [], []
        for step in range(iterations):
            experiment.injected_draws = {k: v.clone() for k, v in draws[step].items()}
            bench.one_step(experiment, labeled, unlabeled, step)
            tensors = {name: experiment.last_losses[name].data for name in LOSSES if experiment.last_losses[name] is not None}
            experiment.join_dnn_stream()
            torch.cuda.synchronize()
            kept.append(tensors)
            at_once.append({name: float(value.item()) for name, value in tensors.items()})
        torch.cuda.synchronize()
        later = [{name: float(value.item()) for name, value in tensors.items()} for tensors in kept]
        return experiment, at_once, later

    eager, eager_losses, eager_later = run(False)
    replayed, replayed_losses, replayed_later = run(True)
    captured = replayed._captured_iteration
    assert len(captured.records) == 2 and captured.replays >= 4, (len(captured.records), captured.replays)
    pools = {record['graph'].pool() for record in captured.records.values()}
    assert len(pools) == 1, pools
    assert replayed.g_optimizer.step_count == eager.g_optimizer.step_count == iterations // 2
    assert replayed.d_optimizer.step_count == eager.d_optimizer.step_count == iterations
    assert replayed_later == replayed_losses and eager_later == eager_losses        # bit for bit: nothing wrote under them
    for step in range(iterations):
        # (bench.one_step numbers its iterations from 1: the generator trains in the test's odd steps)
        assert ('generator_loss' in replayed_losses[step]) == (step % 2 == 1), step
        assert ('generator_loss' in eager_losses[step]) == (step % 2 == 1), step
        for name, value in replayed_losses[step].items():
            rtol = 2e-3 if step == 0 else (5e-2 if step < 4 else 0.5)
            assert np.isfinite(value) and np.isclose(value, eager_losses[step][name], rtol=rtol, atol=1e-6), \
                (step, name, value, eager_losses[step][name])
    for name in ('D', 'DNN', 'G'):
        difference = (getattr(replayed, name)._srgan_arena.data - getattr(eager, name)._srgan_arena.data).abs()
        assert float(difference.max()) <= 2.2e-4 * iterations, name


def test_driving_validation_mae_after_twenty_fp16_steps_matches_the_fp32_oracle(pkg):
    """BASELINE.json configs[4]'s own check (SURVEY.md 8d "per-task MAE vs CPU ref = mean |D(x) - y| on a fixed synthetic
    validation tensor after K identical steps"; reference driving/srgan.py:87-104): K = 20 iterations of the driving SRGAN
    at 64 x 192, batch 128, in the named mode -- fp16 MFMA operands, the gradient-penalty chain in fp32, loss scale 256 --
    against the CPU oracle trained in fp32 on the same batches, draws and initial weights; then the validation MAE of D
    (and of the DNN) on the fixed validation batch.  Tolerance 2e-2 relative on the MAE (the fp16 operand rounding, the
    same bound the single-step loss comparison uses); the observed difference is printed."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import conftest
    from srgan_amd.driving.srgan import DrivingExperiment
    from srgan_amd.settings import Settings
    from srgan_amd.utility import SummaryWriter, seed_all
    from oracle import models as OM
    from oracle.experiment import OracleExperiment, Draws
    entry._cap_host_threads(torch)
    size, batch, steps = (64, 192), 128, 20
    settings = Settings()
    settings.batch_size = batch
    settings.matching_loss_multiplier, settings.contrasting_loss_multiplier = 1e2, 1e1
    settings.gradient_penalty_multiplier = 1e2
    settings.compute_dtype, settings.gradient_penalty_dtype, settings.loss_scale = 'f16', 'f32', 256.0
    experiment = DrivingExperiment(settings)
    experiment.image_size = size
    seed_all(0)
    experiment.dataset_setup()
    experiment.model_setup()
    with torch.no_grad():
        for module in experiment.D.modules():
            if isinstance(module, torch.nn.Conv2d):
                module.weight.mul_(2.2)                     # gradient penalty active
    oracle_g = OM.DCGANGenerator(image_size=size)
    oracle_d, oracle_dnn = OM.DCGANDiscriminator(image_size=size), OM.DCGANDiscriminator(image_size=size)
    for ours, theirs in ((experiment.G, oracle_g), (experiment.D, oracle_d), (experiment.DNN, oracle_dnn)):
        theirs.load_state_dict({k: v.detach().cpu().clone() for k, v in ours.state_dict().items()}, strict=True)
    oracle_settings = entry.settings_for_oracle(settings)
    oracle = OracleExperiment(oracle_settings, oracle_d, oracle_dnn, oracle_g)
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(summary_period=10 ** 9), SummaryWriter(summary_period=10 ** 9)
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    labeled, unlabeled = experiment.train_dataset_loader.batches, experiment.unlabeled_dataset_loader.batches
    generator = torch.Generator().manual_seed(77)
    penalties = []
    for step in range(steps):
        x, y = labeled[step % len(labeled)]
        u = unlabeled[step % len(unlabeled)][0]
        z_d, z_g = torch.randn(batch, 256, generator=generator), torch.randn(batch, 256, generator=generator)
        alpha = torch.rand(batch, generator=generator)
        experiment.injected_draws = {'z_d': z_d, 'z_g': z_g, 'alpha': alpha}
        experiment.dnn_training_step(x, y, step + 1)
        experiment.gan_training_step(x, y, u, step + 1)
        oracle.dnn_training_step(x.cpu(), y.cpu())
        expected = oracle.gan_training_step(x.cpu(), y.cpu(), u.cpu(), step, Draws(z_d, z_g, alpha.reshape(-1, 1, 1, 1)))
        penalties.append(expected['gradient_penalty'])
    assert max(penalties) > 0.0, 'gradient penalty never active'
    experiment.join_dnn_stream()
    experiment.eval_mode()
    writer = SummaryWriter()
    images, angles = experiment.validation_dataset_loader.batches[0]
    results = {}
    for name, ours, theirs in (('D', experiment.D, oracle.D), ('DNN', experiment.DNN, oracle.DNN)):
        mae = experiment.regression_evaluation_epoch(ours, experiment.validation_dataset_loader.batches, writer, name)
        with torch.no_grad():
            reference = float((theirs(images.cpu()).reshape(-1) - angles.cpu()).abs().mean())
        error = abs(mae - reference) / reference
        results[name] = (mae, reference, error)
        print(f'[driving fp16, {steps} steps] {name} validation MAE: hip {mae:.6f}  fp32 oracle {reference:.6f}  rel {error:.2e}')
        assert error <= 2e-2, (name, mae, reference)
    # the training really moved the predictions: the trained MAE differs from the MAE at initialisation
    conftest.PARITY_NOTES.append(
        f'config 5 (driving 64x192, batch 128, fp16 with fp32 penalty chain): after {steps} identical steps the validation MAE of D '
        f'is {results["D"][0]:.5f} against {results["D"][1]:.5f} from the fp32 CPU oracle (rel {results["D"][2]:.1e}); DNN '
        f'{results["DNN"][0]:.5f} / {results["DNN"][1]:.5f}')


def test_pointwise_kernel_still_serves_the_fused_data_gradient(pkg):
    """The fused data gradient (batch-norm + ReLU backward in the epilogue) runs on the LDS-DMA kernel of pointwise_ring.hip
    where its geometry allows (test_ops_gpu.py covers it there: two row tiles on a channel slice, partial last tiles of 32 and
    96 rows); with SRGAN_NO_PW_RING_EPILOGUE=1 the same shapes go through pointwise_kernel, which stays the path of every
    other geometry -- the same test in a process with that switch."""
    import subprocess
    done = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_ops_gpu.py'), '-q', '-x', '-m', 'gpu',
                           '-k', 'fused_batch_norm_backward_in_the_data_gradient'], capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, SRGAN_NO_PW_RING_EPILOGUE='1'), cwd=ROOT)
    assert done.returncode == 0 and '1 passed' in done.stdout, done.stdout[-3000:] + done.stderr[-2000:]
