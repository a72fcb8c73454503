"""Round-5 GPU checks.

* H12 on the device: an UN-INJECTED coefficient training run -- the product's own `seed_all`, set-up, loaders and
  `Experiment.draw_*` -- reproduces golden g3 (reference srgan.py:286-301,364; utility.py:102-116).
* The few-row ordered reduction (the gradient penalty's per-example norms, reference srgan.py:371) under stress: NaN in the
  workspace partials before every launch, a bandwidth-heavy copy in flight on a second stream.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, golden_scalars, assert_close
from test_random_draws_cpu import coefficient_experiment, fetch_batches

pytestmark = pytest.mark.gpu
RTOL = 1e-3
TAGS = {'Generator/Loss': 'generator_loss', 'Discriminator/Labeled Loss': 'labeled_loss',
        'Discriminator/Unlabeled Loss': 'unlabeled_loss', 'Discriminator/Fake Loss': 'fake_loss',
        'Discriminator/Gradient Penalty': 'gradient_penalty', 'Discriminator/Gradient Norm': 'gradient_norm_mean',
        'Feature Norm/Labeled': 'feature_norm_labeled', 'Feature Norm/Unlabeled': 'feature_norm_unlabeled'}


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def test_uninjected_coefficient_run_reproduces_the_reference(pkg):
    """Nothing injected: three iterations of dnn_training_step + gan_training_step whose z_D, alpha and z_G come from the
    product's own draw path, in the reference's order, land on the reference's logged losses (golden g3)."""
    from srgan_amd.utility import SummaryWriter
    g = load_golden('g3_coefficient_srgan')
    experiment = coefficient_experiment(int(g['batch_size']))
    steps = int(g['steps'])
    batches = fetch_batches(experiment, steps)          # (the goldens fetched their batches before the first step too)
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    assert experiment.injected_draws is None
    for step, (x, y, u) in enumerate(batches):
        experiment.dnn_training_step(x.cuda(), y.cuda(), step)
        experiment.gan_training_step(x.cuda(), y.cuda(), u.cuda(), step)
        logged = {TAGS[tag]: values[-1][1] for tag, values in experiment.gan_summary_writer.scalars.items()}
        logged['dnn_loss'] = experiment.dnn_summary_writer.scalars['Discriminator/Labeled Loss'][-1][1]
        for key, value in golden_scalars(g, step).items():
            if key in logged:
                assert_close(logged[key], value, rtol=RTOL, atol=1e-6 if abs(value) < 1e-3 else 0.0,
                             what=f'un-injected step {step} {key}')
        assert_close(experiment.gradient_norm.cpu().numpy(), g[f's{step}/gradient_norm'], rtol=RTOL, atol=1e-6,
                     what=f'un-injected step {step} gradient_norm')


def test_ordered_row_reduction_with_poisoned_partials_next_to_a_heavy_stream(pkg):
    """`chan_reduce_rows_ordered_kernel`: every workgroup's partial must be visible to the row's last workgroup.  The
    workspace is filled with NaN in front of every launch (a partial read before it was written shows), a 1 GiB copy
    loop keeps HBM busy on a second stream, 300 launches over three row shapes: every result equals the first, bit for bit,
    and equals torch's float64 sum."""
    from srgan_amd import functional as F, _lib
    generator = torch.Generator().manual_seed(5)
    source = torch.empty(1 << 28, dtype=torch.float32, device='cuda').normal_()
    sink = torch.empty_like(source)
    side = torch.cuda.Stream()
    for rows, length in [(16, 3 * 512 * 512), (2, 3 * 64 * 64), (16, 3 * 224 * 224)]:
        x = torch.randn(rows, length, generator=generator)
        xv = F.leaf(x.cuda())
        want = (x.double() * x.double()).sum(1)
        handle = _lib.stream_handle()
        workspace = _lib._workspaces[(torch.cuda.current_device(), handle)]
        first = None
        results = []
        for iteration in range(100):
            if iteration % 10 == 0:
                with torch.cuda.stream(side):
                    sink.copy_(source)
            workspace.fill_(float('nan'))
            results.append(F.row_dot(xv, xv).data.clone())
        torch.cuda.synchronize()
        first = results[0]
        assert_close(first.cpu().numpy(), want.float().numpy(), rtol=2e-6, what=f'squared norms {rows} x {length}')
        for iteration, result in enumerate(results):
            assert torch.equal(result, first), f'{rows} x {length}: launch {iteration} differs from the first ({result} vs {first})'


def test_rccl_entry_points_of_the_abi_on_one_rank(pkg):
    """include/srgan_hip.h "collectives": a communicator of ONE rank from a unique id (two ranks cannot share a device under
    RCCL, one rank can); all-reduce, reduce-scatter and all-gather in fp32 and bf16 on the caller's stream are the identity
    there, and they must leave exactly that; argument errors come back as -1 before RCCL is reached."""
    import ctypes
    from srgan_amd import _lib
    lib = _lib.library()
    assert lib.srgan_comm_available() == 1
    identifier = ctypes.create_string_buffer(128)
    _lib.check(lib.srgan_comm_unique_id(identifier), 'srgan_comm_unique_id')
    assert any(identifier.raw)
    comm = ctypes.c_void_p()
    _lib.check(lib.srgan_comm_init(ctypes.byref(comm), 1, 0, identifier.raw), 'srgan_comm_init')
    world = ctypes.c_int32()
    _lib.check(lib.srgan_comm_world_size(comm, ctypes.byref(world)), 'srgan_comm_world_size')
    assert world.value == 1
    stream = torch.cuda.current_stream().cuda_stream
    for dtype, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        source = torch.randn(100003, device='cuda').to(dtype)
        out = torch.zeros_like(source)
        _lib.check(lib.srgan_all_reduce_sum(comm, source.data_ptr(), out.data_ptr(), source.numel(), code, stream), 'all_reduce')
        assert torch.equal(out, source)
        in_place = source.clone()
        _lib.check(lib.srgan_all_reduce_sum(comm, in_place.data_ptr(), in_place.data_ptr(), source.numel(), code, stream), 'all_reduce')
        assert torch.equal(in_place, source)
        shard = torch.zeros_like(source)
        _lib.check(lib.srgan_reduce_scatter_sum(comm, source.data_ptr(), shard.data_ptr(), source.numel(), code, stream), 'reduce_scatter')
        gathered = torch.zeros_like(source)
        _lib.check(lib.srgan_all_gather(comm, shard.data_ptr(), gathered.data_ptr(), source.numel(), code, stream), 'all_gather')
        assert torch.equal(gathered, source)
    assert lib.srgan_all_reduce_sum(comm, source.data_ptr(), out.data_ptr(), 4, 7, stream) == _lib.EINVAL     # unknown dtype
    assert lib.srgan_all_reduce_sum(None, source.data_ptr(), out.data_ptr(), 4, 0, stream) == _lib.EINVAL
    torch.cuda.synchronize()
    _lib.check(lib.srgan_comm_destroy(comm), 'srgan_comm_destroy')


@pytest.mark.parametrize('streams', [False, True])
def test_data_parallel_step_through_the_abi_collectives_equals_the_plain_step(streams):
    """`SRGAN_ABI_COLLECTIVES=1`: the feature-sum all-reduce of the forward pass and every asynchronous gradient bucket go
    through srgan_all_reduce_sum on an RCCL communicator the ABI created (world size 1, exchanges forced on; the process
    group only carries the unique id, the broadcasts and the barrier) -- losses and updated weights equal the plain step."""
    import test_parallel_gpu as parallel_tests
    reference_result, reference_tensors = parallel_tests._step(None)
    (rank, result, tensors, launched), = parallel_tests._run_ranks(1, 'nccl', force=True, streams=streams, abi=True)
    assert launched['DNN'] and launched['D'] and launched['G'], launched
    buckets = sum(len(b) for name, runs in launched.items() if name != 'abi_collectives' for b in runs)
    assert launched['abi_collectives'] >= buckets + 3, launched          # the buckets + the forward feature sums
    for key, value in reference_result.items():
        assert abs(result[key] - value) <= 1e-4 * max(abs(value), 1e-6), (key, result[key], value)
    for key, value in reference_tensors.items():
        limit = 2.2e-4 + 1e-3 * float(np.abs(value).max())
        assert float(np.abs(tensors[key] - value).max()) <= limit, key
    import conftest
    conftest.PARITY_NOTES.append(f'data-parallel step with every exchange through the C ABI\'s RCCL entry points (world size 1, forced): '
                                 f'{launched["abi_collectives"]} collectives, {buckets} gradient buckets, losses equal the plain step '
                                 f'(side streams {"on" if streams else "off"})')


def test_age_vgg_fp32_step_at_batch_128_matches_the_oracle(pkg, monkeypatch):
    """BASELINE.json configs[1] at its STATED batch against the ORACLE (the batch-128 steps were only
    compared with the product's own fp32 step): VGG-16 discriminator on 64 x 64 faces, 128 examples, fp32 -- the five
    logged losses within 1e-3 and the post-Adam weights of all three networks against the PyTorch-CPU restatement."""
    import srgan_amd.age.srgan as age
    from oracle import models as OM
    from test_steps_gpu import _hip_and_oracle_step
    monkeypatch.setattr(age, 'model_architecture', 'vgg')

    def configure(experiment):
        experiment.image_size = 64
    _hip_and_oracle_step(age.AgeExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=64), OM.VGG16(1, 64), OM.VGG16(1, 64)),
                         size=64, batch=128, d_scale=1.3)
    import conftest
    conftest.PARITY_NOTES.append('config 2 (age, VGG-16 @ 64 x 64) fp32 step at batch 128 checked against the CPU oracle')


def test_driving_fp32_step_at_batch_128_matches_the_oracle(pkg):
    """BASELINE.json configs[4] at its stated per-device batch against the ORACLE: the DCGAN pair on 64 x 192 frames, 128
    examples, fp32."""
    from srgan_amd.driving.srgan import DrivingExperiment
    from oracle import models as OM
    from test_steps_gpu import _hip_and_oracle_step
    size = (64, 192)

    def configure(experiment):
        experiment.image_size = size
    _hip_and_oracle_step(DrivingExperiment, configure,
                         lambda: (OM.DCGANGenerator(image_size=size), OM.DCGANDiscriminator(image_size=size),
                                  OM.DCGANDiscriminator(image_size=size)),
                         size=size, batch=128, d_scale=2.2)
    import conftest
    conftest.PARITY_NOTES.append('config 5 (driving, DCGAN @ 64 x 192) fp32 step at batch 128 checked against the CPU oracle')
