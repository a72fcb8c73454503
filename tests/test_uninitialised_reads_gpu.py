"""No kernel may depend on memory nobody wrote: with ``functional.POISON`` every tensor the step allocates starts as NaN
(instead of whatever the caching allocator recycles -- usually a plausible finite leftover, which is how a fill
implemented as ``0 * y + value`` went unnoticed until a HIP-graph replay met NaN leftovers).  The training steps must
still reproduce the reference goldens."""
import pytest

import test_steps_gpu as steps

pytestmark = pytest.mark.gpu


@pytest.fixture
def poisoned(monkeypatch):
    import srgan_amd  # noqa: F401
    from srgan_amd import functional as F
    monkeypatch.setattr(F, 'POISON', True)
    return srgan_amd


@pytest.mark.parametrize('name,size,count,reference_schedule', [
    ('g7c_crowd64_gp_active', 64, 1, False), ('g7c_crowd64_gp_active', 64, 1, True), ('g7b_crowd64', 64, 2, False),
    ('g7_crowd224', 224, 1, False)])
def test_crowd_steps_on_poisoned_allocations(poisoned, name, size, count, reference_schedule):
    steps.test_crowd_steps(poisoned, name, size, count, reference_schedule)


@pytest.mark.parametrize('reference_schedule', [False, True])
def test_dcgan_step_on_poisoned_allocations(poisoned, reference_schedule):
    steps.test_tiny_dcgan_with_active_gradient_penalty(poisoned, reference_schedule)


def test_coefficient_steps_on_poisoned_allocations(poisoned):
    steps.test_coefficient_srgan(poisoned, 'g3b_coefficient_srgan_gp_active', 2, False)
    steps.test_coefficient_sgan(poisoned, 'g4c_coefficient_sgan_gp_active')


def test_fill_does_not_read_its_destination(poisoned):
    import torch
    from srgan_amd import functional as F
    for count in (1, 3, 1000, 100003):
        target = torch.full((count,), float('nan'), device='cuda')
        F.fill_(target, 2.5)
        assert bool((target == 2.5).all())
        F.fill_(target, 0.0)
        assert bool((target == 0.0).all())
