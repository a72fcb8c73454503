"""The grouped 1x1 weight gradients of a dense block (one launch for up to 48 problems) through the Python host and through the
raw C ABI (srgan_wgrad_group_plan / srgan_wgrad_group_run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import srgan_amd
    assert torch.cuda.is_available()
    return srgan_amd


def test_grouped_1x1_weight_gradients_mixing_both_tile_forms(pkg):
    """A grouped 1x1 weight-gradient launch whose problems were planned for BOTH kernels (`srgan_wgrad_group_plan` returns
    the variant per slot, the launch gets their OR: the LDS-staged 128 x 128 tiles for the wide layers, the register-streamed
    64 x 64 tiles for the narrow ones; reference crowd/models.py:340-341 through loss.backward()): the dense-block test with
    the threshold between the block's layers (32 / 40 / 48 input channels), in a process of its own because the threshold
    is read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    environment = dict(os.environ, SRGAN_PWL_MIN_CI='40')
    command = [sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_ops_gpu.py'), '-q', '-x', '-m', 'gpu', '-k',
               'test_fused_dense_block_with_in_kernel_batch_norm and plane0']
    done = subprocess.run(command, capture_output=True, text=True, timeout=900, env=environment, cwd=root)
    assert done.returncode == 0 and '1 passed' in done.stdout, done.stdout[-2000:] + done.stderr[-2000:]


@pytest.mark.parametrize('plane,shares', [((32, 32), True), ((32, 32), False), ((16, 8), True), ((14, 14), True)])
def test_grouped_1x1_weight_gradients_through_the_abi(pkg, plane, shares):
    """`srgan_wgrad_group_plan` / `srgan_wgrad_group_run` called the way the reference's maintainer would: four
    norm -> relu -> 1x1 convolutions that read growing channel prefixes of ONE buffer (reference crowd/models.py:335-353 behind
    loss.backward()) as one launch -- widths that need one, two, three and four 128-column tiles with uneven column blocks
    (96 / 160 / 288 / 416 input channels), the K range split over many workgroups and finished in slice order.  Whole planes take
    the LDS-staged kernel, 14 x 14 the register-streamed one; with and without the group's weights (shares by work / equal
    shares).  Against torch in float64; twice, with NaN in the workspace in between: the same bits."""
    import ctypes
    from srgan_amd import _lib
    lib, stream = _lib.library(), _lib.stream_handle()
    assert lib.srgan_split_is_ordered(stream) == 1
    h, w = plane
    hw, n, width, cins = h * w, 3, 128, (96, 160, 288, 416)
    total = max(cins)
    generator = torch.Generator().manual_seed(29)
    buffer = torch.randn(n, total, h, w, generator=generator)
    gy = torch.randn(len(cins), n, width, h, w, generator=generator)
    norms = [dict(mean=torch.randn(c, generator=generator) * 0.3, var=torch.rand(c, generator=generator) + 0.5,
                  gamma=torch.rand(c, generator=generator) + 0.5, beta=torch.randn(c, generator=generator) * 0.3) for c in cins]
    want = []
    for index, c in enumerate(cins):
        p = norms[index]
        act = torch.nn.functional.batch_norm(buffer[:, :c].double(), p['mean'].double(), p['var'].double(), p['gamma'].double(),
                                             p['beta'].double(), training=False, eps=1e-5).relu()
        want.append(torch.nn.grad.conv2d_weight(act, (width, c, 1, 1), gy[index].double()))
    device = torch.device('cuda', 0)
    d_buffer, d_gy = buffer.to(device), gy.to(device)
    keep, slots = [], (ctypes.c_byte * (128 * len(cins)))()
    gws = [torch.full((width, c, 1, 1), 0.25, device=device) for c in cins]
    grid_x = grid_y = variants = 0
    partial_at = taps = elements = 0
    weights = sum(width * c for c in cins) if shares else 0
    for index, c in enumerate(cins):
        p = norms[index]
        vectors = [t.to(device) for t in (p['mean'], (p['var'] + 1e-5).rsqrt(), p['gamma'], p['beta'])]
        keep.append(vectors)
        bn = _lib.BnRelu(*[t.data_ptr() for t in vectors])
        desc = _lib.ConvDesc(n, c, h, w, width, 1, 1, 1, 1, 0, 0, h, w, total * hw, 0, 0)
        gx, gyy, variant, partial = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _lib.check(lib.srgan_wgrad_group_plan(desc, bn, 0, index * n * width * hw, gws[index].data_ptr(), 0, len(cins), weights,
                                              partial_at, ctypes.byref(slots, 128 * index), ctypes.byref(gx), ctypes.byref(gyy),
                                              ctypes.byref(variant), ctypes.byref(partial)), 'srgan_wgrad_group_plan')
        partial_at += partial.value
        grid_x, grid_y, variants = max(grid_x, gx.value), max(grid_y, gyy.value), variants | variant.value
        taps += width * c
        elements += (c + width) * n * hw
    assert variants == (1 if hw % 32 == 0 else 4), variants        # the staged form on whole planes, the ragged variant on 14 x 14
    assert grid_y > 1 and partial_at > 0                            # the K range IS split: the ordered finish runs
    table = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(device)
    workspace = _lib._workspaces[(torch.cuda.current_device(), stream)]
    results = []
    for run in range(2):
        for gw in gws:
            gw.fill_(0.25)
        workspace.fill_(float('nan'))
        _lib.check(lib.srgan_wgrad_group_run(table.data_ptr(), len(cins), 1, grid_x, grid_y, variants, 1, d_buffer.data_ptr(),
                                             d_gy.data_ptr(), None, taps, n * hw, elements, partial_at, stream), 'srgan_wgrad_group_run')
        torch.cuda.synchronize()
        results.append([gw.clone() for gw in gws])
    for index, c in enumerate(cins):
        got, expected = results[0][index].cpu().double() - 0.25, want[index]
        scale = float(expected.abs().max())
        assert float((got - expected).abs().max()) <= 3e-5 * scale, f'{c} input channels on {h} x {w}'
        assert torch.equal(results[0][index], results[1][index]), f'{c} input channels: two runs differ'
