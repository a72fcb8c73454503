"""Index-math check of the product's convolution plans (sr-gan_amd/csrc/conv_plan.h) on the CPU.

The plans are executed by a plain-loop emulator compiled from tests/csrc (test infrastructure; the GPU
kernels execute the same plans) and compared with torch's conv ops: padding, strides, stride-parity
classes for backward-data, the non-overlapping shortcut, "linear" convolutions, bias and accumulation.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'csrc', 'emul_gather_gemm.cpp')
LIB = os.path.join(ROOT, 'tests', 'csrc', 'libemul_gather_gemm.so')


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'sh', 'sw', 'ph', 'pw', 'OH', 'OW')] + \
               [('x_bs', ctypes.c_int64), ('y_bs', ctypes.c_int64)]


@pytest.fixture(scope='module')
def emul():
    headers = [os.path.join(ROOT, 'sr-gan_amd', 'csrc', h) for h in ('gather_gemm.h', 'conv_plan.h')]
    newest = max(os.path.getmtime(p) for p in headers + [SRC])
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC',
                               '-I' + os.path.join(ROOT, 'sr-gan_amd', 'csrc'), SRC, '-o', LIB])
    return ctypes.CDLL(LIB)


def fptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def geom_of(x, w, stride, pad, y):
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    return ConvGeom(n, c, h, wd, k, r, s, stride[0], stride[1], pad[0], pad[1], y.shape[2], y.shape[3], 0, 0)


CASES = [
    # N, C, H, W, K, R, S, stride, pad
    (2, 3, 9, 8, 5, 3, 3, (1, 1), (1, 1)),      # DenseNet / VGG 3x3
    (2, 4, 7, 7, 6, 1, 1, (1, 1), (0, 0)),      # 1x1 bottleneck
    (2, 3, 12, 10, 4, 4, 4, (2, 2), (1, 1)),    # DCGAN k4 s2 p1
    (1, 3, 15, 13, 4, 7, 7, (2, 2), (3, 3)),    # stem 7x7 s2 p3 (odd sizes: untouched border rows)
    (2, 2, 8, 8, 3, 2, 2, (2, 2), (0, 0)),      # MapModule k2 s2 (non-overlapping)
    (3, 5, 4, 6, 7, 4, 6, (1, 1), (0, 0)),      # "linear" conv: kernel == input
    (1, 2, 9, 9, 2, 3, 3, (2, 2), (0, 0)),      # stride 2, no pad
    (2, 3, 6, 5, 2, 3, 2, (1, 2), (1, 0)),      # anisotropic
    (1, 2, 10, 10, 3, 3, 3, (3, 3), (1, 1)),    # stride 3: parity classes with no tap
]


@pytest.mark.parametrize('case', CASES)
def test_conv_plans_match_torch(emul, case):
    n, c, h, w_, k, r, s, stride, pad = case
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, c, h, w_, generator=gen)
    w = torch.randn(k, c, r, s, generator=gen)
    bias = torch.randn(k, generator=gen)
    y_ref = F.conv2d(x, w, bias, stride, pad)
    geom = geom_of(x, w, stride, pad, y_ref)
    y = torch.full_like(y_ref, float('nan'))
    assert emul.emul_conv2d_fwd(ctypes.byref(geom), fptr(x), fptr(w), fptr(bias), fptr(y)) == 0
    torch.testing.assert_close(y, y_ref, rtol=1e-5, atol=1e-5)

    gy = torch.randn(y_ref.shape, generator=gen)
    gx_ref = torch.nn.grad.conv2d_input(x.shape, w, gy, stride, pad)
    gx = torch.full_like(x, float('nan'))
    launches = emul.emul_conv2d_bwd_data(ctypes.byref(geom), fptr(gy), fptr(w), None, fptr(gx), 0)
    assert launches >= 1
    torch.testing.assert_close(gx, gx_ref, rtol=1e-5, atol=1e-5)
    # bias on the data-gradient side (= forward of a transposed convolution) and accumulate mode
    cbias = torch.randn(c, generator=gen)
    gx2 = gx_ref.clone()
    emul.emul_conv2d_bwd_data(ctypes.byref(geom), fptr(gy), fptr(w), fptr(cbias), fptr(gx2), 1)
    torch.testing.assert_close(gx2, 2 * gx_ref + cbias.view(1, -1, 1, 1), rtol=1e-5, atol=1e-5)

    gw_ref = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride, pad)
    gw = torch.full_like(w, float('nan'))
    assert emul.emul_conv2d_bwd_weight(ctypes.byref(geom), fptr(x), fptr(gy), fptr(gw), 0) == 0
    torch.testing.assert_close(gw, gw_ref, rtol=1e-4, atol=1e-4)


def test_transposed_convolution_is_backward_data(emul):
    """ConvTranspose2d forward == conv backward-data with the same [Cin, Cout, kh, kw] weight
    (reference age/models.py:16-21 builds the generator from these)."""
    gen = torch.Generator().manual_seed(3)
    for (cin, cout, k, s, p, hin) in [(6, 4, 4, 2, 1, 5), (5, 3, 3, 1, 0, 1), (7, 1, 4, 4, 0, 3)]:
        z = torch.randn(2, cin, hin, hin, generator=gen)
        w = torch.randn(cin, cout, k, k, generator=gen)
        b = torch.randn(cout, generator=gen)
        ref = F.conv_transpose2d(z, w, b, stride=s, padding=p)
        out = torch.full_like(ref, float('nan'))
        geom = ConvGeom(2, cout, ref.shape[2], ref.shape[3], cin, k, k, s, s, p, p, hin, hin, 0, 0)
        assert emul.emul_conv2d_bwd_data(ctypes.byref(geom), fptr(z), fptr(w), fptr(b), fptr(out), 0) >= 1
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_channel_slice_views(emul):
    """Batch strides larger than C*H*W address a channel slice of a wider block buffer."""
    gen = torch.Generator().manual_seed(4)
    wide = torch.randn(2, 10, 6, 6, generator=gen)
    w = torch.randn(4, 6, 3, 3, generator=gen)
    ref = F.conv2d(wide[:, :6], w, None, 1, 1)
    out_wide = torch.zeros(2, 9, 6, 6)
    geom = ConvGeom(2, 6, 6, 6, 4, 3, 3, 1, 1, 1, 1, 6, 6, 10 * 36, 9 * 36)
    out_view = out_wide[:, 5:]
    emul.emul_conv2d_fwd(ctypes.byref(geom), fptr(wide), fptr(w), None, ctypes.c_void_p(out_view.data_ptr()))
    torch.testing.assert_close(out_wide[:, 5:], ref, rtol=1e-5, atol=1e-5)
    assert out_wide[:, :5].abs().sum() == 0


def test_gemm_plans(emul):
    gen = torch.Generator().manual_seed(5)
    x, w, b = torch.randn(7, 11, generator=gen), torch.randn(5, 11, generator=gen), torch.randn(5, generator=gen)
    out = torch.empty(7, 5)
    flags = emul.emul_gemm(7, 5, 11, fptr(x), 11, 1, fptr(w), 1, 11, fptr(out), 5, 1, fptr(b), 1, 0)
    assert flags == 3  # both operands are contiguous along k
    torch.testing.assert_close(out, F.linear(x, w, b), rtol=1e-5, atol=1e-5)
    gy = torch.randn(7, 5, generator=gen)
    gx = torch.empty(7, 11)
    emul.emul_gemm(7, 11, 5, fptr(gy), 5, 1, fptr(w), 11, 1, fptr(gx), 11, 1, None, 0, 0)
    torch.testing.assert_close(gx, gy @ w, rtol=1e-5, atol=1e-5)
    gw = torch.ones(5, 11)
    emul.emul_gemm(5, 11, 7, fptr(gy), 1, 5, fptr(x), 11, 1, fptr(gw), 11, 1, None, 0, 1)
    torch.testing.assert_close(gw, gy.t() @ x + 1, rtol=1e-5, atol=1e-5)


def test_fastdiv_exhaustive_edges(emul):
    rng = np.random.default_rng(0)
    for d in [1, 2, 3, 5, 7, 9, 16, 49, 147, 1152, 12544, 16384, 262144, 2 ** 30 + 7]:
        for n in list(rng.integers(0, 2 ** 31 - 1, 200)) + [0, 1, d - 1, d, d + 1, 2 ** 31 - 1]:
            assert emul.emul_fastdiv_check(int(d), int(n)) == 1, (d, n)


def test_host_plan_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """SURVEY.md 5 asks for ``-fsanitize=address`` on the host half of the library: tests/csrc/sanitize_conv_plans.cpp runs
    the plan builders of conv_plan.h / gather_gemm.h through the CPU emulator on exactly-sized buffers (forward, data and
    weight gradient of every geometry family incl. the 66 x 200 driving frame, the strided GEMM, the magic-number
    division), built with ASan + UBSan; any report aborts the program."""
    binary = str(tmp_path / 'sanitize_conv_plans')
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-fno-omit-frame-pointer', '-I' + os.path.join(ROOT, 'sr-gan_amd', 'csrc'),
                           os.path.join(ROOT, 'tests', 'csrc', 'sanitize_conv_plans.cpp'), '-o', binary])
    done = subprocess.run([binary], capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    assert done.stdout.strip().startswith('ok:'), done.stdout
