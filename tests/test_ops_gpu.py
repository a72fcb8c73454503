"""GPU parity tests of the HIP operations (through the C ABI) against plain PyTorch-CPU fp32 references of
the same ops -- the functions the reference itself calls (torch.nn.functional.conv2d etc.).
Tolerance: 1e-3 relative to the tensor's magnitude (north_star), fp32 exact-MFMA results are typically 1e-6."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """CPU: the shared object loads and exports every function include/srgan_hip.h declares."""
    import srgan_amd  # noqa: F401
    from srgan_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'srgan_hip.h')).read()
    declared = set(re.findall(r'\b(srgan_[a-z0-9_]+)\s*\(', header))
    declared.discard('srgan_conv_desc')
    assert len(declared) >= 20
    lib = _lib.library()
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} is declared in srgan_hip.h but not exported'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.library().srgan_version() == 110


def test_product_fails_loudly_without_gpu_tensors():
    """CPU: the product path has no CPU fallback."""
    import srgan_amd  # noqa: F401
    from srgan_amd import functional as F, _lib
    with pytest.raises(_lib.HipLibraryError):
        F.leaf(torch.zeros(4))


gpu = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    import srgan_amd  # noqa: F401
    from srgan_amd import functional
    assert torch.cuda.is_available()
    return functional


def dev(t):
    return t.cuda()


def close(actual, expected, rtol=1e-3, what=''):
    actual = actual.cpu() if hasattr(actual, 'cpu') else actual
    actual = actual.detach().double().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual, np.float64)
    expected = expected.detach().double().numpy()
    assert actual.shape == expected.shape, f'{what}: {actual.shape} vs {expected.shape}'
    denom = max(np.abs(expected).max(), 1e-20)
    err = np.abs(actual - expected).max() / denom
    assert np.isfinite(actual).all(), f'{what}: non-finite values'
    assert err <= rtol, f'{what}: max err / max|ref| = {err:.3e}'


CONV_CASES = [
    # N, C, H, W, K, R, S, stride, pad
    (2, 3, 9, 8, 5, 3, 3, (1, 1), (1, 1)),
    (2, 4, 7, 7, 6, 1, 1, (1, 1), (0, 0)),
    (2, 3, 12, 10, 4, 4, 4, (2, 2), (1, 1)),
    (1, 3, 15, 13, 4, 7, 7, (2, 2), (3, 3)),
    (2, 2, 8, 8, 3, 2, 2, (2, 2), (0, 0)),
    (3, 5, 4, 6, 7, 4, 6, (1, 1), (0, 0)),
    (1, 2, 10, 10, 3, 3, 3, (3, 3), (1, 1)),
    # DenseNet / DCGAN / VGG tile-boundary shapes (multi-tile, split-K, all MFMA tile configs)
    (4, 256, 28, 28, 128, 1, 1, (1, 1), (0, 0)),
    (4, 128, 28, 28, 32, 3, 3, (1, 1), (1, 1)),
    (3, 200, 7, 7, 128, 1, 1, (1, 1), (0, 0)),
    (2, 3, 64, 64, 64, 7, 7, (2, 2), (3, 3)),
    (2, 64, 32, 32, 128, 4, 4, (2, 2), (1, 1)),
    (2, 64, 20, 20, 64, 3, 3, (1, 1), (1, 1)),
    (2, 48, 8, 8, 1, 8, 8, (8, 8), (0, 0)),          # map head in conv form (K = 1 -> direct kernel)
    (2, 32, 16, 16, 20, 16, 16, (1, 1), (0, 0)),     # "linear" conv with a long reduction
    (5, 70, 9, 9, 40, 3, 3, (1, 1), (1, 1)),         # ragged in every dimension
    # pointwise shapes eligible for 16-byte staging (all extents multiples of 4) and near-misses
    (2, 64, 16, 16, 96, 1, 1, (1, 1), (0, 0)),
    (2, 160, 32, 32, 128, 1, 1, (1, 1), (0, 0)),
    (1, 36, 8, 12, 20, 1, 1, (1, 1), (0, 0)),
    (2, 64, 16, 18, 32, 1, 1, (1, 1), (0, 0)),
    (3, 896, 16, 16, 448, 1, 1, (1, 1), (0, 0)),
    # data gradient with <= 128 output channels: the resident-weight kernel, remainder tiles of 96 / 64 rows
    (2, 224, 16, 16, 128, 1, 1, (1, 1), (0, 0)),
    # few pixels, many input channels: the forward takes the kernel that splits K over the waves of a workgroup
    (2, 288, 16, 16, 100, 1, 1, (1, 1), (0, 0)),
    (4, 1024, 8, 8, 136, 1, 1, (1, 1), (0, 0)),
    (2, 192, 8, 32, 64, 1, 1, (1, 1), (0, 0)),
    # many input channels on planes that are no multiple of 32 pixels (the 14 x 14 and 7 x 7 planes of the reference's 224 x 224
    # patches): the streaming kernel with ragged pixel groups and a K split finished in a fixed order
    (16, 256, 14, 14, 128, 1, 1, (1, 1), (0, 0)),
    (4, 1024, 7, 7, 128, 1, 1, (1, 1), (0, 0)),
    (3, 512, 14, 14, 136, 1, 1, (1, 1), (0, 0)),
    (1, 320, 6, 6, 40, 1, 1, (1, 1), (0, 0)),
    # 3x3 / s1 / p1 shapes for the LDS-halo kernel (force = 0): all three channel-tile widths, ragged tiles,
    # fewer input channels than one chunk, split over input-channel chunks
    (2, 128, 32, 32, 32, 3, 3, (1, 1), (1, 1)),
    (1, 3, 40, 48, 64, 3, 3, (1, 1), (1, 1)),
    (2, 32, 16, 64, 130, 3, 3, (1, 1), (1, 1)),
    (2, 24, 33, 35, 16, 3, 3, (1, 1), (1, 1)),
    (16, 128, 64, 64, 32, 3, 3, (1, 1), (1, 1)),
    (2, 128, 16, 16, 32, 3, 3, (1, 1), (1, 1)),      # 16-wide images: two image rows per 32-lane column block
    (3, 40, 13, 16, 24, 3, 3, (1, 1), (1, 1)),
    (2, 128, 14, 14, 32, 3, 3, (1, 1), (1, 1)),      # 14-wide planes of the 224-pixel configuration: two dead columns per tile
    (2, 32, 14, 13, 128, 3, 3, (1, 1), (1, 1)),
    (3, 128, 7, 7, 32, 3, 3, (1, 1), (1, 1)),
    (3, 200, 20, 24, 40, 3, 3, (1, 1), (1, 1)),      # LDS-patch weight gradient: ragged channel chunks and tiles
    (4, 3, 96, 96, 16, 7, 7, (2, 2), (3, 3)),        # stem: the 3-row image gradient takes the few-rows kernel
    (2, 5, 72, 72, 7, 3, 3, (1, 1), (1, 1)),         # 5 and 7 rows (MR = 8) in the few-rows kernel
    (8, 2, 256, 128, 3, 2, 2, (2, 2), (0, 0)),       # weight gradient 3 x 8 over K = 65536 pixels: lanes-along-K kernel
    # k4 / s2 / p1 at the DCGAN pair's own shapes (reference age/models.py:61-65 on 64 x 192 driving frames and 128 x 128
    # faces, crowd/models.py:132-136 backwards): three input channels, rectangular and ragged planes, 24- to 512-row outputs,
    # K splits with the ordered finish
    (3, 3, 64, 192, 64, 4, 4, (2, 2), (1, 1)),
    (2, 64, 32, 96, 128, 4, 4, (2, 2), (1, 1)),
    (2, 128, 16, 48, 256, 4, 4, (2, 2), (1, 1)),
    (2, 256, 16, 24, 512, 4, 4, (2, 2), (1, 1)),
    (5, 20, 36, 44, 24, 4, 4, (2, 2), (1, 1)),
    (1, 64, 128, 128, 32, 4, 4, (2, 2), (1, 1)),
    # the LDS-DMA 1x1 kernel (pointwise_ring.hip; whole 128-row tiles, >= 192 workgroups): 128- and 64-pixel tiles, weights
    # k-contiguous (forward) and m-contiguous (data gradient), several row tiles, remainder rows of 32 / 96 on the old kernel
    (16, 128, 64, 64, 128, 1, 1, (1, 1), (0, 0)),
    (16, 256, 32, 32, 128, 1, 1, (1, 1), (0, 0)),
    (4, 160, 64, 64, 288, 1, 1, (1, 1), (0, 0)),
    (6, 96, 64, 64, 224, 1, 1, (1, 1), (0, 0)),
    (24, 128, 24, 12, 128, 1, 1, (1, 1), (0, 0)),    # planes of 9 x 32 pixels: the 32-pixel tile, both weight layouts
    # the map modules' 2x2 / s2 convolutions (reference crowd/models.py:131-133) at their own channel counts: tiny weight
    # gradients from 10^4 - 10^5 pixels (many K slices, ordered finish), rectangular planes, an odd input height
    (3, 8, 128, 128, 16, 2, 2, (2, 2), (0, 0)),
    (2, 16, 64, 128, 32, 2, 2, (2, 2), (0, 0)),
    (2, 1, 64, 64, 8, 2, 2, (2, 2), (0, 0)),
    (2, 5, 67, 64, 20, 2, 2, (2, 2), (0, 0)),
    (1, 3, 4, 64, 7, 2, 2, (2, 2), (0, 0)),
]


@gpu
@pytest.mark.parametrize('force', [0, 1, 2])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_passes(F, case, force):
    n, c, h, w_, k, r, s, stride, pad = case
    if force == 1 and n * c * h * w_ * k * r * s > 3e8:
        pytest.skip('direct kernel is only a cross-check for small shapes')
    gen = torch.Generator().manual_seed(abs(hash(case)) % 1000)
    x = torch.randn(n, c, h, w_, generator=gen)
    w = torch.randn(k, c, r, s, generator=gen) / (c * r * s) ** 0.5
    b = torch.randn(k, generator=gen)
    y_ref = TF.conv2d(x, w, b, stride, pad)
    gy = torch.randn(y_ref.shape, generator=gen)
    F.FORCE_KERNEL = force
    try:
        xv, wv, bv, gyv = (F.leaf(dev(t)) for t in (x, w, b, gy))
        close(F.conv2d(xv, wv, bv, stride, pad), y_ref, what='fwd')
        close(F.conv2d_backward_data(gyv, wv, x.shape, stride, pad), torch.nn.grad.conv2d_input(x.shape, w, gy, stride, pad),
              what='bwd_data')
        close(F.conv2d_backward_weight(xv, gyv, w.shape, stride, pad),
              torch.nn.grad.conv2d_weight(x, w.shape, gy, stride, pad), what='bwd_weight')
    finally:
        F.FORCE_KERNEL = 0


@gpu
def test_conv_transpose(F):
    gen = torch.Generator().manual_seed(3)
    for (cin, cout, k, s, p, hin, batch) in [(6, 4, 4, 2, 1, 5, 2), (32, 48, 3, 1, 0, 1, 3), (40, 1, 4, 4, 0, 6, 2),
                                             (64, 3, 4, 2, 1, 16, 2), (256, 64, 2, 1, 0, 1, 4),
                                             (64, 3, 4, 2, 1, 64, 2)]:    # generator output layer: few-rows kernel
        z = torch.randn(batch, cin, hin, hin, generator=gen)
        w = torch.randn(cin, cout, k, k, generator=gen) / cin ** 0.5
        b = torch.randn(cout, generator=gen)
        ref = TF.conv_transpose2d(z, w, b, stride=s, padding=p)
        out = F.conv_transpose2d(F.leaf(dev(z)), F.leaf(dev(w)), F.leaf(dev(b)), s, p)
        close(out, ref, what=f'convT {cin}->{cout} k{k}s{s}')


@gpu
@pytest.mark.parametrize('force', [0, 2])
def test_linear_and_mm(F, force):
    gen = torch.Generator().manual_seed(5)
    F.FORCE_KERNEL = force
    try:
        for (bsz, fin, fout) in [(7, 11, 5), (256, 50, 10), (64, 300, 130), (2, 25088 // 8, 512), (130, 64, 1),
                                 (16, 65536, 20)]:   # tiny output, long K: split-K through the partial-sum workspace
            x, w, b = torch.randn(bsz, fin, generator=gen), torch.randn(fout, fin, generator=gen), torch.randn(fout, generator=gen)
            xv, wv, bv = F.leaf(dev(x)), F.leaf(dev(w)), F.leaf(dev(b))
            close(F.linear(xv, wv, bv), TF.linear(x, w, b), what=f'linear {bsz}x{fin}x{fout}')
            gy = torch.randn(bsz, fout, generator=gen)
            gv = F.leaf(dev(gy))
            close(F.mm(gv, wv), gy @ w, what='g @ w')
            close(F.mm(gv, xv, True, False), gy.t() @ x, what='g^T @ x')
            close(F.mm(xv, gv, True, False), x.t() @ gy, what='x^T @ g')
            close(F.mm(wv, xv, False, True), w @ x.t(), what='w @ x^T')
    finally:
        F.FORCE_KERNEL = 0


@gpu
def test_elementwise(F):
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(1031, generator=gen)
    pos = x.abs() + 0.1
    xv, pv = F.leaf(dev(x)), F.leaf(dev(pos))
    cases = [(F.neg(xv), -x), (F.abs_(xv), x.abs()), (F.sign(xv), x.sign()), (F.sqrt(pv), pos.sqrt()), (F.exp(xv), x.exp()),
             (F.log(pv), pos.log()), (F.log1p(pv), pos.log1p()), (F.square(xv), x * x), (F.tanh(xv), x.tanh()),
             (F.relu(xv), x.relu()), (F.leaky_relu(xv, 0.05), TF.leaky_relu(x, 0.05)), (F.affine(xv, 2.5, -1.0), 2.5 * x - 1),
             (F.pow_scalar(pv, 3.0), pos ** 3), (F.pow_scalar(pv, 1.5), pos ** 1.5), (F.sigmoid(xv), x.sigmoid()),
             (F.softplus(xv), TF.softplus(x)), (F.one_minus_square(xv), 1 - x * x)]
    for i, (actual, expected) in enumerate(cases):
        close(actual, expected, rtol=1e-5, what=f'unary {i}')
    y = torch.randn(1031, generator=gen)
    y[5] = 0.0
    yv = F.leaf(dev(y))
    close(F.add(xv, yv), x + y, 1e-6)
    close(F.sub(xv, yv), x - y, 1e-6)
    close(F.mul(xv, yv), x * y, 1e-6)
    close(F.div(xv, pv), x / pos, 1e-6)
    safe = torch.where(y == 0, torch.zeros_like(x), x / y)
    close(F.div_safe(xv, yv), safe, 1e-6)
    close(F.mask_mul(xv, yv, 0.01), torch.where(y > 0, x, 0.01 * x), 1e-6)


@gpu
def test_channel_ops_and_frozen_batch_norm(F):
    gen = torch.Generator().manual_seed(7)
    for shape in [(3, 5, 6, 7), (2, 4, 40, 40), (4, 300, 1, 1), (2, 7, 3, 3)]:
        x = torch.randn(shape, generator=gen)
        c = shape[1]
        mean, var = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
        gamma, beta = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
        ref = TF.batch_norm(x, mean, var, gamma, beta, training=False, eps=1e-5)
        inv = (var + 1e-5).rsqrt()
        out = F.chan_affine(F.leaf(dev(x)), F.leaf(dev(mean)), F.leaf(dev(inv)), F.leaf(dev(gamma)), F.leaf(dev(beta)))
        close(out, ref, 1e-5, what=f'bn {shape}')
        g = torch.randn(shape, generator=gen)
        close(F.chan_reduce(F.leaf(dev(g))), g.sum(dim=(0, 2, 3)), 1e-5, 'chan sum')
        expected = (g * (x - mean.view(1, -1, 1, 1))).sum(dim=(0, 2, 3)) * inv
        close(F.chan_reduce(F.leaf(dev(g)), F.leaf(dev(x)), F.leaf(dev(mean)), F.leaf(dev(inv))), expected, 1e-4, 'gamma grad')
    feats = torch.randn(16, 70000, generator=gen)
    fv = F.leaf(dev(feats))
    close(F.row_dot(fv, fv), (feats * feats).sum(1), 1e-5, 'row_dot')
    close(F.row_norm(fv), feats.norm(dim=1), 1e-5, 'row_norm')
    close(F.col_sum(fv), feats.sum(0), 1e-4, 'col_sum')
    close(F.sum_all(fv), feats.sum().view(1), 1e-3, 'sum_all')
    s = torch.randn(16, generator=gen)
    close(F.row_scale(fv, F.leaf(dev(s))), feats * s.view(-1, 1), 1e-6, 'row_scale')
    close(F.row_broadcast(F.leaf(dev(s)), (16, 33)), s.view(-1, 1).expand(16, 33), 1e-6, 'row_broadcast')


@gpu
def test_row_reductions_are_one_launch_in_a_fixed_order(F):
    """The per-example norms of the gradient penalty (reference srgan.py:371) and the other few-row reductions: workgroup
    partials through the workspace, the last workgroup of a row adds them in a fixed order (reduce.hip).  Values against
    torch in float64; two runs give the same BITS; two streams reducing at once do not share tickets; the accumulate form
    adds to what is there; rows long and short, aligned and not, with the batch-norm style centring."""
    from srgan_amd import _lib
    lib = _lib.library()
    gen = torch.Generator().manual_seed(11)
    for rows, length in [(16, 3 * 512 * 512), (16, 70001), (1, 1 << 22), (64, 12345), (3, 8 * 4096 + 4)]:
        x, y = torch.randn(rows, length, generator=gen), torch.randn(rows, length, generator=gen)
        xv, yv = F.leaf(dev(x)), F.leaf(dev(y))
        first = F.row_dot(xv, xv).data.clone()
        close(first, (x.double() * x.double()).sum(1).float(), 2e-6, f'squared norms {rows} x {length}')
        close(F.row_dot(xv, yv), (x.double() * y.double()).sum(1).float(), 1e-4 if rows > 1 else 1e-3, 'row_dot')
        for _ in range(3):
            assert torch.equal(F.row_dot(xv, xv).data, first), 'the reduction order moved between two runs'
        # accumulate: out += ...
        out = torch.full((rows,), 2.5, device='cuda')
        _lib.check(lib.srgan_chan_reduce(xv.data.data_ptr(), None, None, None, out.data_ptr(), 1, rows, length, 1,
                                         _lib.stream_handle()), 'srgan_chan_reduce')
        close(out, x.double().sum(1).float() + 2.5, 1e-3, 'accumulate')
    # two streams at once, many times: each stream's results must be its own
    a, b = dev(torch.randn(16, 400000, generator=gen)), dev(torch.randn(16, 400000, generator=gen))
    av, bv = F.leaf(a), F.leaf(b)
    want_a, want_b = F.row_dot(av, av).data.clone(), F.row_dot(bv, bv).data.clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    got_a, got_b = [], []
    for _ in range(20):
        got_a.append(F.row_dot(av, av).data)
        with torch.cuda.stream(side):
            got_b.append(F.row_dot(bv, bv).data)
    torch.cuda.synchronize()
    assert all(torch.equal(g, want_a) for g in got_a) and all(torch.equal(g, want_b) for g in got_b)


@gpu
def test_fused_batch_norm_backward(F):
    """srgan_bn_act_bwd: input gradient + both parameter gradients of frozen BN(+ReLU) in one pass, against torch
    autograd; dense, odd (scalar-path) and channel-slice / accumulate forms."""
    import ctypes
    from srgan_amd import tape, _lib
    gen = torch.Generator().manual_seed(17)
    for shape in [(3, 5, 6, 7), (2, 4, 40, 40), (2, 6, 32, 48), (4, 9, 1, 1)]:
        for relu in (False, True):
            x = torch.randn(shape, generator=gen)
            c = shape[1]
            mean, var = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
            gamma, beta = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
            g = torch.randn(shape, generator=gen)
            tx, tg, tb = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
            ref = TF.batch_norm(tx, mean, var, tg, tb, training=False, eps=1e-5)
            ref = ref.relu() if relu else ref
            ref.backward(g)
            inv = (var + 1e-5).rsqrt()
            xv, gv, bv = (F.leaf(dev(t), requires_grad=True) for t in (x, gamma, beta))
            out = F.batch_norm_eval(xv, F.leaf(dev(mean)), F.leaf(dev(inv)), gv, bv, relu=relu)
            close(out, ref.detach(), 1e-5, what=f'bn fwd {shape} relu={relu}')
            tape.backward(out, F.leaf(dev(g)))
            close(xv.grad, tx.grad, 1e-5, what=f'bn gx {shape} relu={relu}')
            close(gv.grad, tg.grad, 1e-4, what=f'bn ggamma {shape} relu={relu}')
            close(bv.grad, tb.grad, 1e-4, what=f'bn gbeta {shape} relu={relu}')
    # channel-slice view of a wider buffer, gradient accumulated into a slice of a wider gradient buffer
    n, c, total, h, w = 2, 5, 9, 16, 16
    wide = torch.randn(n, total, h, w, generator=gen)
    gwide = torch.randn(n, total, h, w, generator=gen)
    g = torch.randn(n, c, h, w, generator=gen)
    mean, var = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
    gamma, beta = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
    inv = (var + 1e-5).rsqrt()
    tx = wide[:, :c].clone().requires_grad_()
    TF.batch_norm(tx, mean, var, gamma, beta, training=False, eps=1e-5).relu().backward(g)
    expected = gwide.clone()
    expected[:, :c] += tx.grad
    d = {k: dev(v) for k, v in dict(wide=wide, gwide=gwide, g=g, mean=mean, inv=inv, gamma=gamma, beta=beta).items()}
    _lib.check(_lib.library().srgan_bn_act_bwd(d['g'].data_ptr(), d['wide'].data_ptr(), d['mean'].data_ptr(),
                                               d['inv'].data_ptr(), d['gamma'].data_ptr(), d['beta'].data_ptr(), 1,
                                               d['gwide'].data_ptr(), None, None, n, c, h * w, 0, total * h * w,
                                               total * h * w, 1, 0, torch.cuda.current_stream().cuda_stream), 'bn_act_bwd')
    close(d['gwide'], expected, 1e-5, what='bn_act_bwd strided accumulate')


@gpu
def test_pooling_and_layout(F):
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(2, 5, 13, 11, generator=gen)
    xv = F.leaf(dev(x), requires_grad=True)
    for (k, s, p) in [(3, 2, 1), (2, 2, 0)]:
        close(F.max_pool2d(xv, k, s, p), TF.max_pool2d(x, k, s, p), 1e-6, f'maxpool k{k}')
    xr = x.clone().requires_grad_()
    ref = TF.max_pool2d(xr.relu(), 3, 2, 1)     # many ties at zero: arg-max choice must match torch
    g = torch.randn(ref.shape, generator=gen)
    ref.backward(g)
    from srgan_amd.tape import backward
    out = F.max_pool2d(F.relu(xv), 3, 2, 1)
    backward(out, grad=F.leaf(dev(g)))
    close(xv.grad, xr.grad, 1e-6, 'maxpool backward with ties')
    for (k, s, p, h, w) in [(2, 2, 0, 12, 10), (3, 1, 1, 9, 7), (3, 2, 1, 16, 16), (3, 3, 0, 10, 11), (2, 1, 0, 5, 6),
                            (2, 2, 0, 12, 8), (3, 2, 1, 15, 20), (3, 2, 1, 8, 4)]:     # four-pixels-per-thread forms
        x3 = torch.randn(3, 4, h, w, generator=gen).requires_grad_()      # gather-form backward: every window overlap
        ref = TF.max_pool2d(x3, k, s, p)
        g = torch.randn(ref.shape, generator=gen)
        ref.backward(g)
        x3v = F.leaf(dev(x3.detach()), requires_grad=True)
        backward(F.max_pool2d(x3v, k, s, p), grad=F.leaf(dev(g)))
        close(x3v.grad, x3.grad, 1e-6, f'maxpool backward k{k} s{s} p{p}')
    # (rows of whole float4s take the four-pixels-per-thread backward, 10 x 10 the scalar one; 16 / 16 = a global pool)
    for (k, s, size) in [(2, 2, 12), (3, 1, 12), (4, 4, 12), (2, 2, 10), (3, 2, 10), (16, 16, 16)]:
        x2 = torch.randn(2, 3, size, size, generator=gen).requires_grad_()
        ref = TF.avg_pool2d(x2, k, s)
        g = torch.randn(ref.shape, generator=gen)
        ref.backward(g)
        x2v = F.leaf(dev(x2.detach()), requires_grad=True)
        out = F.avg_pool2d(x2v, k, s)
        close(out, ref, 1e-6, 'avgpool')
        backward(out, grad=F.leaf(dev(g)))
        close(x2v.grad, x2.grad, 1e-6, 'avgpool backward')
    a, b = torch.randn(2, 3, 4, 4, generator=gen), torch.randn(2, 5, 4, 4, generator=gen)
    cat = F.cat_channels([F.leaf(dev(a)), F.leaf(dev(b))])
    close(cat, torch.cat([a, b], 1), 1e-7, 'cat')
    close(F.slice_channels(cat, 2, 6), torch.cat([a, b], 1)[:, 2:6], 1e-7, 'slice')


@gpu
def test_gradient_penalty_double_backward_matches_torch(F):
    """conv -> leaky -> conv(stride 2) -> frozen BN -> relu -> maxpool -> linear features; penalty on the input
    gradient; parameter gradients of the penalty through the double backward."""
    from srgan_amd.tape import backward
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 2, 10, 10, generator=gen)
    w1 = torch.randn(6, 2, 3, 3, generator=gen) * 0.5
    b1 = torch.randn(6, generator=gen) * 0.1
    w2 = torch.randn(4, 6, 4, 4, generator=gen) * 0.3
    gamma, beta = torch.rand(4, generator=gen) + 0.5, torch.randn(4, generator=gen) * 0.1
    mean, var = torch.randn(4, generator=gen) * 0.1, torch.rand(4, generator=gen) + 0.5
    wl = torch.randn(5, 36, generator=gen) * 0.5
    params = [w1, b1, w2, gamma, beta, wl]
    tp = [p.clone().requires_grad_() for p in params]
    xt = x.clone().requires_grad_()
    h = TF.leaky_relu(TF.conv2d(xt, tp[0], tp[1], 1, 1), 0.05)
    h = TF.conv2d(h, tp[2], None, 2, 1)
    h = TF.batch_norm(h, mean, var, tp[3], tp[4], training=False).relu()
    h = TF.max_pool2d(h, 3, 2, 1)
    feats = TF.linear(h.reshape(3, -1), tp[5])
    f = feats.norm(dim=1)
    (gx,) = torch.autograd.grad(f, xt, torch.ones_like(f), create_graph=True)
    gn = gx.reshape(3, -1).norm(dim=1)
    penalty = (torch.relu(gn - 0.1) ** 2).mean() * 10
    penalty.backward()

    pv = [F.leaf(dev(p), requires_grad=True) for p in params]
    xv = F.leaf(dev(x), requires_grad=True)
    hv = F.leaky_relu(F.conv2d(xv, pv[0], pv[1], 1, 1), 0.05)
    hv = F.conv2d(hv, pv[2], None, 2, 1)
    inv = F.leaf(dev((var + 1e-5).rsqrt()))
    hv = F.relu(F.chan_affine(hv, F.leaf(dev(mean)), inv, pv[3], pv[4]))
    hv = F.max_pool2d(hv, 3, 2, 1)
    fv = F.row_norm(F.linear(F.flatten2d(hv), pv[5]))
    close(fv, f, 1e-5, 'feature norm')
    (gxv,) = backward(fv, grad=F.full_like(fv, 1.0), inputs=[xv], create_graph=True)
    close(gxv, gx, 1e-4, 'input gradient')
    gnv = F.row_norm(F.flatten2d(gxv))
    pen = F.scale(F.mean_all(F.square(F.relu(F.add_scalar(gnv, -0.1)))), 10.0)
    close(pen, penalty.view(1), 1e-4, 'penalty')
    backward(pen)
    for i, (v, t) in enumerate(zip(pv, tp)):
        close(v.grad, t.grad, 1e-3, f'penalty grad of param {i}')


@gpu
def test_adam_matches_torch(F):
    import srgan_amd  # noqa: F401
    from srgan_amd import nn
    from srgan_amd.optim import Adam
    torch.manual_seed(0)
    module = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    reference = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    reference.load_state_dict(module.state_dict())
    arena = nn.flatten_parameters(module, torch.device('cuda', 0))
    optimizer = Adam(arena, lr=1e-2, weight_decay=0.1)
    reference_optimizer = torch.optim.Adam(reference.parameters(), lr=1e-2, weight_decay=0.1)
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        for p, q in zip(module.parameters(), reference.parameters()):
            g = torch.randn(q.shape, generator=gen)
            q.grad = g.clone()
            p.grad.copy_(g)
        optimizer.step()
        reference_optimizer.step()
    for p, q in zip(module.parameters(), reference.parameters()):
        close(p.data, q.data, 1e-6, 'adam parameter')
    state = optimizer.state_dict()
    assert set(state['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    close(state['state'][0]['exp_avg'], reference_optimizer.state_dict()['state'][0]['exp_avg'], 1e-6, 'exp_avg')


@gpu
def test_row_stacking_ops(F):
    """cat_rows / narrow_rows (the stacked discriminator pass): values, views and both gradients."""
    from srgan_amd.tape import backward
    gen = torch.Generator().manual_seed(31)
    a, b = torch.randn(3, 4, 5, generator=gen), torch.randn(2, 4, 5, generator=gen)
    av, bv = F.leaf(dev(a), requires_grad=True), F.leaf(dev(b), requires_grad=True)
    stacked = F.cat_rows([av, bv])
    close(stacked, torch.cat([a, b]), 0.0, 'cat_rows')
    middle = F.narrow_rows(stacked, 1, 3)
    close(middle, torch.cat([a, b])[1:4], 0.0, 'narrow_rows')
    assert middle.data.data_ptr() == stacked.data.data_ptr() + 4 * 20        # a view, not a copy
    cotangent = torch.randn(3, 4, 5, generator=gen)
    backward(middle, grad=F.leaf(dev(cotangent)))
    expected = torch.zeros(5, 4, 5)
    expected[1:4] = cotangent
    close(av.grad, expected[:3], 0.0, 'cat_rows / narrow_rows gradient (first part)')
    close(bv.grad, expected[3:], 0.0, 'cat_rows / narrow_rows gradient (second part)')
    with pytest.raises(ValueError):
        F.narrow_rows(stacked, 4, 3)


@gpu
def test_fused_batch_norm_convolutions(F):
    """srgan_conv2d_fwd_bnrelu / srgan_conv2d_bwd_weight_bnrelu (normalisation evaluated inside the convolution
    kernels) against the two-step form relu(bn(x)) -> conv of torch, 1x1 and 3x3, dense and channel-slice inputs."""
    from srgan_amd import _lib
    lib = _lib.library()
    stream = _lib.stream_handle()
    gen = torch.Generator().manual_seed(23)
    for (n, c, total, h, w, k, r) in [(2, 48, 80, 16, 16, 32, 1), (3, 160, 160, 8, 32, 128, 1), (2, 320, 352, 16, 16, 96, 1),
                                      (1, 512, 512, 8, 8, 40, 1), (2, 32, 32, 16, 16, 8, 3),
                                      (2, 128, 128, 32, 32, 32, 3), (1, 70, 96, 20, 24, 40, 3),
                                      # the ragged planes of the reference's 224 x 224 (28 / 14 / 7 wide), dense and as
                                      # channel slices whose rows are only 4-byte aligned
                                      (3, 160, 200, 28, 28, 128, 1), (2, 200, 264, 14, 14, 128, 1), (5, 96, 131, 7, 7, 128, 1),
                                      (16, 64, 64, 7, 7, 40, 1), (2, 34, 41, 9, 7, 20, 1),
                                      (2, 128, 128, 14, 14, 32, 3), (3, 128, 128, 7, 7, 32, 3), (2, 40, 57, 7, 9, 33, 3),
                                      (2, 128, 128, 28, 28, 32, 3),
                                      # many input channels on ragged planes (block 3 / 4 of the 224-pixel configuration), channel
                                      # slices of a wider block buffer
                                      (16, 1024, 1056, 14, 14, 128, 1), (16, 896, 928, 7, 7, 128, 1), (3, 512, 640, 14, 14, 128, 1),
                                      # the LDS-DMA 1x1 kernel with the prologue: 64-pixel tile on a channel slice, 128-pixel
                                      # tile, two row tiles
                                      (16, 256, 320, 32, 32, 128, 1), (16, 64, 64, 64, 64, 128, 1), (4, 160, 192, 64, 64, 256, 1)]:
        pad = r // 2
        wide = torch.randn(n, total, h, w, generator=gen)
        x = wide[:, :c]
        mean, var = torch.randn(c, generator=gen) * 0.3, torch.rand(c, generator=gen) + 0.5
        gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.3
        weight = torch.randn(k, c, r, r, generator=gen) / (c * r * r) ** 0.5
        act = TF.batch_norm(x, mean, var, gamma, beta, training=False, eps=1e-5).relu()
        y_ref = TF.conv2d(act, weight, None, 1, pad)
        gy = torch.randn(y_ref.shape, generator=gen)
        gw_ref = torch.nn.grad.conv2d_weight(act, weight.shape, gy, 1, pad)
        d = {name: dev(t) for name, t in dict(wide=wide, mean=mean, inv=(var + 1e-5).rsqrt(), gamma=gamma, beta=beta,
                                              weight=weight, gy=gy).items()}
        desc = _lib.ConvDesc(n, c, h, w, k, r, r, 1, 1, pad, pad, h, w, total * h * w, 0)
        bn = _lib.BnRelu(d['mean'].data_ptr(), d['inv'].data_ptr(), d['gamma'].data_ptr(), d['beta'].data_ptr())
        for kind in (0, 2):
            assert lib.srgan_conv2d_bnrelu_supported(desc, kind) == 1, (n, c, h, w, k, r, kind)
        y = torch.empty(y_ref.shape, device='cuda')
        _lib.check(lib.srgan_conv2d_fwd_bnrelu(desc, d['wide'].data_ptr(), bn, d['weight'].data_ptr(), None, y.data_ptr(),
                                               stream), 'fwd_bnrelu')
        close(y, y_ref, what=f'fused bn conv forward {c}->{k} k{r}')
        gw = torch.full(weight.shape, 0.5, device='cuda')
        _lib.check(lib.srgan_conv2d_bwd_weight_bnrelu(desc, d['wide'].data_ptr(), bn, d['gy'].data_ptr(), gw.data_ptr(), 1,
                                                      stream), 'bwd_weight_bnrelu')
        close(gw, gw_ref + 0.5, what=f'fused bn conv weight gradient {c}->{k} k{r} (accumulate)')
        _lib.check(lib.srgan_conv2d_bwd_weight_bnrelu(desc, d['wide'].data_ptr(), bn, d['gy'].data_ptr(), gw.data_ptr(), 0,
                                                      stream), 'bwd_weight_bnrelu')
        close(gw, gw_ref, what=f'fused bn conv weight gradient {c}->{k} k{r}')
    # geometries without a fused form are reported, not guessed: a plane of fewer than 32 pixels has none; few input
    # channels have the fused forward but not the fused weight gradient
    tiny = _lib.ConvDesc(2, 32, 5, 5, 16, 1, 1, 1, 1, 0, 0, 5, 5, 0, 0)
    assert lib.srgan_conv2d_bnrelu_supported(tiny, 0) == 0 and lib.srgan_conv2d_bnrelu_supported(tiny, 2) == 0
    for (n, c, h, w, k) in [(2, 8, 9, 7, 16), (3, 12, 14, 14, 128), (2, 10, 7, 7, 40)]:
        x = torch.randn(n, c, h, w, generator=gen)
        mean, var = torch.randn(c, generator=gen) * 0.3, torch.rand(c, generator=gen) + 0.5
        gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.3
        weight = torch.randn(k, c, 1, 1, generator=gen) / c ** 0.5
        y_ref = TF.conv2d(TF.batch_norm(x, mean, var, gamma, beta, training=False, eps=1e-5).relu(), weight)
        d = {name: dev(t) for name, t in dict(x=x, mean=mean, inv=(var + 1e-5).rsqrt(), gamma=gamma, beta=beta,
                                              weight=weight).items()}
        desc = _lib.ConvDesc(n, c, h, w, k, 1, 1, 1, 1, 0, 0, h, w, 0, 0)
        assert lib.srgan_conv2d_bnrelu_supported(desc, 0) == 1 and lib.srgan_conv2d_bnrelu_supported(desc, 2) == 0
        bn = _lib.BnRelu(d['mean'].data_ptr(), d['inv'].data_ptr(), d['gamma'].data_ptr(), d['beta'].data_ptr())
        y = torch.full(y_ref.shape, float('nan'), device='cuda')
        _lib.check(lib.srgan_conv2d_fwd_bnrelu(desc, d['x'].data_ptr(), bn, d['weight'].data_ptr(), None, y.data_ptr(),
                                               stream), 'fwd_bnrelu')
        close(y, y_ref, what=f'fused bn conv forward on a ragged plane {h}x{w}')


@gpu
def test_fused_batch_norm_backward_in_the_data_gradient(F):
    """srgan_conv2d_bwd_data_bnrelu (data gradient of norm -> relu -> conv w.r.t. the normalisation's input in one
    kernel, parameter gradients included) against torch autograd of the two-step form.  1x1: x / gx are channel-slice
    views of wider buffers, gx stored or accumulated, row counts with and without a remainder tile.  3x3: dense x / gx
    (stored), gy a channel-slice view, 16- and 32-wide tiles, ragged heights."""
    from srgan_amd import _lib
    lib = _lib.library()
    stream = _lib.stream_handle()
    gen = torch.Generator().manual_seed(29)
    for (n, c, total, h, w, k, r) in [(2, 48, 80, 16, 16, 32, 1), (3, 160, 160, 8, 32, 128, 1), (2, 192, 224, 32, 32, 128, 1),
                                      (1, 512, 512, 8, 8, 40, 1), (4, 96, 256, 64, 64, 128, 1), (2, 300, 300, 16, 16, 128, 1),
                                      (4, 128, 128, 32, 32, 32, 3), (16, 128, 128, 16, 16, 32, 3), (2, 128, 128, 64, 64, 32, 3),
                                      (3, 40, 40, 20, 24, 16, 3), (16, 128, 128, 64, 64, 32, 3),
                                      # the planes of the reference's 224 x 224: 28 / 14 wide (whole float4s, ragged
                                      # 32-pixel groups) and 7 x 7 (49 pixels: rows only 4-byte aligned)
                                      (3, 160, 200, 28, 28, 128, 1), (2, 200, 264, 14, 14, 128, 1), (5, 96, 131, 7, 7, 128, 1),
                                      (16, 64, 64, 7, 7, 128, 1), (2, 34, 41, 9, 7, 20, 1),
                                      (2, 128, 128, 14, 14, 32, 3), (3, 128, 128, 7, 7, 32, 3),
                                      # the LDS-DMA 1x1 kernel with the epilogue: two row tiles on a channel slice, three tiles
                                      # + 32 remainder rows, one tile + 96 remainder rows
                                      (16, 256, 320, 32, 32, 128, 1), (8, 416, 512, 32, 32, 128, 1), (8, 224, 256, 64, 64, 128, 1)]:
        pad = r // 2
        wide = torch.randn(n, total, h, w, generator=gen)
        x = wide[:, :c].clone().requires_grad_(True)
        mean, var = torch.randn(c, generator=gen) * 0.3, torch.rand(c, generator=gen) + 0.5
        gamma = (torch.rand(c, generator=gen) + 0.5).requires_grad_(True)
        beta = (torch.randn(c, generator=gen) * 0.3).requires_grad_(True)
        weight = torch.randn(k, c, r, r, generator=gen) / (c * r * r) ** 0.5
        y = TF.conv2d(TF.batch_norm(x, mean, var, gamma, beta, training=False, eps=1e-5).relu(), weight, None, 1, pad)
        gy_wide = torch.randn(n, k + 8, h, w, generator=gen)        # the 3x3 cases read gy through a channel-slice view
        gy = gy_wide[:, 4:4 + k]
        gx_ref, ggamma_ref, gbeta_ref = torch.autograd.grad(y, (x, gamma, beta), gy)
        d = {name: dev(t.detach()) for name, t in dict(wide=wide, mean=mean, inv=(var + 1e-5).rsqrt(), gamma=gamma,
                                                       beta=beta, weight=weight, gy_wide=gy_wide,
                                                       gy=gy.contiguous()).items()}
        if r == 1:
            desc = _lib.ConvDesc(n, c, h, w, k, 1, 1, 1, 1, 0, 0, h, w, total * h * w, 0)
            gy_pointer = d['gy'].data_ptr()
        else:
            desc = _lib.ConvDesc(n, c, h, w, k, 3, 3, 1, 1, 1, 1, h, w, 0, (k + 8) * h * w)
            gy_pointer = d['gy_wide'].data_ptr() + 4 * 4 * h * w
        bn = _lib.BnRelu(d['mean'].data_ptr(), d['inv'].data_ptr(), d['gamma'].data_ptr(), d['beta'].data_ptr())
        assert lib.srgan_conv2d_bnrelu_supported(desc, 1) == 1, (n, c, h, w, k, r)
        old = torch.randn(n, total, h, w, generator=gen)
        for accumulate in ((0, 1) if r == 1 else (0,)):
            gx_wide = dev(old)
            g_gamma, g_beta = torch.full((c,), 0.25, device='cuda'), torch.full((c,), -0.5, device='cuda')
            _lib.check(lib.srgan_conv2d_bwd_data_bnrelu(desc, gy_pointer, d['weight'].data_ptr(), bn,
                                                        d['wide'].data_ptr(), gx_wide.data_ptr(), g_gamma.data_ptr(),
                                                        g_beta.data_ptr(), accumulate, stream), 'bwd_data_bnrelu')
            what = f'fused bn backward {k}->{c} k{r} {h}x{w} accumulate={accumulate}'
            close(gx_wide[:, :c], gx_ref + (old[:, :c] if accumulate else 0.0), what=what + ' gx')
            if total > c:
                close(gx_wide[:, c:], old[:, c:], 0.0, what + ' (channels beyond the view untouched)')
            close(g_gamma, ggamma_ref + 0.25, what=what + ' gamma gradient')
            close(g_beta, gbeta_ref - 0.5, what=what + ' beta gradient')
        gx_wide = dev(old)                          # input gradient only (frozen parameters)
        _lib.check(lib.srgan_conv2d_bwd_data_bnrelu(desc, gy_pointer, d['weight'].data_ptr(), bn,
                                                    d['wide'].data_ptr(), gx_wide.data_ptr(), None, None, 0, stream),
                   'bwd_data_bnrelu')
        close(gx_wide[:, :c], gx_ref, what=f'fused bn backward {k}->{c} k{r} (no parameter gradients)')
    tiny = _lib.ConvDesc(2, 32, 5, 5, 16, 1, 1, 1, 1, 0, 0, 5, 5, 0, 0)       # fewer than 32 pixels: no fused form
    assert lib.srgan_conv2d_bnrelu_supported(tiny, 1) == 0


@gpu
def test_fused_dense_block_matches_primitive_path(F):
    """The concat-free one-node dense block (fused.py) against the layer-by-layer primitive ops: forward,
    input gradient and every parameter gradient, with trainable and with frozen parameters."""
    import srgan_amd  # noqa: F401
    from srgan_amd import fused, nn
    from srgan_amd.crowd.models import _DenseBlock
    from srgan_amd.tape import backward
    torch.manual_seed(3)
    block = _DenseBlock(num_layers=3, num_input_features=8, bn_size=2, growth_rate=4)
    gen = torch.Generator().manual_seed(4)
    for m in block.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=gen) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=gen) * 0.2
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=gen) * 0.2
            m.running_var.data = torch.rand(m.running_var.shape, generator=gen) + 0.5
    arena = nn.flatten_parameters(block, torch.device('cuda', 0))
    x_host = torch.randn(3, 8, 9, 7, generator=gen)
    cotangent = torch.randn(3, 20, 9, 7, generator=gen)
    results = {}
    for enabled in (False, True):
        fused.ENABLED = enabled
        try:
            arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            y = block(x)
            backward(y, grad=F.leaf(dev(cotangent)))
            results[enabled] = (y.cpu(), x.grad.cpu(), arena.grad.detach().cpu().clone())
        finally:
            fused.ENABLED = True
    for i, what in enumerate(('output', 'input gradient', 'parameter gradients')):
        close(results[True][i], results[False][i], 1e-4, what)
    # frozen parameters (generator update): input gradient only, the arena stays untouched
    arena.zero_grad()
    with nn.frozen_parameters(block):
        x = F.leaf(dev(x_host), requires_grad=True)
        y = block(x)
    backward(y, grad=F.leaf(dev(cotangent)))
    close(x.grad, results[False][1], 1e-4, 'input gradient with frozen parameters')
    assert float(arena.grad.abs().sum()) == 0.0
    # one gradient tensor fanned out to two blocks (add's backward hands the SAME var to both inputs) and a gradient that
    # arrives exclusively through another op: the fused backward may accumulate in place only in the second case
    shared = {}
    for enabled in (False, True):
        fused.ENABLED = enabled
        try:
            arena.zero_grad()
            x1, x2 = F.leaf(dev(x_host), requires_grad=True), F.leaf(dev(x_host * 0.5 + 0.1), requires_grad=True)
            total = F.add(block(x1), F.scale(block(x2), 1.5))            # scale's backward makes a fresh gradient
            root = F.sum_all(F.mul(F.add(total, block(x2)), F.leaf(dev(cotangent))))     # a third call shares add's g
            backward(root)
            shared[enabled] = (x1.grad.cpu(), x2.grad.cpu(), arena.grad.detach().cpu().clone())
        finally:
            fused.ENABLED = True
    for i, what in enumerate(('first input gradient', 'second input gradient', 'parameter gradients')):
        close(shared[True][i], shared[False][i], 1e-4, 'shared / exclusive incoming gradients: ' + what)
    # second order (gradient penalty): recorded input gradient, then the parameter gradients of a function of it --
    # the fused node's linearised-forward double backward against the primitive ops (two channel widths so that the
    # block's downstream gradient itself depends on a parameter-carrying recorded op)
    head = F.leaf(dev(torch.randn(20, generator=gen)), requires_grad=True)
    second = {}
    for enabled in (False, True):
        fused.ENABLED = enabled
        try:
            arena.zero_grad()
            head.grad = None
            x = F.leaf(dev(x_host), requires_grad=True)
            y = block(x)
            scalar = F.sum_all(F.mul(F.chan_affine(y, None, head, None, None), F.leaf(dev(cotangent))))
            (gx,) = backward(scalar, inputs=[x], create_graph=True)
            penalty = F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0)))
            backward(penalty)
            second[enabled] = (gx.cpu(), penalty.cpu(), arena.grad.detach().cpu().clone(), head.grad.cpu())
        finally:
            fused.ENABLED = True
    for i, what in enumerate(('recorded input gradient', 'penalty', 'penalty parameter gradients', 'penalty head gradient')):
        close(second[True][i], second[False][i], 1e-4, 'second order: ' + what)
    assert float(second[True][2].abs().max()) > 0.0


@gpu
@pytest.mark.parametrize('plane,batch', [((16, 16), 2), ((14, 14), 3), ((7, 7), 5), ((28, 28), 2), ((9, 7), 2)])
def test_fused_dense_block_with_in_kernel_batch_norm(F, plane, batch):
    """A block geometry that takes the in-kernel batch-norm path (fused.PROLOGUE) against the materialised one:
    forward, first-order gradients with trainable parameters, and the gradient-penalty second order -- on whole and on
    ragged planes (those of 224 x 224), with the block's weight gradients as two grouped launches and one by one."""
    import srgan_amd  # noqa: F401
    from srgan_amd import fused, nn
    from srgan_amd.crowd.models import _DenseBlock
    from srgan_amd.tape import backward
    torch.manual_seed(5)
    block = _DenseBlock(num_layers=3, num_input_features=32, bn_size=4, growth_rate=8)
    gen = torch.Generator().manual_seed(6)
    for m in block.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=gen) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=gen) * 0.2
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=gen) * 0.2
            m.running_var.data = torch.rand(m.running_var.shape, generator=gen) + 0.5
    arena = nn.flatten_parameters(block, torch.device('cuda', 0))
    x_host = torch.randn(batch, 32, *plane, generator=gen)
    cotangent = torch.randn(batch, 56, *plane, generator=gen)
    results = {}
    for prologue in (False, True, 'one by one'):
        fused.PROLOGUE = bool(prologue)
        fused.GROUPED_WGRAD = prologue is True
        try:
            arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            y = block(x)
            backward(y, grad=F.leaf(dev(cotangent)))
            first = (y.cpu(), x.grad.cpu(), arena.grad.detach().cpu().clone())
            arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            y = block(x)
            scalar = F.sum_all(F.mul(y, F.leaf(dev(cotangent))))
            (gx,) = backward(scalar, inputs=[x], create_graph=True)
            penalty = F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0)))
            backward(penalty)
            results[prologue] = first + (gx.cpu(), arena.grad.detach().cpu().clone())
        finally:
            fused.PROLOGUE = fused.GROUPED_WGRAD = True
    for i, what in enumerate(('output', 'input gradient', 'parameter gradients', 'recorded input gradient',
                              'penalty parameter gradients')):
        close(results[True][i], results[False][i], 1e-4, 'in-kernel batch-norm: ' + what)
        close(results['one by one'][i], results[False][i], 1e-4, 'in-kernel batch-norm, weight gradients one by one: ' + what)


@gpu
def test_fused_transition_matches_primitive_path(F):
    """The DenseNet transition's norm -> relu -> conv as one fused node (fused.bn_relu_conv) against the primitive ops:
    forward, first-order gradients (trainable and frozen parameters) and the gradient-penalty second order."""
    import srgan_amd  # noqa: F401
    from srgan_amd import fused, nn
    from srgan_amd.crowd.models import _Transition
    from srgan_amd.tape import backward
    torch.manual_seed(7)
    transition = _Transition(64, 32)
    gen = torch.Generator().manual_seed(8)
    norm = transition.norm
    norm.weight.data = torch.rand(norm.weight.shape, generator=gen) + 0.5
    norm.bias.data = torch.randn(norm.bias.shape, generator=gen) * 0.2
    norm.running_mean.data = torch.randn(norm.running_mean.shape, generator=gen) * 0.2
    norm.running_var.data = torch.rand(norm.running_var.shape, generator=gen) + 0.5
    arena = nn.flatten_parameters(transition, torch.device('cuda', 0))
    x_host = torch.randn(3, 64, 16, 16, generator=gen)
    cotangent = torch.randn(3, 32, 8, 8, generator=gen)
    results = {}
    for enabled in (False, True):
        fused.ENABLED = enabled
        try:
            arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            y = transition(x)
            backward(y, grad=F.leaf(dev(cotangent)))
            first = (y.cpu(), x.grad.cpu(), arena.grad.detach().cpu().clone())
            with nn.frozen_parameters(transition):
                arena.zero_grad()
                x = F.leaf(dev(x_host), requires_grad=True)
                backward(transition(x), grad=F.leaf(dev(cotangent)))
                frozen = (x.grad.cpu(), float(arena.grad.abs().max()))
            arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            scalar = F.sum_all(F.mul(transition(x), F.leaf(dev(cotangent))))
            (gx,) = backward(scalar, inputs=[x], create_graph=True)
            penalty = F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0)))
            backward(penalty)
            results[enabled] = first + (gx.cpu(), arena.grad.detach().cpu().clone()) + frozen
        finally:
            fused.ENABLED = True
    for i, what in enumerate(('output', 'input gradient', 'parameter gradients', 'recorded input gradient',
                              'penalty parameter gradients', 'input gradient with frozen parameters')):
        close(results[True][i], results[False][i], 1e-4, 'fused transition: ' + what)
    assert results[True][6] == 0.0 and float(results[True][4].abs().max()) > 0.0


@gpu
def test_device_side_crowd_patch_batches(F):
    """srgan_crowd_extract_patches / DeviceCrowdPatchLoader against the host-side NumPy transforms
    of the reference pipeline (patch around a centre with zero padding, left-right flip, [-1, 1] normalisation, CHW)."""
    from srgan_amd.crowd.data import CrowdExample, DeviceCrowdPatchLoader, extract_padded_patch, negative_one_to_one
    generator = np.random.RandomState(11)
    scenes = []
    for shape in [(70, 90), (64, 64), (100, 81)]:
        scenes.append(CrowdExample(image=generator.randint(0, 256, size=shape + (3,)).astype(np.uint8),
                                   label=generator.rand(*shape).astype(np.float32),
                                   map_=generator.rand(*shape).astype(np.float32)))
    size = 64
    loader = DeviceCrowdPatchLoader(scenes, batch_size=6, image_patch_size=size, seed=3)
    assert loader.length == 7 * 27 + 1 + 37 * 18
    # explicit draws, including centres whose patch leaves the scene (padding) and both flip states
    draws = [(0, 32, 32, 0), (0, 38, 58, 1), (1, 32, 32, 1), (2, 68, 49, 0), (0, 5, 80, 0), (2, 99, 2, 1)]
    image, label, map_ = loader.batch_for(draws)
    for index, (scene, y, x, flip) in enumerate(draws):
        example = scenes[scene]
        expected_image = negative_one_to_one(extract_padded_patch(example.image, y, x, size))
        expected_label = extract_padded_patch(example.label[:, :, None], y, x, size)[:, :, 0]
        expected_map = extract_padded_patch(example.map[:, :, None], y, x, size)[:, :, 0]
        if flip:
            expected_image, expected_label, expected_map = (np.flip(a, axis=1) for a in
                                                            (expected_image, expected_label, expected_map))
        close(image[index], torch.from_numpy(expected_image.transpose(2, 0, 1).copy()), 1e-6, f'patch image {index}')
        close(label[index], torch.from_numpy(expected_label.copy()), 0.0, f'patch label {index}')
        close(map_[index], torch.from_numpy(expected_map.copy()), 0.0, f'patch map {index}')
    batch = next(iter(loader))                       # random draws stay inside the scenes
    assert tuple(batch[0].shape) == (6, 3, size, size) and tuple(batch[1].shape) == (6, size, size)
    assert float(batch[0].min()) >= -1.0 and float(batch[0].max()) <= 1.0
    for scene, y, x, flip in loader.draw_positions():
        height, width = loader.shapes[scene]
        assert 32 <= y <= height - 32 and 32 <= x <= width - 32 and flip in (0, 1)


@gpu
def test_crowd_offline_labels(F):
    """srgan_crowd_iknn_map / crowd.labels.generate_iknn_map against the reference preprocessor's own output (golden
    g12: generate_knn_map through scikit-learn's ball tree, k = 1..5, a scene with fewer heads than neighbours, and
    the clipped variant) and srgan_crowd_density_label / generate_density_label against its generate_density_label
    (beta 0.1 / 0.3 / 0.5; 40 and 400 heads); float32 on the device against float64 on the host."""
    from helpers import load_golden
    from srgan_amd.crowd.labels import generate_iknn_map
    g = load_golden('g12_crowd_labels')
    for index in range(3):
        heads, shape = g[f'scene{index}/heads_yx'], tuple(int(v) for v in g[f'scene{index}/shape'])
        for k in (1, 2, 3, 4, 5):
            close(generate_iknn_map(heads, shape, number_of_neighbors=k), torch.from_numpy(g[f'scene{index}/i{k}nn_map']),
                  1e-5, f'scene {index} i{k}nn map')
        close(generate_iknn_map(heads, shape, number_of_neighbors=3, upper_bound=6.0),
              torch.from_numpy(g[f'scene{index}/i3nn_map_bounded']), 1e-5, f'scene {index} bounded i3nn map')
    from srgan_amd.crowd.labels import generate_density_label
    for index in range(2):                      # Gaussian density labels: 40 and 400 heads, windows clipped at the bottom / right
        heads, shape = g[f'dscene{index}/heads_yx'], tuple(int(v) for v in g[f'dscene{index}/shape'])
        for beta in (0.1, 0.3, 0.5):
            expected = torch.from_numpy(g[f'dscene{index}/density_beta{beta}'])
            label = generate_density_label(heads, shape, neighbor_deviation_beta=beta, yx_order=True)
            close(label, expected, 1e-4, f'scene {index} density label beta {beta}')
            assert abs(float(label.sum()) - float(expected.sum())) < 1e-3 * float(expected.sum())
    # a scene larger than one chunk of the head list, against a brute-force torch reference
    generator = torch.Generator().manual_seed(5)
    heads = torch.rand(2500, 2, generator=generator) * torch.tensor([95.0, 127.0])
    ys, xs = torch.meshgrid(torch.arange(96.0), torch.arange(128.0), indexing='ij')
    distances = torch.cdist(torch.stack([ys.reshape(-1), xs.reshape(-1)], 1).double(), heads.double())
    expected = 1 / (distances.topk(4, dim=1, largest=False).values.mean(1).reshape(96, 128) + 1)
    close(generate_iknn_map(heads.numpy(), (96, 128), number_of_neighbors=4), expected, 1e-5, 'large scene i4nn map')


@gpu
def test_crowd_density_label_variants(F):
    """Every branch of the reference's generate_density_label (crowd/database_preprocessor.py:113-223) against its own output
    (golden g12b): a perspective map, include_body, ignore_tiny (dropped heads do not count), perspective_resizing=False,
    force_full_image_count_normalize=False, body parts without a perspective map, and the (x, y) position order."""
    from helpers import load_golden
    from srgan_amd.crowd.labels import generate_density_label
    g = load_golden('g12b_crowd_label_variants')
    for index in range(2):
        heads, shape = g[f'scene{index}/heads_yx'], tuple(int(v) for v in g[f'scene{index}/shape'])
        perspective = g[f'scene{index}/perspective_map']
        variants = {
            'perspective': dict(perspective=perspective),
            'perspective_body': dict(perspective=perspective, include_body=True),
            'perspective_tiny': dict(perspective=perspective, ignore_tiny=True),
            'perspective_body_tiny': dict(perspective=perspective, include_body=True, ignore_tiny=True),
            'perspective_body_unnormalized': dict(perspective=perspective, include_body=True,
                                                  force_full_image_count_normalize=False),
            'body_without_perspective': dict(include_body=True, neighbor_deviation_beta=0.3),
            'fixed_sigma': dict(perspective_resizing=False),
            'fixed_sigma_unnormalized': dict(perspective_resizing=False, force_full_image_count_normalize=False),
        }
        for name, arguments in variants.items():
            expected = torch.from_numpy(g[f'scene{index}/{name}'])
            label = generate_density_label(heads, shape, yx_order=True, **arguments)
            close(label, expected, 1e-4, f'scene {index} {name}')
            assert abs(float(label.sum()) - float(expected.sum())) <= 1e-3 * float(expected.sum()), name
        label = generate_density_label(heads[:, ::-1], shape, perspective=perspective, include_body=True)     # the reference's default order
        close(label, torch.from_numpy(g[f'scene{index}/perspective_body_xy']), 1e-4, f'scene {index} (x, y) order')
        counted = float(generate_density_label(heads, shape, perspective=perspective, ignore_tiny=True, yx_order=True).sum())
        with pytest.raises(ValueError):
            generate_density_label(heads, shape, perspective=perspective, include_body=True, perspective_resizing=False, yx_order=True)
        assert abs(counted - (len(heads) - int(g[f'scene{index}/tiny_heads']))) < 1e-2          # ignored heads are not counted


@gpu
@pytest.mark.parametrize('shape', [(2, 8, 16, 16), (3, 5, 10, 12), (2, 64, 56, 56)])
def test_fused_stem_norm_relu_pool(F, shape):
    """functional.bn_relu_max_pool2d (norm0 -> relu0 -> pool0 in one pass each way) against the two-op form: output,
    first-order gradients of x / gamma / beta, and the recorded (gradient-penalty) route through it."""
    from srgan_amd.tape import backward
    gen = torch.Generator().manual_seed(41)
    n, c, h, w = shape
    x_host = torch.randn(n, c, h, w, generator=gen)
    stats = dict(mean=torch.randn(c, generator=gen) * 0.3, inv=(torch.rand(c, generator=gen) + 0.5).rsqrt(),
                 gamma=torch.rand(c, generator=gen) + 0.5, beta=torch.randn(c, generator=gen) * 0.3)
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    cotangent = torch.randn(n, c, oh, ow, generator=gen)
    results = {}
    for fused_form in (False, True):
        x = F.leaf(dev(x_host), requires_grad=True)
        gamma, beta = F.leaf(dev(stats['gamma']), requires_grad=True), F.leaf(dev(stats['beta']), requires_grad=True)
        mean, inv = F.constant(dev(stats['mean'])), F.constant(dev(stats['inv']))
        def forward(x):
            if fused_form:
                y = F.bn_relu_max_pool2d(x, mean, inv, gamma, beta, 3, 2, 1)
                assert y is not None
                return y
            return F.max_pool2d(F.batch_norm_eval(x, mean, inv, gamma, beta, relu=True), 3, 2, 1)
        y = forward(x)
        backward(y, grad=F.leaf(dev(cotangent)))
        first = (y.cpu(), x.grad.cpu(), gamma.grad.cpu(), beta.grad.cpu())
        # second order: a function of the recorded input gradient, back to gamma and x
        x2 = F.leaf(dev(x_host), requires_grad=True)
        gamma.grad = beta.grad = None
        scalar = F.sum_all(F.mul(forward(x2), F.leaf(dev(cotangent))))
        (gx,) = backward(scalar, inputs=[x2], create_graph=True)
        penalty = F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0)))
        backward(penalty)
        results[fused_form] = first + (gx.cpu(), gamma.grad.cpu())
    for i, what in enumerate(('output', 'input gradient', 'gamma gradient', 'beta gradient', 'recorded input gradient',
                              'penalty gamma gradient')):
        close(results[True][i], results[False][i], 1e-5, 'fused stem pool: ' + what)


@gpu
@pytest.mark.parametrize('shape', [(2, 8, 16, 16), (3, 5, 10, 12), (2, 64, 56, 56), (2, 6, 6, 14)])
def test_fused_transition_norm_relu_avg_pool(F, shape):
    """functional.bn_relu_avg_pool2d (the transitions' norm -> relu -> pool in one pass each way, pooling BEFORE the 1x1
    convolution) against the two-op form: output, first-order gradients of x / gamma / beta, and the recorded
    (gradient-penalty) route through it.  (2, 6, 6, 14): a width the fused kernels do not have -> None."""
    from srgan_amd.tape import backward
    gen = torch.Generator().manual_seed(43)
    n, c, h, w = shape
    x_host = torch.randn(n, c, h, w, generator=gen)
    stats = dict(mean=torch.randn(c, generator=gen) * 0.3, inv=(torch.rand(c, generator=gen) + 0.5).rsqrt(),
                 gamma=torch.rand(c, generator=gen) + 0.5, beta=torch.randn(c, generator=gen) * 0.3)
    cotangent = torch.randn(n, c, h // 2, w // 2, generator=gen)
    if w % 4:
        x = F.leaf(dev(x_host))
        assert F.bn_relu_avg_pool2d(x, F.constant(dev(stats['mean'])), F.constant(dev(stats['inv'])),
                                    F.leaf(dev(stats['gamma'])), F.leaf(dev(stats['beta']))) is None
        return
    results = {}
    for fused_form in (False, True):
        x = F.leaf(dev(x_host), requires_grad=True)
        gamma, beta = F.leaf(dev(stats['gamma']), requires_grad=True), F.leaf(dev(stats['beta']), requires_grad=True)
        mean, inv = F.constant(dev(stats['mean'])), F.constant(dev(stats['inv']))

        def forward(x):
            if fused_form:
                y = F.bn_relu_avg_pool2d(x, mean, inv, gamma, beta)
                assert y is not None
                return y
            return F.avg_pool2d(F.batch_norm_eval(x, mean, inv, gamma, beta, relu=True), 2, 2)
        y = forward(x)
        backward(y, grad=F.leaf(dev(cotangent)))
        first = (y.cpu(), x.grad.cpu(), gamma.grad.cpu(), beta.grad.cpu())
        x2 = F.leaf(dev(x_host), requires_grad=True)
        gamma.grad = beta.grad = None
        scalar = F.sum_all(F.mul(forward(x2), F.leaf(dev(cotangent))))
        (gx,) = backward(scalar, inputs=[x2], create_graph=True)
        penalty = F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0)))
        backward(penalty)
        results[fused_form] = first + (gx.cpu(), gamma.grad.cpu())
    for i, what in enumerate(('output', 'input gradient', 'gamma gradient', 'beta gradient', 'recorded input gradient',
                              'penalty gamma gradient')):
        close(results[True][i], results[False][i], 1e-5, 'fused transition pool: ' + what)


@gpu
def test_transition_with_the_pooling_first_equals_the_reference_order(F):
    """crowd.models._Transition: norm -> relu -> pool -> conv (fused.POOL_FIRST) against the reference's norm -> relu -> conv
    -> pool (crowd/models.py:364-371) -- output, input gradient, all parameter gradients, and the recorded route."""
    from srgan_amd import fused, nn
    from srgan_amd.crowd.models import _Transition
    from srgan_amd.tape import backward
    gen = torch.Generator().manual_seed(44)
    torch.manual_seed(44)
    module = _Transition(24, 12)
    with torch.no_grad():
        module.norm.running_mean.copy_(torch.randn(24, generator=gen) * 0.3)
        module.norm.running_var.copy_(torch.rand(24, generator=gen) + 0.5)
        module.norm.weight.copy_(torch.rand(24, generator=gen) + 0.5)
        module.norm.bias.copy_(torch.randn(24, generator=gen) * 0.3)
    nn.flatten_parameters(module, torch.device('cuda'))
    x_host = torch.randn(3, 24, 16, 24, generator=gen)
    cotangent = torch.randn(3, 12, 8, 12, generator=gen)
    saved, results = fused.POOL_FIRST, {}
    try:
        for pool_first in (False, True):
            fused.POOL_FIRST = pool_first
            module._srgan_arena.zero_grad()
            x = F.leaf(dev(x_host), requires_grad=True)
            y = module(x)
            backward(y, grad=F.leaf(dev(cotangent)))
            first = (y.cpu(), x.grad.cpu(), module._srgan_arena.grad.detach().cpu().clone())
            module._srgan_arena.zero_grad()
            x2 = F.leaf(dev(x_host), requires_grad=True)
            scalar = F.sum_all(F.mul(module(x2), F.leaf(dev(cotangent))))
            (gx,) = backward(scalar, inputs=[x2], create_graph=True)
            backward(F.mean_all(F.square(F.add_scalar(F.row_norm(F.flatten2d(gx)), -1.0))))
            results[pool_first] = first + (gx.cpu(), module._srgan_arena.grad.detach().cpu().clone())
    finally:
        fused.POOL_FIRST = saved
    for i, what in enumerate(('output', 'input gradient', 'parameter gradients', 'recorded input gradient',
                              'penalty parameter gradients')):
        close(results[True][i], results[False][i], 2e-5, 'pooling first: ' + what)
