"""The 16-bit data path (``blocked16``: bf16 / fp16 tensors in the blocked layout, fused activations) on the GPU.

Every kernel of ``csrc/blocked16*.hip`` is held against the fp32 kernels of the parity path (which the other GPU test files
pin against the reference-generated goldens) on operands that bf16 / fp16 represent EXACTLY (small integers): the fp32
result, rounded once to the 16-bit type, must then be reproduced bit for bit -- this pins the blocked layout, the packed
weight shadows, the transpose-read fragments of the weight gradients, the epilogues (bias + activation, mask by reference) and
the tie rule of the pool independently of rounding.  Then whole VGG passes (first and second order) against the fp32-storage
path with the rounding-sized tolerances stated in the tests."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ['bf16', 'f16']
TORCH = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}


@pytest.fixture(scope='module')
def F():
    import srgan_amd  # noqa: F401
    from srgan_amd import functional
    assert torch.cuda.is_available()
    return functional


@pytest.fixture(scope='module')
def B():
    import srgan_amd  # noqa: F401
    from srgan_amd import blocked16
    return blocked16


def integers(shape, low, high, seed):
    generator = torch.Generator().manual_seed(seed)
    return torch.randint(low, high + 1, shape, generator=generator).float().cuda()


def rounded(t, mode):
    return t.to(TORCH[mode]).float()


def blocked(F, B, t, mode):
    return B.pack(F.leaf(t), B.CODES[mode])


def nchw(B, var):
    meta = var.meta
    return B.unpack(var, (meta.n, meta.c, meta.h, meta.w)).data


# planes: 64-wide and ragged 40-wide (32-column tiles), 16 x 16, 8 x 8, 4 x 4 (whole images side by side), 14 x 14 and 7 x 7
# (guards of the 16-wide tile); channel tails on both sides; few / many images
CONVS = [(2, 64, 64, 64, 64), (3, 24, 12, 40, 40), (2, 3, 64, 64, 64), (5, 128, 16, 16, 72), (19, 40, 4, 4, 72), (9, 32, 8, 8, 200),
         (32, 256, 4, 4, 128), (3, 48, 14, 14, 16), (2, 16, 7, 7, 8), (4, 64, 32, 32, 128), (130, 16, 8, 8, 64), (2, 64, 20, 16, 3),
         (3, 40, 16, 16, 136)]


@pytest.mark.parametrize('tile', ['1', '2', '4'])
@pytest.mark.parametrize('case', [(2, 64, 64, 64, 64), (5, 128, 16, 16, 72), (9, 32, 8, 8, 200), (3, 24, 12, 40, 40), (19, 40, 4, 4, 72),
                                  (3, 72, 20, 16, 136), (2, 8, 33, 70, 64)])
def test_conv3x3_is_exact_with_every_pixel_tile(F, B, monkeypatch, tile, case):
    """The 128- / 256- / 512-pixel tiles of the 3x3 kernels (the plan picks by workgroup count; forced here).  The staging
    variant -- LDS-DMA ring of 2 / 3 stages or registers -- is read once per process (SRGAN_H_DMA_RING): the default policy
    runs here (a ring of 2; registers where K is split or the tile has 32 rows),
    `test_conv3x3_staging_variants_agree` runs the others in child processes."""
    monkeypatch.setenv('SRGAN_H_CONV_NI', tile)
    test_conv3x3_forward_data_gradient_and_weight_gradient_are_exact_on_integers(F, B, 'bf16', case)


@pytest.mark.parametrize('ring', ['0', '2', '3'])
def test_conv3x3_staging_variants_agree(ring):
    """The exactness tests again with the staging variant forced for every launch (a child process each: the switch is read
    once)."""
    import subprocess
    environment = dict(os.environ, SRGAN_H_DMA_RING=ring)
    completed = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k',
                                'every_pixel_tile or exact_on_integers'], env=environment, capture_output=True, text=True,
                               cwd=ROOT, timeout=900)
    assert completed.returncode == 0, completed.stdout[-3000:] + completed.stderr[-2000:]


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('case', CONVS)
def test_conv3x3_forward_data_gradient_and_weight_gradient_are_exact_on_integers(F, B, mode, case):
    n, c, h, w, k = case
    layer = torch.nn.Conv2d(c, k, 3, padding=1).cuda()
    with torch.no_grad():
        layer.weight.copy_(integers((k, c, 3, 3), -1, 1, 1))
        layer.bias.copy_(integers((k,), -3, 3, 2))
    x = integers((n, c, h, w), -2, 2, 3)
    from srgan_amd.tape import no_grad
    with no_grad():
        xb = blocked(F, B, x, mode)
        # forward: relu(conv + bias), fp32 result rounded once
        got = nchw(B, B.conv3x3(xb, layer, slope=0.0))
        want = F.relu(F.conv2d(F.leaf(x), F.leaf(layer.weight.data), F.leaf(layer.bias.data), 1, 1)).data
        assert torch.equal(got, rounded(want, mode))
        # leaky and plain epilogues
        got = nchw(B, B.conv3x3(xb, layer, slope=0.25))
        plain = F.conv2d(F.leaf(x), F.leaf(layer.weight.data), F.leaf(layer.bias.data), 1, 1).data
        assert torch.equal(got, rounded(torch.where(plain > 0, plain, plain * 0.25), mode))
        assert torch.equal(nchw(B, B.conv3x3(xb, layer)), rounded(plain, mode))
        # data gradient with the mask epilogue: convT(s) * (ref > 0 ? 1 : slope), ref = an activated tensor of x's shape
        s = integers((n, k, h, w), -2, 2, 4)
        ref = integers((n, c, h, w), -1, 1, 5)
        sb, refb = blocked(F, B, s, mode), blocked(F, B, ref, mode)
        shadow = B.shadow_of(layer, 'conv3x3', B.CODES[mode])
        got = nchw(B, B._layer(sb, layer, shadow, True, 2, 0.5, refb.data, False))
        plain = F.conv2d_backward_data(F.leaf(s), F.leaf(layer.weight.data), (n, c, h, w), (1, 1), (1, 1)).data
        assert torch.equal(got, rounded(plain * torch.where(ref > 0, 1.0, 0.5), mode))
        assert torch.equal(nchw(B, B._layer(sb, layer, shadow, True, 0, 1.0, None, False)), rounded(plain, mode))
        # weight gradient (fp32, accumulated into): exact integers
        into = torch.full((k, c, 3, 3), 7.0, device='cuda')
        B._weight_gradient(shadow, layer, xb, sb, into)
        want = F.conv2d_backward_weight(F.leaf(x), F.leaf(s), (k, c, 3, 3), (1, 1), (1, 1)).data + 7.0
        assert torch.equal(into, want)
        # bias gradient
        sums = torch.full((k,), 1.0, device='cuda')
        from srgan_amd import _lib
        _lib.check(_lib.library().srgan_h_channel_sums(sb.data.data_ptr(), sums.data_ptr(), n, k, h * w, B.CODES[mode], F._stream()),
                   'srgan_h_channel_sums')
        assert torch.equal(sums, s.sum(dim=(0, 2, 3)) + 1.0)


@pytest.mark.parametrize('mode', MODES)
def test_pack_unpack_add_and_masks(F, B, mode):
    x = integers((3, 13, 6, 10), -200, 200, 1) / 8.0
    ref = integers((3, 13, 6, 10), -1, 1, 2)
    from srgan_amd.tape import no_grad
    with no_grad():
        xb, refb = blocked(F, B, x, mode), blocked(F, B, ref, mode)
        assert xb.data.shape == (3, 2, 6, 10, 8) and xb.data.dtype == TORCH[mode]
        assert torch.equal(nchw(B, xb), x)
        assert torch.all(xb.data[:, 1, :, :, 5:] == 0)                     # channels 13..15 of the last group are zeros
        masked = B.pack(F.leaf(x), B.CODES[mode], refb.data, 0.125)
        assert torch.equal(nchw(B, masked), x * torch.where(ref > 0, 1.0, 0.125))
        assert torch.equal(nchw(B, B.add(xb, refb)), rounded(x + ref, mode))
        matrix = integers((5, 20), -9, 9, 3)
        mb = blocked(F, B, matrix, mode)
        assert mb.data.shape == (5, 3, 1, 1, 8) and torch.equal(B.unpack(mb).data, matrix)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('shape', [(3, 24, 8, 12), (2, 5, 2, 2), (4, 64, 16, 16)])
def test_max_pool_forward_backward_and_gather_follow_torchs_first_maximum(F, B, mode, shape):
    n, c, h, w = shape
    x = torch.relu(integers(shape, -2, 2, 1))              # many ties (zeros and equal positives), as behind a ReLU in bf16
    g = integers((n, c, h // 2, w // 2), -3, 3, 2)
    tangent = integers(shape, -3, 3, 3)
    reference = torch.nn.functional.max_pool2d(x.cpu().requires_grad_(True), 2, 2, return_indices=True)
    pooled_cpu, indices = reference
    from srgan_amd.tape import no_grad
    with no_grad():
        xb = blocked(F, B, x, mode)
        xb.meta.mask_ref, xb.meta.slope = xb.data, 0.0           # an activated (ReLU) tensor
        pooled = B.max_pool2(xb)
        assert torch.equal(nchw(B, pooled), pooled_cpu.detach().cuda())
        gb = blocked(F, B, g, mode)
        gx = nchw(B, B._pool_backward(xb.data, xb.meta, gb))
        want = torch.zeros(n, c, h * w)
        want.scatter_(2, indices.view(n, c, -1), g.cpu().view(n, c, -1))
        want = want.view(shape).cuda() * (x > 0)                 # ... times the ReLU's derivative
        assert torch.equal(gx, want)
        gathered = nchw(B, B._pool_gather(xb.data, xb.meta, blocked(F, B, tangent, mode)))
        want = tangent.cpu().view(n, c, -1).gather(2, indices.view(n, c, -1)).view(n, c, h // 2, w // 2).cuda()
        assert torch.equal(gathered, want)


LINEARS = [(7, 40, 2, 2, 24), (130, 64, 1, 1, 200), (33, 16, 4, 12, 1), (384, 512, 2, 2, 4096 // 16)]


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('case', LINEARS)
def test_linear_layers_behind_a_flattened_plane_are_exact_on_integers(F, B, mode, case):
    n, c, h, w, outputs = case
    inputs = c * h * w
    layer = torch.nn.Linear(inputs, outputs).cuda()
    with torch.no_grad():
        layer.weight.copy_(integers((outputs, inputs), -1, 1, 1))
        layer.bias.copy_(integers((outputs,), -3, 3, 2))
    x = integers((n, c, h, w), -1, 1, 3)
    s = integers((n, outputs), -2, 2, 4)
    from srgan_amd.tape import no_grad
    with no_grad():
        flat = B.flatten(blocked(F, B, x, mode))
        want = torch.nn.functional.linear(x.view(n, -1), layer.weight.data, layer.bias.data)
        assert torch.equal(B.unpack(B.linear(flat, layer, slope=0.0)).data, rounded(torch.relu(want), mode))
        assert torch.equal(B.unpack(B.linear(flat, layer)).data, rounded(want, mode))
        # data gradient (blocked order, viewed back to the plane) and weight gradient (in the layer's own layout)
        sb = blocked(F, B, s, mode)
        shadow = B.shadow_of(layer, 'linear', B.CODES[mode], in_blocked=flat.meta.c, plane=flat.meta.plane)
        gx = B._layer(sb, layer, shadow, True, 0, 1.0, None, False)
        back = gx.data.view(n, (c + 7) // 8, h, w, 8)
        unblocked = back.permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, :c].float()
        assert torch.equal(unblocked, rounded((s @ layer.weight.data).view(n, c, h, w), mode))
        into = torch.full_like(layer.weight.data, 3.0)
        B._weight_gradient(shadow, layer, flat, sb, into)
        assert torch.equal(into, s.t() @ x.view(n, -1) + 3.0)


def _vgg(size, scale):
    import srgan_amd  # noqa: F401
    from srgan_amd.age import vgg
    from srgan_amd import nn
    torch.manual_seed(3)
    model = vgg.vgg16(num_classes=1, image_size=size)
    with torch.no_grad():
        for module in model.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear)):
                module.weight.mul_(scale)
                if module.bias is not None:
                    module.bias.normal_(0, 0.05)
    nn.flatten_parameters(model, torch.device('cuda'))
    return model


def _penalty_step(F, model, x, storage):
    """Feature loss + gradient penalty of the interpolates through one VGG: first-order and double backward (reference
    srgan.py:360-375) -> (loss, penalty, gradient norms, the arena's gradient)."""
    from srgan_amd.tape import backward
    arena = model._srgan_arena
    arena.zero_grad()
    with F.compute_dtype('bf16'), F.storage_dtype('bf16' if storage else None):
        scores = model(F.leaf(x))
        loss = F.add(F.mean_all(F.square(scores)), F.mean_all(F.abs_(model.features)))
        backward(loss)
        interpolates = F.leaf(x * 0.5, requires_grad=True)
        model(interpolates)
        norms = F.row_norm(F.flatten2d(model.features))
        gradients, = backward(norms, grad=F.full_like(norms, 1.0), inputs=[interpolates], create_graph=True)
        gradient_norm = F.row_norm(F.flatten2d(gradients))
        penalty = F.mean_all(F.square(F.relu(F.add_scalar(gradient_norm, -1.0))))
        backward(penalty)
    torch.cuda.synchronize()
    return loss.item(), penalty.item(), gradient_norm.data.clone(), arena.grad.clone()


@pytest.mark.parametrize('size', [32, 64])
def test_vgg_first_and_second_order_match_the_fp32_storage_path(F, size):
    """The whole 16-bit graph (13 fused convolutions, 5 pools, 3 linear layers; the recorded backward w.r.t. the images and
    its double backward) against the same bf16-operand arithmetic on fp32 tensors: the two differ by the rounding of the
    STORED activations / gradients only (2^-9 relative each), stated bounds below."""
    model = _vgg(size, 1.6)
    generator = torch.Generator().manual_seed(5)
    x = (torch.rand(6, 3, size, size, generator=generator) * 2 - 1).cuda()
    loss_a, penalty_a, norms_a, grad_a = _penalty_step(F, model, x, storage=False)
    loss_b, penalty_b, norms_b, grad_b = _penalty_step(F, model, x, storage=True)
    assert penalty_a > 1e-3, 'the penalty must be active for the double backward to be tested'
    assert abs(loss_a - loss_b) <= 2e-2 * abs(loss_a)
    assert torch.allclose(norms_a, norms_b, rtol=3e-2)
    assert abs(penalty_a - penalty_b) <= 6e-2 * abs(penalty_a)
    scale = grad_a.abs().max().item()
    assert scale > 0 and (grad_a - grad_b).abs().max().item() <= 4e-2 * scale
    # per layer: the cosine between the two gradients of every weight tensor
    arena = model._srgan_arena
    for parameter, offset, count in zip(arena.parameters, arena.offsets, arena.sizes):
        a, b = grad_a[offset:offset + count], grad_b[offset:offset + count]
        if a.norm() > 0:
            cosine = torch.dot(a, b) / (a.norm() * b.norm())
            assert cosine > 0.98, (tuple(parameter.shape), cosine.item())


def test_the_16_bit_path_is_bit_reproducible(F):
    model = _vgg(32, 1.6)
    generator = torch.Generator().manual_seed(6)
    x = (torch.rand(5, 3, 32, 32, generator=generator) * 2 - 1).cuda()
    first = _penalty_step(F, model, x, storage=True)
    second = _penalty_step(F, model, x, storage=True)
    assert first[0] == second[0] and first[1] == second[1]
    assert torch.equal(first[2], second[2]) and torch.equal(first[3], second[3])


def test_shadows_follow_the_optimizer(F, B):
    from srgan_amd import optim
    model = _vgg(32, 1.0)
    arena = model._srgan_arena
    optimizer = optim.Adam(arena, lr=1e-2)
    x = (torch.rand(4, 3, 32, 32) * 2 - 1).cuda()
    from srgan_amd.tape import backward, no_grad
    with F.compute_dtype('bf16'), F.storage_dtype('bf16'):
        before = model(F.leaf(x))
        arena.zero_grad()
        backward(F.mean_all(F.square(before)))
        optimizer.step()                                         # raw-pointer update: the shadows are re-rounded behind it
        with no_grad():
            after = model(F.leaf(x))
    with F.compute_dtype('bf16'), no_grad():
        want = model(F.leaf(x))                                  # fp32 tensors, bf16 operands: reads the masters
    assert not torch.equal(before.data, after.data)
    assert torch.allclose(after.data, want.data, rtol=3e-2, atol=3e-2 * want.data.abs().max().item())


def test_batched_shadow_refresh_equals_the_single_layer_packers(F, B, monkeypatch):
    """``refresh(arena)`` re-rounds every convolution shadow of a network in ONE launch (srgan_h_pack_batched) from a device
    job table; the result must be the single-layer packers' bit for bit -- 3x3 shadows in bf16 (VGG) and the 4x4 / stride 2
    shadows in fp16 and in blocked fp32 (the DCGAN discriminator: down and the four up classes)."""
    from srgan_amd import nn
    from srgan_amd.age.models import Discriminator
    from srgan_amd.utility import seed_all
    seed_all(3)
    vgg = _vgg(32, 1.0)
    dcgan = Discriminator(image_size=64)
    nn.flatten_parameters(dcgan, torch.device('cuda'))
    with F.compute_dtype('bf16'), F.storage_dtype('bf16'):
        vgg(F.leaf((torch.rand(2, 3, 32, 32) * 2 - 1).cuda()))
    x = F.leaf((torch.rand(2, 3, 64, 64) * 2 - 1).cuda())
    with F.compute_dtype('f16'), F.storage_dtype('f16'):
        dcgan(x)
    with F.compute_dtype('f32'), F.storage_dtype('f32b'):
        dcgan(x)
    checked = 0
    for model in (vgg, dcgan):
        arena = model._srgan_arena
        shadows = [s for s in arena.shadows if s.kind in B.BATCHED_KINDS]
        assert shadows
        with torch.no_grad():
            arena.data.mul_(1.37).add_(0.01)
        forms = ('forward', 'transposed', 'down', 'up')
        monkeypatch.setenv('SRGAN_H_NO_BATCHED_PACK', '1')
        B.refresh(arena)
        single = [[getattr(s, name).clone() for name in forms if getattr(s, name) is not None] for s in shadows]
        for s in shadows:
            for name in forms:
                if getattr(s, name) is not None:
                    getattr(s, name).fill_(-1)
        monkeypatch.delenv('SRGAN_H_NO_BATCHED_PACK')
        B.refresh(arena)
        assert arena._pack_plan.count >= 2 * len(shadows)
        for s, want in zip(shadows, single):
            got = [getattr(s, name) for name in forms if getattr(s, name) is not None]
            assert len(got) == len(want) == 2
            for a, b in zip(got, want):
                assert torch.equal(a, b), (s.kind, s.code)
                checked += 1
    assert checked >= 2 * (13 + 2 * 4)


@pytest.mark.parametrize('batch', [8, 128])
def test_age_vgg_step_on_bf16_storage_against_the_fp32_oracle(monkeypatch, batch):
    """BASELINE.json configs[1] on the 16-bit data path (bf16 activations / gradients / weight shadows in the blocked layout,
    the gradient-penalty chain included) against the fp32 CPU oracle: five losses within 5e-2 and the post-Adam weights, at a
    small batch and at the configuration's own batch of 128."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_mixed_precision_gpu import _step_against_fp32_oracle
    import srgan_amd.age.srgan as age
    from oracle import models as OM
    monkeypatch.setattr(age, 'model_architecture', 'vgg')

    def configure(experiment):
        experiment.image_size = 64
    _step_against_fp32_oracle(age.AgeExperiment, configure,
                              lambda: (OM.DCGANGenerator(image_size=64), OM.VGG16(1, 64), OM.VGG16(1, 64)),
                              size=64, batch=batch, d_scale=1.3,
                              settings_overrides=dict(compute_dtype='bf16', gradient_penalty_dtype='bf16', storage_dtype='bf16'),
                              tolerance=5e-2)


# (n, small-plane channels A, big-plane channels B, big plane H x W): the DCGAN stages on driving frames (64 x 192 down to 4 x 12) and
# on square images, channel tails, a ragged small plane (6 x 10), three big-plane channels (the images)
K4S2 = [(2, 64, 3, 64, 192), (3, 128, 64, 32, 96), (5, 72, 40, 16, 48), (9, 136, 24, 8, 24), (4, 24, 20, 12, 20), (6, 64, 3, 32, 32),
        (33, 40, 16, 8, 8), (130, 32, 8, 4, 4)]


@pytest.mark.parametrize('mode', MODES + ['f32'])
@pytest.mark.parametrize('case', K4S2)
def test_4x4_stride_2_family_is_exact_on_integers(F, B, mode, case):
    """conv2d / conv_transpose2d 4x4 / s2 / p1 (reference age/models.py:37-51,61-73): forward with the fused bias + leaky
    epilogue, both data gradients with the mask epilogue, the weight gradient -- against the fp32 kernels, bit for bit.  Mode
    'f32' = fp32 tensors in the blocked layout (four channels per slot, v_mfma_f32_32x32x2_f32): the exact form."""
    n, a, b, h, w = case
    code = B.CODES[mode]
    conv = torch.nn.Conv2d(b, a, 4, 2, 1).cuda()
    deconv = torch.nn.ConvTranspose2d(a, b, 4, 2, 1).cuda()
    with torch.no_grad():
        for layer in (conv, deconv):
            layer.weight.copy_(integers(tuple(layer.weight.shape), -1, 1, 1))
            layer.bias.copy_(integers(tuple(layer.bias.shape), -3, 3, 2))
    big = integers((n, b, h, w), -2, 2, 3)
    small = integers((n, a, h // 2, w // 2), -2, 2, 4)
    leaky = lambda t, slope: torch.where(t > 0, t, t * slope)
    from srgan_amd.tape import no_grad
    with no_grad():
        bigb, smallb = blocked(F, B, big, mode), blocked(F, B, small, mode)
        # strided convolution: forward, its data gradient (masked), its weight gradient
        want = F.conv2d(F.leaf(big), F.leaf(conv.weight.data), F.leaf(conv.bias.data), 2, 1).data
        assert torch.equal(nchw(B, B.conv4x4s2(bigb, conv, slope=0.25)), rounded(leaky(want, 0.25), mode))
        assert torch.equal(nchw(B, B.conv4x4s2(bigb, conv)), rounded(want, mode))
        shadow = B.shadow_of(conv, 'k4s2', code)
        ref = integers((n, b, h, w), -1, 1, 5)
        refb = blocked(F, B, ref, mode)
        want = F.conv2d_backward_data(F.leaf(small), F.leaf(conv.weight.data), (n, b, h, w), (2, 2), (1, 1)).data
        assert torch.equal(nchw(B, B._layer(smallb, conv, shadow, True, 2, 0.5, refb.data, False)), rounded(want * torch.where(ref > 0, 1.0, 0.5), mode))
        into = torch.full_like(conv.weight.data, 5.0)
        B._weight_gradient(shadow, conv, bigb, smallb, into)
        assert torch.equal(into, F.conv2d_backward_weight(F.leaf(big), F.leaf(small), tuple(conv.weight.shape), (2, 2), (1, 1)).data + 5.0)
        # transposed convolution: forward, its data gradient (masked), its weight gradient
        want = F.conv_transpose2d(F.leaf(small), F.leaf(deconv.weight.data), F.leaf(deconv.bias.data), 2, 1).data
        assert torch.equal(nchw(B, B.conv_transpose4x4s2(smallb, deconv, slope=0.25)), rounded(leaky(want, 0.25), mode))
        shadow_t = B.shadow_of(deconv, 'k4s2', code)
        ref_small = integers((n, a, h // 2, w // 2), -1, 1, 6)
        want = F.conv2d(F.leaf(big), F.leaf(deconv.weight.data), None, 2, 1).data          # d convT / d input = the strided conv
        got = nchw(B, B._layer(bigb, deconv, shadow_t, True, 2, 0.5, blocked(F, B, ref_small, mode).data, False))
        assert torch.equal(got, rounded(want * torch.where(ref_small > 0, 1.0, 0.5), mode))
        into = torch.full_like(deconv.weight.data, -2.0)
        B._weight_gradient(shadow_t, deconv, smallb, bigb, into)
        want = F.conv2d_backward_weight(F.leaf(big), F.leaf(small), tuple(deconv.weight.shape), (2, 2), (1, 1)).data
        assert torch.equal(into, want - 2.0)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('case', [(7, 256, 512, 4, 12), (130, 40, 24, 4, 4), (3, 16, 8, 2, 3)])
def test_seed_transposed_convolution_is_exact_on_integers(F, B, mode, case):
    """The generator's ``fc``: conv_transpose2d of a 1 x 1 code with a full-plane kernel (reference age/models.py:37,47)."""
    n, c_in, c_out, r, s = case
    layer = torch.nn.ConvTranspose2d(c_in, c_out, (r, s), 1, 0).cuda()
    with torch.no_grad():
        layer.weight.copy_(integers(tuple(layer.weight.shape), -1, 1, 1))
        layer.bias.copy_(integers((c_out,), -3, 3, 2))
    z = integers((n, c_in), -2, 2, 3)
    g = integers((n, c_out, r, s), -2, 2, 4)
    from srgan_amd.tape import no_grad
    with no_grad():
        zb, gb = blocked(F, B, z, mode), blocked(F, B, g, mode)
        want = torch.nn.functional.conv_transpose2d(z.view(n, c_in, 1, 1), layer.weight.data, layer.bias.data)
        assert torch.equal(nchw(B, B.seed_conv_transpose(zb, layer)), rounded(want, mode))
        shadow = B.shadow_of(layer, 'linear_t', B.CODES[mode])
        back = B._layer(gb, layer, shadow, True, 0, 1.0, None, False)
        assert torch.equal(B.unpack(back).data, rounded(torch.einsum('ncrs,kcrs->nk', g, layer.weight.data), mode))
        into = torch.full_like(layer.weight.data, 1.0)
        B._weight_gradient(shadow, layer, zb, gb, into)
        assert torch.equal(into, torch.einsum('nk,ncrs->kcrs', z, g) + 1.0)


def _dcgan_pass(F, size, storage, mode, second_order, conv_dim=32):
    """One DCGAN generator -> discriminator pass with a backward into both (and, optionally, the penalty's double backward
    through the discriminator) -> losses and the two gradient arenas."""
    import srgan_amd  # noqa: F401
    from srgan_amd.age import models
    from srgan_amd import nn
    from srgan_amd.tape import backward
    torch.manual_seed(2)
    G, D = models.Generator(image_size=size, conv_dim=conv_dim), models.Discriminator(image_size=size, conv_dim=conv_dim)
    with torch.no_grad():
        for module in D.modules():
            if isinstance(module, torch.nn.Conv2d):
                module.weight.mul_(2.5)
    for network in (G, D):
        nn.flatten_parameters(network, torch.device('cuda'))
    z = torch.randn(6, 256, generator=torch.Generator().manual_seed(3)).cuda()
    with F.compute_dtype('f32' if mode == 'f32b' else mode), F.storage_dtype(mode if storage else None):
        fake = G(F.leaf(z))
        scores = D(fake)
        loss = F.add(F.mean_all(F.square(scores)), F.mean_all(F.abs_(D.features)))
        backward(loss)
        penalty_value = 0.0
        if second_order:
            interpolates = F.leaf(fake.data * 0.7, requires_grad=True)
            D(interpolates)
            norms = F.row_norm(F.flatten2d(D.features))
            gradients, = backward(norms, grad=F.full_like(norms, 1.0), inputs=[interpolates], create_graph=True)
            penalty = F.mean_all(F.square(F.relu(F.add_scalar(F.row_norm(F.flatten2d(gradients)), -1.0))))
            backward(penalty)
            penalty_value = penalty.item()
    torch.cuda.synchronize()
    return loss.item(), penalty_value, G._srgan_arena.grad.clone(), D._srgan_arena.grad.clone(), fake.data.clone()


@pytest.mark.parametrize('size', [64, (64, 192)])
@pytest.mark.parametrize('second_order', [False, True])
def test_dcgan_pair_matches_the_fp32_storage_path(F, size, second_order):
    loss_a, penalty_a, g_a, d_a, fake_a = _dcgan_pass(F, size, False, 'bf16', second_order)
    loss_b, penalty_b, g_b, d_b, fake_b = _dcgan_pass(F, size, True, 'bf16', second_order)
    assert (fake_a - fake_b).abs().max().item() <= 3e-2
    assert abs(loss_a - loss_b) <= 3e-2 * abs(loss_a)
    if second_order:
        assert penalty_a > 1e-3 and abs(penalty_a - penalty_b) <= 8e-2 * penalty_a
    for a, b in ((g_a, g_b), (d_a, d_b)):
        # (a bias gradient is a sum of ~7e4 terms of both signs: the rounding of the stored gradients moves a small total by
        # more than its own size, so the bound is on the arena's scale and the direction)
        assert (a - b).abs().max().item() <= 1e-1 * a.abs().max().item()
        assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.99


@pytest.mark.parametrize('size,conv_dim', [(64, 64), ((64, 192), 32), (32, 16)])
def test_dcgan_pair_on_blocked_fp32_equals_the_nchw_kernels(F, size, conv_dim):
    """The exact form (fp32 tensors, four channels per slot): generator and discriminator stages on the blocked kernels
    against the fp32 NCHW kernels -- the same fp32 products in another summation order: 1e-5 of the scale, first and second
    order."""
    loss_a, penalty_a, g_a, d_a, fake_a = _dcgan_pass(F, size, False, 'f32b', True, conv_dim)
    loss_b, penalty_b, g_b, d_b, fake_b = _dcgan_pass(F, size, True, 'f32b', True, conv_dim)
    assert (fake_a - fake_b).abs().max().item() <= 1e-5, (fake_a - fake_b).abs().max().item()
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_a) and penalty_a > 1e-3 and abs(penalty_a - penalty_b) <= 1e-4 * penalty_a
    for a, b in ((g_a, g_b), (d_a, d_b)):
        assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item(), ((a - b).abs().max().item(), a.abs().max().item())


@pytest.mark.parametrize('batch', [8, 128])
def test_driving_step_on_fp16_storage_against_the_fp32_oracle(batch):
    """BASELINE.json configs[4] on the 16-bit data path (fp16 tensors in the blocked layout for D / DNN / G, the gradient-penalty
    chain in fp32 on the fp32 kernels, static loss scale) against the fp32 CPU oracle, at a small batch and at 128."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_mixed_precision_gpu import _step_against_fp32_oracle
    import srgan_amd  # noqa: F401
    from srgan_amd.driving.srgan import DrivingExperiment
    from oracle import models as OM
    size = (64, 192)

    def configure(experiment):
        experiment.image_size = size
    _step_against_fp32_oracle(DrivingExperiment, configure,
                              lambda: (OM.DCGANGenerator(image_size=size), OM.DCGANDiscriminator(image_size=size),
                                       OM.DCGANDiscriminator(image_size=size)),
                              size=size, batch=batch, d_scale=2.2,
                              settings_overrides=dict(compute_dtype='f16', gradient_penalty_dtype='f32', storage_dtype='f16',
                                                      loss_scale=256.0),
                              tolerance=2e-2)


@pytest.mark.parametrize('name,steps', [('g7b_crowd64', 2), ('g7c_crowd64_gp_active', 1)])
def test_blocked_fp32_stages_in_a_crowd_step_match_the_goldens(F, name, steps):
    """``settings.blocked_fp32``: the crowd generator's 4x4 / stride 2 transposed convolutions (reference
    crowd/models.py:132-146) on fp32 tensors in the blocked layout -- forward in the discriminator and generator steps, data
    and weight gradients in the generator step -- against the reference-generated goldens (forward outputs, every logged loss,
    the post-step weights), 1e-3."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import srgan_amd
    import test_steps_gpu as reference_tests
    reference_tests.test_crowd_steps(srgan_amd, name, 64, steps, False, extra_settings=dict(blocked_fp32=True))
