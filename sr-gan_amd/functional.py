"""Differentiable operations: each is one (or a few) launches of libsrgan_hip.so recorded on the tape.

Every backward below is written with the operations of this module, so gradients of gradients (the
gradient penalty, reference srgan.py:360-375) come for free.  Tensors are contiguous fp32 on the device;
torch is the allocator only -- no torch arithmetic is used on this path.
"""
import math
import os

import torch

from . import _lib
from .tape import Var, Node, grad_enabled, no_grad, accumulates_into

# op codes of srgan_ew_unary / srgan_ew_binary (include/srgan_hip.h)
U_COPY, U_NEG, U_ABS, U_SIGN, U_SQRT, U_EXP, U_LOG, U_LOG1P, U_SQUARE, U_RECIP, U_TANH, U_RELU, U_STEP, U_AFFINE, \
    U_POW, U_LEAKY, U_SIGMOID, U_SOFTPLUS, U_ONE_MINUS_SQ, U_RSQRT = range(20)
B_ADD, B_SUB, B_MUL, B_DIV, B_DIV_SAFE, B_MAX, B_LEAKY_MASK_MUL, B_AXPY = range(8)

FORCE_KERNEL = 0   # tests set 1 (direct) / 2 (MFMA) to cross-check the two contraction kernels

# MFMA operand type of the contractions launched from here on (include/srgan_hip.h SRGAN_COMPUTE_*): tensors stay fp32,
# only the two operands of a convolution / linear pass are rounded when the matrix instruction is fed.  Read when an
# operation LAUNCHES, so a backward pass runs in whatever mode is active while it is swept.
COMPUTE_DTYPES = {'f32': 0, 'fp32': 0, 'bf16': 1, 'f16': 2, 'fp16': 2}
COMPUTE_DTYPE = 0


# Storage type of the activations / gradients of the networks that have a 16-bit data path (``blocked16``): 0 = fp32
# tensors (the operands are rounded per fragment when COMPUTE_DTYPE asks for it), 1 / 2 = bf16 / fp16 tensors in the
# blocked layout with fused activations.  Read when a network's forward runs.
STORAGE_DTYPE = 0
STORAGE_BLOCKED_F32 = 4        # 'f32b': fp32 tensors in the blocked layout (four channels per slot): the exact form of the 4x4 / s2 family


class storage_dtype:
    """``with F.storage_dtype('bf16'): ...`` -- networks with a 16-bit data path run it inside; ``'f32b'``: the DCGAN stacks
    run their 4x4 / stride 2 stages on fp32 tensors in the blocked layout (exact arithmetic, other kernels)."""

    def __init__(self, name):
        if name == 'f32b':
            self.code = STORAGE_BLOCKED_F32
        else:
            self.code = (COMPUTE_DTYPES[name] if isinstance(name, str) else int(name)) if name else 0

    def __enter__(self):
        global STORAGE_DTYPE
        self.previous, STORAGE_DTYPE = STORAGE_DTYPE, self.code
        return self

    def __exit__(self, *exc):
        global STORAGE_DTYPE
        STORAGE_DTYPE = self.previous


class compute_dtype:
    """``with F.compute_dtype('bf16'): ...`` -- contractions launched inside use bf16 (or 'f16') MFMA operands with fp32
    accumulation; 'f32' restores the exact path (e.g. around the gradient-penalty chain of an fp16 step)."""

    def __init__(self, name):
        self.code = COMPUTE_DTYPES[name] if isinstance(name, str) else int(name)

    def __enter__(self):
        global COMPUTE_DTYPE
        self.previous, COMPUTE_DTYPE = COMPUTE_DTYPE, self.code
        return self

    def __exit__(self, *exc):
        global COMPUTE_DTYPE
        COMPUTE_DTYPE = self.previous


# ------------------------------------------------------------------------------------------- plumbing
def _stream():
    return _lib.stream_handle()


def _ptr(x):
    if x is None:
        return None
    return (x.data if isinstance(x, Var) else x).data_ptr()


POISON = bool(os.environ.get('SRGAN_POISON_EMPTY'))      # debugging aid: every fresh tensor starts as NaN, so a kernel
                                                        # that reads memory nobody wrote shows up in the results


def _empty(shape, like):
    if POISON:
        return torch.full(tuple(shape), float('nan'), dtype=torch.float32, device=like.device)
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _zeros(shape, like):
    data = _empty(shape, like)
    _call('srgan_fill', data.data_ptr(), data.numel(), 0.0, _stream())
    return data


def _check_device(t):
    if not t.is_cuda:
        raise _lib.HipLibraryError('srgan_amd operations need device tensors (there is no CPU fallback)')


def leaf(tensor, requires_grad=False):
    """Wrap a torch tensor (made contiguous fp32) as a graph leaf."""
    data = tensor.detach()
    if data.dtype != torch.float32:
        data = data.float()
    if not data.is_contiguous():
        data = data.contiguous()
    _check_device(data)
    return Var(data, requires_grad=requires_grad)


def constant(tensor):
    return leaf(tensor, False)


def _out(data, inputs, backward, name):
    requires = grad_enabled() and any(v is not None and v.requires_grad for v in inputs)
    out = Var(data, requires_grad=requires)
    if requires:
        out.node = Node(tuple(inputs), backward, name)
    return out


def _call(name, *args):
    _lib.check(getattr(_lib.library(), name)(*args), name)


# ------------------------------------------------------------------------------------------- raw kernels
def _unary_raw(op, x, p0=0.0, p1=0.0, out=None):
    out = _empty(x.shape, x) if out is None else out
    _call('srgan_ew_unary', op, x.data_ptr(), out.data_ptr(), x.numel(), p0, p1, _stream())
    return out


def _binary_raw(op, a, b, p0=0.0, out=None):
    if a.numel() != b.numel():
        raise ValueError(f'elementwise shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}')
    out = _empty(a.shape, a) if out is None else out
    _call('srgan_ew_binary', op, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), p0, _stream())
    return out


def fill_(tensor, value):
    _call('srgan_fill', tensor.data_ptr(), tensor.numel(), float(value), _stream())
    return tensor


def accumulate_(buffer, var):
    """buffer += var (in place; used for parameter-gradient accumulation into the flat arena)."""
    _binary_raw(B_ADD, buffer, var.data, out=buffer)


def full_like(var, value):
    return Var(fill_(_empty(var.shape, var.data), value))


def full(shape, value, like):
    return Var(fill_(_empty(shape, like.data if isinstance(like, Var) else like), value))


# ------------------------------------------------------------------------------------------- elementwise
def _unary(op, x, backward, name, p0=0.0, p1=0.0):
    return _out(_unary_raw(op, x.data, p0, p1), (x,), backward, name)


def add(a, b):
    if a.meta is not None or b.meta is not None:          # 16-bit blocked tensors (gradients summed by the tape's sweep)
        from . import blocked16
        return blocked16.add(a, b)
    return _out(_binary_raw(B_ADD, a.data, b.data), (a, b), lambda g, needs: (g, g), 'add')


def sub(a, b):
    return _out(_binary_raw(B_SUB, a.data, b.data), (a, b), lambda g, needs: (g, neg(g) if needs[1] else None), 'sub')


def mul(a, b):
    return _out(_binary_raw(B_MUL, a.data, b.data), (a, b),
                lambda g, needs: (mul(g, b) if needs[0] else None, mul(g, a) if needs[1] else None), 'mul')


def _div(a, b, op, name):
    out = _out(_binary_raw(op, a.data, b.data), (a, b), None, name)
    if out.node is not None:
        fn = div if op == B_DIV else div_safe
        out.node.backward = lambda g, needs: (fn(g, b) if needs[0] else None,
                                             neg(mul(g, fn(out, b))) if needs[1] else None)
    return out


def div(a, b):
    return _div(a, b, B_DIV, 'div')


def div_safe(a, b):
    """a / b with 0 where b == 0 (the sub-gradient torch uses for norm() at zero)."""
    return _div(a, b, B_DIV_SAFE, 'div_safe')


def mask_mul(g, reference, slope=0.0):
    """g where reference > 0 else slope * g: the backward of relu (slope 0) / leaky_relu; linear in g."""
    return _out(_binary_raw(B_LEAKY_MASK_MUL, g.data, reference.data, slope), (g, reference),
                lambda gg, needs: (mask_mul(gg, reference, slope) if needs[0] else None, None), 'mask_mul')


def affine(x, multiplier=1.0, offset=0.0):
    return _unary(U_AFFINE, x, lambda g, needs: (affine(g, multiplier),), 'affine', multiplier, offset)


def scale(x, c):
    return affine(x, c, 0.0)


def add_scalar(x, c):
    return affine(x, 1.0, c)


def neg(x):
    return _unary(U_NEG, x, lambda g, needs: (neg(g),), 'neg')


def sign(x):
    return Var(_unary_raw(U_SIGN, x.data))


def abs_(x):
    return _unary(U_ABS, x, lambda g, needs: (mul(g, sign(x)),), 'abs')


def sqrt(x):
    out = _unary(U_SQRT, x, None, 'sqrt')
    if out.node is not None:
        out.node.backward = lambda g, needs: (div(scale(g, 0.5), out),)
    return out


def exp(x):
    out = _unary(U_EXP, x, None, 'exp')
    if out.node is not None:
        out.node.backward = lambda g, needs: (mul(g, out),)
    return out


def log(x):
    return _unary(U_LOG, x, lambda g, needs: (div(g, x),), 'log')


def log1p(x):
    return _unary(U_LOG1P, x, lambda g, needs: (div(g, add_scalar(x, 1.0)),), 'log1p')


def square(x):
    return _unary(U_SQUARE, x, lambda g, needs: (mul(scale(g, 2.0), x),), 'square')


def one_minus_square(y):
    return _unary(U_ONE_MINUS_SQ, y, lambda g, needs: (scale(mul(g, y), -2.0),), 'one_minus_square')


def tanh(x):
    out = _unary(U_TANH, x, None, 'tanh')
    if out.node is not None:
        out.node.backward = lambda g, needs: (mul(g, one_minus_square(out)),)
    return out


def relu(x):
    return _unary(U_RELU, x, lambda g, needs: (mask_mul(g, x, 0.0),), 'relu')


def leaky_relu(x, negative_slope=0.01):
    return _unary(U_LEAKY, x, lambda g, needs: (mask_mul(g, x, negative_slope),), 'leaky_relu', negative_slope)


def pow_scalar(x, p):
    if p == 2:
        return square(x)
    if p == 1:
        return x
    return _unary(U_POW, x, lambda g, needs: (mul(g, scale(pow_scalar(x, p - 1.0), p)),), 'pow', p)


def sigmoid(x):
    out = _unary(U_SIGMOID, x, None, 'sigmoid')
    if out.node is not None:
        out.node.backward = lambda g, needs: (mul(g, sub(out, square(out))),)
    return out


def softplus(x):
    return _unary(U_SOFTPLUS, x, lambda g, needs: (mul(g, sigmoid(x)),), 'softplus')


def view(x, shape):
    shape = tuple(shape)
    original = x.shape
    return _out(x.data.view(shape), (x,), lambda g, needs: (view(g, original),), 'view')


def flatten2d(x):
    return view(x, (x.shape[0], -1))


# ------------------------------------------------------------------------------------------- channel ops
def _dims_nchw(shape):
    if len(shape) < 2:
        raise ValueError('need at least [N, C]')
    hw = 1
    for extent in shape[2:]:
        hw *= extent
    return shape[0], shape[1], hw


def chan_affine(x, mean=None, scale_a=None, scale_b=None, shift=None, dims=None, out_shape=None):
    """y[n,c,i] = ((x or 1) - mean[c]) * scale_a[c] * scale_b[c] + shift[c]; ``dims`` = (N, C, HW) view of the
    data (defaults to NCHW).  ``mean`` is treated as a constant.  NB a missing ``x`` is ONE, so a pure broadcast
    of a vector passes it as ``scale_a`` (not as ``shift``)."""
    reference = x if x is not None else next(v for v in (scale_a, scale_b, shift) if v is not None)
    if dims is None:
        dims = _dims_nchw(x.shape)
    if out_shape is None:
        out_shape = x.shape
    n, c, hw = dims
    data = _empty(out_shape, reference.data)
    if data.numel() != n * c * hw:
        raise ValueError(f'chan_affine: dims {dims} do not match shape {tuple(out_shape)}')
    for vector in (mean, scale_a, scale_b, shift):
        if vector is not None and vector.numel() != c:
            raise ValueError(f'chan_affine: vector of {vector.numel()} elements for {c} channels')
    _call('srgan_chan_affine', _ptr(x), _ptr(mean), _ptr(scale_a), _ptr(scale_b), _ptr(shift), data.data_ptr(), n, c, hw,
          _stream())

    def backward(g, needs):
        gx = chan_affine(g, None, scale_a, scale_b, None, dims, out_shape) if needs[0] else None
        ga = chan_reduce(g, x, mean, scale_b, dims, like=scale_a) if needs[1] else None
        gb = chan_reduce(g, x, mean, scale_a, dims, like=scale_b) if needs[2] else None
        gs = chan_reduce(g, None, None, None, dims, like=shift) if needs[3] else None
        return gx, ga, gb, gs
    return _out(data, (x, scale_a, scale_b, shift), backward, 'chan_affine')


def batch_norm_eval(x, mean, inv_std, gamma, beta, relu=False):
    """Frozen batch-norm (running statistics) optionally fused with the ReLU that follows it:
    y = [relu]((x - mean) * inv_std * gamma + beta) in ONE pass.  Its backward is two fused passes (input gradient;
    both parameter gradients) when only first-order gradients are needed, and is composed of differentiable
    primitives when the backward itself is being recorded (gradient penalty)."""
    n, c, hw = _dims_nchw(x.shape)
    data = _empty(x.shape, x.data)
    _call('srgan_chan_affine_act', _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma), _ptr(beta), None, 1 if relu else 0,
          data.data_ptr(), n, c, hw, _stream())
    out = _out(data, (x, gamma, beta), None, 'batch_norm_eval')
    if out.node is None:
        return out

    def backward(g, needs):
        if grad_enabled():
            gm = mask_mul(g, out, 0.0) if relu else g
            gx = chan_affine(gm, None, inv_std, gamma, None) if needs[0] else None
            ggamma = chan_reduce(gm, x, mean, inv_std) if needs[1] else None
            gbeta = chan_reduce(gm) if needs[2] else None
            return gx, ggamma, gbeta
        gx = ggamma = gbeta = None
        want_params = needs[1] or needs[2]
        gx_data = _empty(x.shape, x.data) if needs[0] else None
        # both parameter sums are ADDED by the kernel: straight into the gradient arena when the sweep accumulates there
        direct = needs[1] and needs[2] and accumulates_into(gamma) and accumulates_into(beta)
        both = _zeros((2, c), x.data) if want_params and not direct else None
        into_gamma = gamma.grad_buffer.data_ptr() if direct else (both[0].data_ptr() if want_params else None)
        into_beta = beta.grad_buffer.data_ptr() if direct else (both[1].data_ptr() if want_params else None)
        _call('srgan_bn_act_bwd', _ptr(g), _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma), _ptr(beta),
              1 if relu else 0, gx_data.data_ptr() if needs[0] else None, into_gamma, into_beta, n, c, hw, 0, 0, 0, 0, 0,
              _stream())
        if needs[0]:
            gx = Var(gx_data)
        if want_params and not direct:
            ggamma, gbeta = Var(both[0]), Var(both[1])
        return gx, ggamma, gbeta
    out.node.backward = backward
    return out


def chan_reduce(a, b=None, mean=None, scale=None, dims=None, like=None):
    """out[c] = scale[c] * sum_{n,i} a[n,c,i] * ((b or 1) - mean[c]) -> shape of ``like`` (default [C])."""
    if dims is None:
        dims = _dims_nchw(a.shape)
    n, c, hw = dims
    if a.numel() != n * c * hw or (b is not None and b.numel() != a.numel()):
        raise ValueError('chan_reduce: operand sizes do not match dims')
    out_shape = like.shape if like is not None else (c,)
    data = _empty(out_shape, a.data)
    _call('srgan_chan_reduce', _ptr(a), _ptr(b), _ptr(mean), _ptr(scale), data.data_ptr(), n, c, hw, 0, _stream())
    a_shape, b_shape = a.shape, (b.shape if b is not None else None)

    def backward(g, needs):
        gflat = view(g, (c,)) if g.shape != (c,) else g
        ga = chan_affine(b, mean, scale, gflat, None, dims, a_shape) if needs[0] else None
        gb = chan_affine(a, None, scale, gflat, None, dims, b_shape) if needs[1] else None
        gscale = None
        if needs[2]:
            gscale = mul(gflat, chan_reduce(a, b, mean, None, dims))
            if scale.shape != (c,):
                gscale = view(gscale, scale.shape)
        return ga, gb, gscale
    return _out(data, (a, b, scale), backward, 'chan_reduce')


def sum_all(x):
    """Sum of every element -> shape [1]."""
    return chan_reduce(x, dims=(1, 1, x.numel()))


def mean_all(x):
    return scale(sum_all(x), 1.0 / x.numel())


def scalar_mul(x, s):
    """x * s for a device scalar ``s`` of shape [1]."""
    return chan_affine(x, None, s, None, None, dims=(1, 1, x.numel()), out_shape=x.shape)


def row_scale(x, s):
    """x[b, ...] * s[b]."""
    b = x.shape[0]
    return chan_affine(x, None, view(s, (b,)) if s.shape != (b,) else s, None, None,
                       dims=(1, b, x.numel() // b), out_shape=x.shape)


def row_dot(a, b):
    """sum over everything but the leading dimension of a * b -> [B]."""
    rows = a.shape[0]
    return chan_reduce(a, b, dims=(1, rows, a.numel() // rows))


def row_sum(a):
    rows = a.shape[0]
    return chan_reduce(a, None, dims=(1, rows, a.numel() // rows))


def col_sum(x):
    """Sum over the leading (batch) dimension of [B, F...] -> [F]."""
    rows = x.shape[0]
    return chan_reduce(x, None, dims=(rows, x.numel() // rows, 1))


def row_broadcast(s, shape):
    """out[b, ...] = s[b]."""
    rows = shape[0]
    count = 1
    for extent in shape[1:]:
        count *= extent
    return chan_affine(None, None, view(s, (rows,)) if s.shape != (rows,) else s, None, None,
                       dims=(1, rows, count), out_shape=tuple(shape))


def col_broadcast(v, rows):
    """out[b, f] = v[f] for b < rows."""
    width = v.numel()
    return chan_affine(None, None, view(v, (width,)) if v.shape != (width,) else v, None, None,
                       dims=(rows, width, 1), out_shape=(rows, width))


def row_norm(x):
    """Per-example L2 norm over everything but the leading dimension -> [B]
    (reference srgan.py:371 and :381: ``.norm(dim=1)``), zero sub-gradient at zero like torch."""
    with no_grad():
        out = sqrt(row_dot(x, x))
    if not (grad_enabled() and x.requires_grad):
        return out
    result = Var(out.data, requires_grad=True)
    result.node = Node((x,), lambda g, needs: (row_scale(x, div_safe(g, result)),), 'row_norm')
    return result


# ------------------------------------------------------------------------------------------- contractions
def _conv_desc(x_shape, w_shape, stride, padding, y_shape):
    n, c, h, w = x_shape
    k, c2, r, s = w_shape
    if c != c2:
        raise ValueError(f'conv: input has {c} channels, weight expects {c2}')
    return _lib.ConvDesc(n, c, h, w, k, r, s, stride[0], stride[1], padding[0], padding[1], y_shape[2], y_shape[3], 0, 0,
                         COMPUTE_DTYPE)


def _pair(value):
    return (value, value) if isinstance(value, int) else tuple(value)


def conv_output_shape(x_shape, w_shape, stride, padding):
    n, _, h, w = x_shape
    k, _, r, s = w_shape
    return (n, k, (h + 2 * padding[0] - r) // stride[0] + 1, (w + 2 * padding[1] - s) // stride[1] + 1)


def conv2d(x, weight, bias=None, stride=1, padding=0):
    """torch.nn.functional.conv2d (groups = dilation = 1) on the MFMA gather-GEMM kernel."""
    stride, padding = _pair(stride), _pair(padding)
    y_shape = conv_output_shape(x.shape, weight.shape, stride, padding)
    desc = _conv_desc(x.shape, weight.shape, stride, padding, y_shape)
    data = _empty(y_shape, x.data)
    _call('srgan_conv2d_fwd', desc, _ptr(x), _ptr(weight), _ptr(bias), data.data_ptr(), FORCE_KERNEL, _stream())
    x_shape, w_shape = x.shape, weight.shape

    def backward(g, needs):
        gx = conv2d_backward_data(g, weight, x_shape, stride, padding) if needs[0] else None
        gw = gb = None
        if needs[1] and accumulates_into(weight):          # straight into the gradient arena (kernel accumulate mode)
            _call('srgan_conv2d_bwd_weight', _conv_desc(x_shape, w_shape, stride, padding, g.shape), _ptr(x), _ptr(g),
                  weight.grad_buffer.data_ptr(), 1, FORCE_KERNEL, _stream())
        elif needs[1]:
            gw = conv2d_backward_weight(x, g, w_shape, stride, padding)
        if needs[2] and accumulates_into(bias):
            _bias_gradient_into(bias.grad_buffer, g)
        elif needs[2]:
            gb = chan_reduce(g)
        return gx, gw, gb
    return _out(data, (x, weight, bias), backward, 'conv2d')


def _bias_gradient_into(buffer, g):
    """buffer[c] += sum over batch and pixels of g[n, c, ...] (a bias gradient added straight into the arena)."""
    n, c, hw = _dims_nchw(g.shape)
    _call('srgan_chan_reduce', _ptr(g), None, None, None, buffer.data_ptr(), n, c, hw, 1, _stream())


def conv2d_backward_data(gy, weight, x_shape, stride, padding, bias=None):
    """d conv2d / d input applied to ``gy`` (+ bias over the result's channels).  Also the forward of a
    transposed convolution."""
    x_shape = tuple(x_shape)
    desc = _conv_desc(x_shape, weight.shape, stride, padding, gy.shape)
    data = _empty(x_shape, gy.data)
    _call('srgan_conv2d_bwd_data', desc, _ptr(gy), _ptr(weight), _ptr(bias), data.data_ptr(), 0, FORCE_KERNEL, _stream())
    w_shape = weight.shape

    def backward(g, needs):
        ggy = conv2d(g, weight, None, stride, padding) if needs[0] else None
        gw = gb = None
        if needs[1] and accumulates_into(weight):
            _call('srgan_conv2d_bwd_weight', _conv_desc(g.shape, w_shape, stride, padding, gy.shape), _ptr(g), _ptr(gy),
                  weight.grad_buffer.data_ptr(), 1, FORCE_KERNEL, _stream())
        elif needs[1]:
            gw = conv2d_backward_weight(g, gy, w_shape, stride, padding)
        if needs[2] and accumulates_into(bias):
            _bias_gradient_into(bias.grad_buffer, g)
        elif needs[2]:
            gb = chan_reduce(g)
        return ggy, gw, gb
    return _out(data, (gy, weight, bias), backward, 'conv2d_backward_data')


def conv2d_backward_weight(x, gy, w_shape, stride, padding):
    """d conv2d / d weight applied to ``gy``."""
    w_shape = tuple(w_shape)
    desc = _conv_desc(x.shape, w_shape, stride, padding, gy.shape)
    data = _empty(w_shape, x.data)
    _call('srgan_conv2d_bwd_weight', desc, _ptr(x), _ptr(gy), data.data_ptr(), 0, FORCE_KERNEL, _stream())
    x_shape = x.shape

    def backward(g, needs):
        gx = conv2d_backward_data(gy, g, x_shape, stride, padding) if needs[0] else None
        ggy = conv2d(x, g, None, stride, padding) if needs[1] else None
        return gx, ggy
    return _out(data, (x, gy), backward, 'conv2d_backward_weight')


def conv_transpose2d(x, weight, bias=None, stride=1, padding=0):
    """torch.nn.functional.conv_transpose2d (weight [Cin, Cout, R, S]) = conv backward-data."""
    stride, padding = _pair(stride), _pair(padding)
    n, cin, h, w = x.shape
    _, cout, r, s = weight.shape
    out_shape = (n, cout, (h - 1) * stride[0] - 2 * padding[0] + r, (w - 1) * stride[1] - 2 * padding[1] + s)
    return conv2d_backward_data(x, weight, out_shape, stride, padding, bias)


def mm(a, b, transpose_a=False, transpose_b=False, bias=None, bias_on_columns=True):
    """op(a) @ op(b) for 2-D operands (+ bias broadcast over rows or columns)."""
    m, k = (a.shape[1], a.shape[0]) if transpose_a else a.shape
    k2, n = (b.shape[1], b.shape[0]) if transpose_b else b.shape
    if k != k2:
        raise ValueError(f'mm: inner dimensions differ ({k} vs {k2})')
    sai, sak = (1, a.shape[1]) if transpose_a else (a.shape[1], 1)
    sbk, sbj = (1, b.shape[1]) if transpose_b else (b.shape[1], 1)
    data = _empty((m, n), a.data)
    _call('srgan_gemm', m, n, k, _ptr(a), sai, sak, _ptr(b), sbk, sbj, data.data_ptr(), n, 1, _ptr(bias),
          1 if bias_on_columns else 0, 0, FORCE_KERNEL, COMPUTE_DTYPE, _stream())

    def backward(g, needs):
        ga = gb = gbias = None
        if needs[0]:
            ga = mm(g, b, False, not transpose_b) if not transpose_a else mm(b, g, transpose_b, True)
        if needs[1]:
            gb = mm(a, g, not transpose_a, False) if not transpose_b else mm(g, a, True, transpose_a)
        if needs[2] and accumulates_into(bias):
            dims = (m, n, 1) if bias_on_columns else (1, m, n)
            _call('srgan_chan_reduce', _ptr(g), None, None, None, bias.grad_buffer.data_ptr(), dims[0], dims[1], dims[2], 1,
                  _stream())
        elif needs[2]:
            gbias = chan_reduce(g, dims=(m, n, 1)) if bias_on_columns else chan_reduce(g, dims=(1, m, n))
        return ga, gb, gbias
    return _out(data, (a, b, bias), backward, 'mm')


def linear(x, weight, bias=None):
    """torch.nn.functional.linear for x [B, in], weight [out, in]."""
    return mm(x, weight, False, True, bias, True)


# ------------------------------------------------------------------------------------------- layout ops
def cat_channels(parts):
    """Concatenate [N, Ci, ...] tensors along dim 1 (reference crowd/models.py:353,1159-1165)."""
    n = parts[0].shape[0]
    spatial = parts[0].shape[2:]
    hw = 1
    for extent in spatial:
        hw *= extent
    widths = [p.shape[1] for p in parts]
    total = sum(widths)
    data = _empty((n, total) + tuple(spatial), parts[0].data)
    first = 0
    for part, width in zip(parts, widths):
        if part.shape[0] != n or tuple(part.shape[2:]) != tuple(spatial):
            raise ValueError('cat_channels: incompatible shapes')
        _call('srgan_copy_channels', _ptr(part), width, 0, data.data_ptr(), total, first, width, n, hw, 0, _stream())
        first += width

    def backward(g, needs):
        grads, start = [], 0
        for need, width in zip(needs, widths):
            grads.append(slice_channels(g, start, start + width) if need else None)
            start += width
        return tuple(grads)
    return _out(data, tuple(parts), backward, 'cat_channels')


def cat_rows(parts):
    """Concatenate along the leading (batch) dimension: one contiguous copy per part."""
    trailing = tuple(parts[0].shape[1:])
    counts = [p.shape[0] for p in parts]
    data = _empty((sum(counts),) + trailing, parts[0].data)
    start = 0
    for part, count in zip(parts, counts):
        if tuple(part.shape[1:]) != trailing:
            raise ValueError('cat_rows: incompatible shapes')
        _unary_raw(U_COPY, part.data, out=data[start:start + count])
        start += count

    def backward(g, needs):
        grads, first = [], 0
        for need, count in zip(needs, counts):
            grads.append(narrow_rows(g, first, count) if need else None)
            first += count
        return tuple(grads)
    return _out(data, tuple(parts), backward, 'cat_rows')


def narrow_rows(x, first, count):
    """Rows [first, first + count) of the leading dimension as a VIEW (no copy); the gradient is the cotangent
    placed into a zero tensor of the full shape."""
    total = x.shape[0]
    if first < 0 or count < 0 or first + count > total:
        raise ValueError('narrow_rows: range outside the leading dimension')
    shape = tuple(x.shape)

    def backward(g, needs):
        full = _zeros(shape, g.data)
        _unary_raw(U_COPY, g.data, out=full[first:first + count])
        return (_out(full, (g,), lambda gg, n2: (narrow_rows(gg, first, count),), 'narrow_rows_backward'),)
    return _out(x.data[first:first + count], (x,), backward, 'narrow_rows')


def slice_channels(x, first, last):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    count = last - first
    data = _empty((n, count) + tuple(x.shape[2:]), x.data)
    _call('srgan_copy_channels', _ptr(x), c, first, data.data_ptr(), count, 0, count, n, hw, 0, _stream())
    return _out(data, (x,), lambda g, needs: (embed_channels(g, c, first),), 'slice_channels')


def embed_channels(x, total, first):
    """Zero tensor with ``total`` channels holding x at channels [first, first + C)."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    data = fill_(_empty((n, total) + tuple(x.shape[2:]), x.data), 0.0)
    _call('srgan_copy_channels', _ptr(x), c, 0, data.data_ptr(), total, first, c, n, hw, 0, _stream())
    return _out(data, (x,), lambda g, needs: (slice_channels(g, first, first + c),), 'embed_channels')


# ------------------------------------------------------------------------------------------- pooling
def max_pool2d(x, kernel_size, stride, padding=0):
    n, c, h, w = x.shape
    oh = (h + 2 * padding - kernel_size) // stride + 1
    ow = (w + 2 * padding - kernel_size) // stride + 1
    data = _empty((n, c, oh, ow), x.data)
    argmax = torch.empty((n, c, oh, ow), dtype=torch.int32, device=x.data.device)
    _call('srgan_maxpool2d_fwd', _ptr(x), data.data_ptr(), argmax.data_ptr(), n * c, h, w, kernel_size, stride, padding,
          oh, ow, _stream())
    in_shape = x.shape
    geometry = (kernel_size, stride, padding, oh, ow)
    return _out(data, (x,), lambda g, needs: (_max_pool2d_backward(g, argmax, in_shape, geometry),), 'max_pool2d')


def bn_relu_max_pool2d(x, mean, inv_std, gamma, beta, kernel_size, stride, padding=0):
    """``max_pool2d(relu(batch_norm_eval(x)))`` as ONE pass (the DenseNet stem's norm0 -> relu0 -> pool0, reference
    crowd/models.py:1072-1076): the activated tensor -- four times the pooled one -- is never written.  First-order
    backward: one pass as well (`srgan_bn_relu_maxpool_bwd`: pooled gradient gathered per pixel, mask, scale, both
    parameter sums).  A RECORDED backward (gradient penalty) re-evaluates the two-op form and differentiates that, so
    second order costs what it did.  Returns None when the backward kernel does not have the geometry."""
    n, c, h, w = x.shape
    if not _lib.library().srgan_bn_relu_maxpool_bwd_supported(n, c, h, w, kernel_size, stride, padding):
        return None
    oh = (h + 2 * padding - kernel_size) // stride + 1
    ow = (w + 2 * padding - kernel_size) // stride + 1
    data = _empty((n, c, oh, ow), x.data)
    argmax = torch.empty((n, c, oh, ow), dtype=torch.int32, device=x.data.device)
    _call('srgan_bn_relu_maxpool_fwd', _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma), _ptr(beta), data.data_ptr(),
          argmax.data_ptr(), n, c, h, w, kernel_size, stride, padding, oh, ow, _stream())
    out = _out(data, (x, gamma, beta), None, 'bn_relu_max_pool2d')
    if out.node is None:
        return out

    def backward(g, needs):
        if grad_enabled():
            from .tape import backward as sweep
            pooled = max_pool2d(batch_norm_eval(x, mean, inv_std, gamma, beta, relu=True), kernel_size, stride, padding)
            wanted = [v for v, need in zip((x, gamma, beta), needs) if need]
            grads = iter(sweep(pooled, grad=g, inputs=wanted, create_graph=True))
            return tuple(next(grads) if need else None for need in needs)
        want_params = needs[1] or needs[2]
        direct = needs[1] and needs[2] and accumulates_into(gamma) and accumulates_into(beta)
        both = _zeros((2, c), x.data) if want_params and not direct else None
        into_gamma = gamma.grad_buffer.data_ptr() if direct else (both[0].data_ptr() if want_params else None)
        into_beta = beta.grad_buffer.data_ptr() if direct else (both[1].data_ptr() if want_params else None)
        gx_data = _empty(x.shape, x.data)
        _call('srgan_bn_relu_maxpool_bwd', _ptr(g), argmax.data_ptr(), _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma),
              _ptr(beta), gx_data.data_ptr(), into_gamma, into_beta, n, c, h, w, kernel_size, stride, padding, oh, ow, _stream())
        ggamma = Var(both[0]) if want_params and not direct else None
        gbeta = Var(both[1]) if want_params and not direct else None
        return (Var(gx_data) if needs[0] else None), ggamma, gbeta
    out.node.backward = backward
    return out


def bn_relu_avg_pool2d(x, mean, inv_std, gamma, beta):
    """``avg_pool2d(relu(batch_norm_eval(x)), 2, 2)`` as ONE pass each way (the DenseNet transitions evaluated as norm -> relu
    -> pool -> conv, see ``crowd.models._Transition``): the activated tensor -- four times the pooled one -- is never
    written; first-order backward in one pass too (`srgan_bn_relu_avgpool2_bwd`).  A RECORDED backward (gradient penalty)
    re-evaluates the two-op form and differentiates that.  Returns None when the kernels do not have the geometry."""
    n, c, h, w = x.shape
    if not _lib.library().srgan_bn_relu_avgpool2_supported(n, c, h, w) or (x.data.data_ptr() & 15):
        return None
    data = _empty((n, c, h // 2, w // 2), x.data)
    _call('srgan_bn_relu_avgpool2_fwd', _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma), _ptr(beta), data.data_ptr(),
          n, c, h, w, _stream())
    out = _out(data, (x, gamma, beta), None, 'bn_relu_avg_pool2d')
    if out.node is None:
        return out

    def backward(g, needs):
        if grad_enabled():
            from .tape import backward as sweep
            pooled = avg_pool2d(batch_norm_eval(x, mean, inv_std, gamma, beta, relu=True), 2, 2)
            wanted = [v for v, need in zip((x, gamma, beta), needs) if need]
            grads = iter(sweep(pooled, grad=g, inputs=wanted, create_graph=True))
            return tuple(next(grads) if need else None for need in needs)
        want_params = needs[1] or needs[2]
        direct = needs[1] and needs[2] and accumulates_into(gamma) and accumulates_into(beta)
        both = _zeros((2, c), x.data) if want_params and not direct else None
        into_gamma = gamma.grad_buffer.data_ptr() if direct else (both[0].data_ptr() if want_params else None)
        into_beta = beta.grad_buffer.data_ptr() if direct else (both[1].data_ptr() if want_params else None)
        gx_data = _empty(x.shape, x.data)
        upstream = g.data if g.data.is_contiguous() else g.data.contiguous()
        _call('srgan_bn_relu_avgpool2_bwd', upstream.data_ptr(), _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(gamma), _ptr(beta),
              gx_data.data_ptr(), into_gamma, into_beta, n, c, h, w, _stream())
        ggamma = Var(both[0]) if want_params and not direct else None
        gbeta = Var(both[1]) if want_params and not direct else None
        return (Var(gx_data) if needs[0] else None), ggamma, gbeta
    out.node.backward = backward
    return out


def _max_pool2d_backward(g, argmax, in_shape, geometry):
    """Gather form (every input element written once: no zero-fill, no atomics); its own backward is the gather of
    the incoming tensor at the forward's arg-max, as for the scatter form."""
    n, c, h, w = in_shape
    kernel_size, stride, padding, oh, ow = geometry
    data = _empty(in_shape, g.data)
    _call('srgan_maxpool2d_bwd', _ptr(g), argmax.data_ptr(), data.data_ptr(), n * c, h, w, kernel_size, stride, padding,
          oh, ow, _stream())
    return _out(data, (g,), lambda gg, needs: (_pool_gather(gg, argmax, g.shape),), 'max_pool2d_backward')


def _pool_scatter(g, argmax, in_shape):
    n, c, h, w = in_shape
    data = _empty(in_shape, g.data)
    out_plane = g.numel() // (n * c)
    _call('srgan_pool_scatter', _ptr(g), argmax.data_ptr(), data.data_ptr(), n * c, h * w, out_plane, _stream())
    return _out(data, (g,), lambda gg, needs: (_pool_gather(gg, argmax, g.shape),), 'pool_scatter')


def _pool_gather(src, argmax, out_shape):
    n, c, h, w = src.shape
    data = _empty(out_shape, src.data)
    out_plane = data.numel() // (n * c)
    _call('srgan_pool_gather', _ptr(src), argmax.data_ptr(), data.data_ptr(), n * c, h * w, out_plane, _stream())
    in_shape = src.shape
    return _out(data, (src,), lambda g, needs: (_pool_scatter(g, argmax, in_shape),), 'pool_gather')


def avg_pool2d(x, kernel_size, stride=None):
    stride = kernel_size if stride is None else stride
    n, c, h, w = x.shape
    oh, ow = (h - kernel_size) // stride + 1, (w - kernel_size) // stride + 1
    data = _empty((n, c, oh, ow), x.data)
    _call('srgan_avgpool2d_fwd', _ptr(x), data.data_ptr(), n * c, h, w, kernel_size, stride, oh, ow, _stream())
    in_shape = x.shape
    return _out(data, (x,), lambda g, needs: (_avg_pool2d_backward(g, in_shape, kernel_size, stride),), 'avg_pool2d')


def _avg_pool2d_backward(g, in_shape, kernel_size, stride):
    n, c, h, w = in_shape
    oh, ow = g.shape[2], g.shape[3]
    data = _empty(in_shape, g.data)
    _call('srgan_avgpool2d_bwd', _ptr(g), data.data_ptr(), n * c, h, w, kernel_size, stride, oh, ow, _stream())
    return _out(data, (g,), lambda gg, needs: (avg_pool2d(gg, kernel_size, stride),), 'avg_pool2d_backward')


# ------------------------------------------------------------------------------------------- step-specific
def gp_interpolate(unlabeled, fake, alpha):
    """alpha * u + (1 - alpha) * fake on detached inputs -> a fresh leaf that requires grad
    (reference srgan.py:365-366)."""
    b = unlabeled.shape[0]
    if fake.shape != unlabeled.shape or alpha.numel() != b:
        raise ValueError('gp_interpolate: shapes differ (alpha must have settings.batch_size entries, '
                         'reference srgan.py:363)')
    data = _empty(unlabeled.shape, unlabeled.data)
    _call('srgan_gp_interpolate', _ptr(unlabeled), _ptr(fake), _ptr(alpha), data.data_ptr(), b, unlabeled.numel() // b,
          _stream())
    return Var(data, requires_grad=True)


def crowd_map_l1(maps, target):
    """rows[b] = sum_{h,w} mean_c |maps[b,c,h,w] - target[b,h,w]| (reference crowd/srgan.py:252).
    First-order differentiable in ``maps`` (it is never inside the gradient-penalty graph)."""
    b, cm = maps.shape[0], maps.shape[1]
    hw = maps.numel() // (b * cm)
    data = _empty((b,), maps.data)
    _call('srgan_crowd_map_l1_fwd', _ptr(maps), _ptr(target), data.data_ptr(), b, cm, hw, _stream())

    def backward(g, needs):
        if grad_enabled():
            raise NotImplementedError('crowd_map_l1 is first-order only')
        gm = _empty(maps.shape, maps.data)
        _call('srgan_crowd_map_l1_bwd', _ptr(maps), _ptr(target), _ptr(g), gm.data_ptr(), b, cm, hw, _stream())
        return (Var(gm),)
    return _out(data, (maps,), backward, 'crowd_map_l1')


def row_max(x2d):
    """Constant (non-differentiable) per-row maximum of [B, F]."""
    b, f = x2d.shape
    data = _empty((b,), x2d.data)
    _call('srgan_row_max', _ptr(x2d), data.data_ptr(), b, f, _stream())
    return Var(data)


def nearest_bin_onehot(values, bins):
    """Constant one-hot [B, K] of the nearest bin centre (reference utility.py:141-144)."""
    b, k = values.numel(), bins.numel()
    data = _empty((b, k), values.data)
    _call('srgan_nearest_bin_onehot', _ptr(values), _ptr(bins), data.data_ptr(), b, k, _stream())
    return Var(data)


def logsumexp_rows(logits):
    """log(sum(exp(x), dim=1)) of [B, K], stabilised by the row maximum (reference utility.py:161-182; the
    maximum enters as a constant, which leaves value and gradient unchanged)."""
    b = logits.shape[0]
    s = row_max(logits)
    shifted = sub(logits, row_broadcast(s, logits.shape))
    return add(s, log(row_sum(exp(shifted))))
