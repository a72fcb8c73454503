"""The 16-bit data path on the tape (BASELINE.json configs[1] "bf16", configs[4] "fp16"; the reference is fp32-only).

Activations, their gradients and a shadow of every layer's weights are bf16 / fp16 tensors in the BLOCKED layout of
``csrc/blocked16.h`` -- ``[N][C / 8][H][W][8]``: the 8 channels of a group at a pixel are one 16-byte MFMA operand slot --
while the master weights, the weight / bias gradients (accumulated straight into the fp32 arena), Adam and the losses stay
fp32.  A ``Var`` of this kind carries ``meta`` (a ``Blocked``); ``pack`` / ``unpack`` are the only crossings.

Fused activations and the "pre-masked gradient" convention.  A layer is ``y = act(conv(x) + b)`` in ONE kernel (the
reference's ``conv -> ReLU`` / ``leaky_relu(conv)`` pairs, age/vgg.py:78-82, age/models.py:48-50,70-73) and only ``y`` is
stored.  Its ``meta.mask_ref`` is ``y`` itself: the activation's derivative is 1 where ``y > 0`` and ``slope`` elsewhere.
The gradient handed to the producer of a tensor with a ``mask_ref`` is ALREADY multiplied by that derivative: every
operation that consumes such a tensor applies the mask in the epilogue of the kernel that produces its input gradient (the
data-gradient convolution, the pool backward, the fp32 -> 16-bit conversion), so the un-masked gradient never exists in HBM
and nothing like ``unary_kernel`` / ``binary_kernel`` runs on this path.  Sums of gradients commute with the mask, so
several consumers simply add up.

Second order (the gradient penalty, reference srgan.py:360-375).  With relu / leaky_relu and max-pool the networks are
piecewise linear, so every backward operation is again "a linear map, then a mask": ``conv(s, W^T) * mask(ref)``.  Its own
backward is the same kind of call with the roles swapped (the LINEARISED forward: ``conv(s, W) * mask(ref')``) plus a
weight gradient from (tangent, first-backward gradient), so the recorded backward and its double backward reuse the
forward kernels; masks and arg-max positions are constants."""
import os

import torch

from . import _lib
from . import functional as F
from .tape import Var, accumulates_into

TORCH_DTYPE = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}
CODES = {'f32': 0, 'bf16': 1, 'f16': 2, 'fp16': 2}
# code 0: fp32 in the same blocked layout with FOUR channels per 16-byte slot -- the exact form of the 4x4 / stride 2 family
# (v_mfma_f32_32x32x2_f32; the fp32 gradient-penalty chain of the fp16 configuration, the crowd generator).  Conversions, sums
# and that family take it; the 3x3 / pooling / linear kernels are 16-bit only.


def group_size(code):
    return 4 if code == 0 else 8


def active_code():
    """The blocked dtype code the active ``F.storage_dtype`` asks for (None: plain fp32 tensors)."""
    if not F.STORAGE_DTYPE:
        return None
    return 0 if F.STORAGE_DTYPE == F.STORAGE_BLOCKED_F32 else F.STORAGE_DTYPE


class Blocked:
    """Logical shape and conventions of a 16-bit blocked tensor: ``c`` channels (``(c + 7) // 8`` groups) on an h x w plane;
    ``plane`` > 1 marks a FLATTENED tensor (h = w = 1) whose ``c`` entries are the slots of a [c0 / 8][plane][8] tensor."""
    __slots__ = ('n', 'c', 'h', 'w', 'code', 'mask_ref', 'slope', 'plane', 'real')

    def __init__(self, n, c, h, w, code, mask_ref=None, slope=0.0, plane=1, real=None):
        self.n, self.c, self.h, self.w, self.code = n, c, h, w, code
        self.mask_ref, self.slope, self.plane = mask_ref, slope, plane
        self.real = c if real is None else real          # features that exist (flattened tensors: channels x plane)

    @property
    def group(self):
        return group_size(self.code)

    @property
    def groups(self):
        return (self.c + self.group - 1) // self.group

    def like(self, **changes):
        values = {name: getattr(self, name) for name in self.__slots__}
        values.update(changes)
        return Blocked(**values)


def _call(name, *args):
    _lib.check(getattr(_lib.library(), name)(*args), name)


def _new(n, c, h, w, code, device):
    group = group_size(code)
    shape = (n, (c + group - 1) // group, h, w, group)
    if F.POISON:
        return torch.full(shape, float('nan'), dtype=TORCH_DTYPE[code], device=device)
    return torch.empty(shape, dtype=TORCH_DTYPE[code], device=device)


def _ptr(tensor):
    return None if tensor is None else tensor.data_ptr()


# ------------------------------------------------------------------------------------------------ conversions
def pack(x, code, mask_ref=None, slope=0.0):
    """fp32 [N, C, H, W] (or [N, F]) -> blocked 16-bit; with ``mask_ref`` the values are multiplied by the derivative of the
    activation that produced ``mask_ref`` (this is then the PRE-masked gradient of that activated tensor)."""
    shape = x.shape
    n, c = shape[0], shape[1]
    h, w = (shape[2], shape[3]) if len(shape) == 4 else (1, 1)
    if len(shape) not in (2, 4):
        raise ValueError(f'blocked16.pack takes [N, C, H, W] or [N, F], not {shape}')
    F._check_device(x.data)
    data = _new(n, c, h, w, code, x.data.device)
    _call('srgan_h_pack', x.data.data_ptr(), data.data_ptr(), _ptr(mask_ref), float(slope), n, c, h * w, code, F._stream())
    out = F._out(data, (x,), lambda s, needs: (unpack(s, shape),), 'h_pack')
    out.meta = Blocked(n, c, h, w, code, mask_ref, slope)
    return out


def unpack(x, shape=None):
    """blocked 16-bit -> fp32 [N, C, H, W] ([N, C] for 1 x 1 planes unless ``shape`` says otherwise)."""
    meta = x.meta
    if meta.plane != 1:
        raise ValueError('a flattened blocked tensor has no NCHW form: unpack before flatten')
    if shape is None:
        shape = (meta.n, meta.c) if meta.h == meta.w == 1 else (meta.n, meta.c, meta.h, meta.w)
    data = F._empty(shape, x.data)
    _call('srgan_h_unpack', x.data.data_ptr(), data.data_ptr(), meta.n, meta.c, meta.h * meta.w, meta.code, F._stream())
    return F._out(data, (x,), lambda g, needs: (pack(g, meta.code, meta.mask_ref, meta.slope),), 'h_unpack')


def flatten(x):
    """[N, C, H, W] -> [N, C/8 * H * W * 8] as a VIEW (the blocked order: a linear layer behind it permutes its weight
    shadow's columns instead, ``Shadow``)."""
    meta = x.meta
    if meta.h == meta.w == 1:
        return x
    if meta.code == 0:
        raise NotImplementedError('blocked16.flatten: the linear layers behind it are 16-bit only')
    plane, length = meta.h * meta.w, meta.groups * 8 * meta.h * meta.w

    def flat(tensor):
        return None if tensor is None else tensor.view(meta.n, length // 8, 1, 1, 8)
    flat_meta = Blocked(meta.n, length, 1, 1, meta.code, flat(meta.mask_ref), meta.slope, plane, meta.c * plane)

    def backward(s, needs):
        data = s.data.view(meta.n, meta.groups, meta.h, meta.w, 8)
        back = F._out(data, (s,), lambda t, n2: (_reflatten(t, flat_meta),), 'h_unflatten')
        back.meta = meta
        return (back,)
    out = F._out(flat(x.data), (x,), backward, 'h_flatten')
    out.meta = flat_meta
    return out


def _reflatten(t, flat_meta):
    out = F._out(t.data.view(flat_meta.n, flat_meta.c // 8, 1, 1, 8), (t,), None, 'h_flatten')
    out.meta = flat_meta
    if out.node is not None:
        raise NotImplementedError('third-order use of blocked16.flatten')
    return out


def add(a, b):
    if a.meta is None or b.meta is None or a.data.shape != b.data.shape or a.meta.code != b.meta.code:
        raise ValueError('blocked16.add: both operands must be 16-bit blocked tensors of one shape')
    data = torch.empty_like(a.data)
    _call('srgan_h_add', a.data.data_ptr(), b.data.data_ptr(), data.data_ptr(), data.numel() // a.meta.group, a.meta.code, F._stream())
    out = F._out(data, (a, b), lambda g, needs: (g, g), 'h_add')
    out.meta = a.meta
    return out


# ------------------------------------------------------------------------------------------------ pooling
def max_pool2(x):
    """max_pool2d(x, 2, 2) (reference age/vgg.py:76).  Backward: the pooled gradient placed at the arg-max and multiplied
    by the derivative of the activation that produced ``x`` in one pass; the arg-max is recomputed (first maximum)."""
    meta = x.meta
    if meta.h % 2 or meta.w % 2:
        raise ValueError('blocked16.max_pool2 needs even planes')
    if meta.mask_ref is not None and meta.mask_ref.data_ptr() != x.data.data_ptr():
        raise ValueError('blocked16.max_pool2 pools activated tensors (mask = the tensor itself) or plain ones')
    data = _new(meta.n, meta.c, meta.h // 2, meta.w // 2, meta.code, x.data.device)
    _call('srgan_h_maxpool2', x.data.data_ptr(), None, data.data_ptr(), meta.n * meta.groups, meta.h, meta.w, 0, meta.code,
          F._stream())
    out = F._out(data, (x,), lambda g, needs: (_pool_backward(x.data, meta, g),), 'h_max_pool2')
    out.meta = Blocked(meta.n, meta.c, meta.h // 2, meta.w // 2, meta.code)
    return out


def _pool_backward(x_data, meta, g):
    data = torch.empty_like(x_data)
    _call('srgan_h_maxpool2_bwd', x_data.data_ptr(), g.data.data_ptr(), data.data_ptr(), meta.n * meta.groups, meta.h, meta.w,
          1 if meta.mask_ref is not None else 0, float(meta.slope), meta.code, F._stream())
    out = F._out(data, (g,), lambda s, needs: (_pool_gather(x_data, meta, s),), 'h_max_pool2_backward')
    out.meta = meta
    return out


def _pool_gather(x_data, meta, s):
    """``s`` (shape of x, pre-masked) at the arg-max of x: the pool's action on a tangent."""
    data = _new(meta.n, meta.c, meta.h // 2, meta.w // 2, meta.code, x_data.device)
    _call('srgan_h_maxpool2', x_data.data_ptr(), s.data.data_ptr(), data.data_ptr(), meta.n * meta.groups, meta.h, meta.w, 2,
          meta.code, F._stream())
    out = F._out(data, (s,), lambda t, needs: (_pool_backward(x_data, meta, t),), 'h_max_pool2_gather')
    out.meta = Blocked(meta.n, meta.c, meta.h // 2, meta.w // 2, meta.code)
    return out


# ------------------------------------------------------------------------------------------------ weight shadows
class Shadow:
    """The 16-bit operand forms of one layer's fp32 master weights: ``forward`` (rows = outputs) and ``transposed`` (the
    data gradient's operand), re-rounded whenever the masters changed -- by ``refresh(arena)`` right behind the optimizer
    update (same stream as the update), or lazily when the version key below moved (checkpoint load, tests)."""

    def __init__(self, module, kind, code, in_blocked=0, plane=1):
        self.module, self.kind, self.code, self.in_blocked, self.plane = module, kind, code, in_blocked, plane
        self.forward = self.transposed = self.down = self.up = self.bias_rows = None
        self.key = None
        arena = getattr(module.weight, '_srgan_arena', None)
        if arena is not None:
            arena.shadows.append(self)

    def fresh(self):
        if self.key != self._version():
            # the lazy path (first use, checkpoint load, a test that edited the weights): other streams may launch readers of
            # this shadow right away -- they are only ordered behind whatever wrote the masters, not behind this repack
            self.repack()
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
        return self

    def repack(self):
        weight = self.module.weight
        lib, stream, device = _lib.library(), F._stream(), weight.device

        buffer = self._buffer
        if self.kind == 'conv3x3':
            k, c, r, s = weight.shape
            for name, transposed, rows, reduced in (('forward', 0, k, c), ('transposed', 1, c, k)):
                _call('srgan_h_pack_conv_weights', weight.data_ptr(), buffer(name, lib.srgan_h_conv_weight_slots(rows, reduced, r, s)),
                      k, c, r, s, transposed, self.code, stream)
        elif self.kind == 'linear':
            outputs, inputs = weight.shape[0], weight.numel() // weight.shape[0]
            blocked = self.in_blocked or (inputs + 7) // 8 * 8
            outputs_padded = (outputs + 7) // 8 * 8
            # forward operand A[o][f'] = W[o][map(f')]; transposed operand A[f'][o] = W[o][map(f')]
            _call('srgan_h_pack_matrix', weight.data_ptr(), buffer('forward', outputs * blocked // 8), outputs, blocked, outputs, inputs,
                  inputs, 1, 1, self.plane, self.code, stream)
            _call('srgan_h_pack_matrix', weight.data_ptr(), buffer('transposed', blocked * outputs_padded // 8), blocked, outputs_padded,
                  inputs, outputs, 1, inputs, self.plane, 1, self.code, stream)
        elif self.kind == 'k4s2':
            # [A][B][4][4], A = the channels on the small plane: conv2d weights [K][C], conv_transpose2d weights [Cin][Cout]
            a, b = weight.shape[0], weight.shape[1]
            for name, direction in (('down', 0), ('up', 1)):
                _call('srgan_h_pack_k4s2_weights', weight.data_ptr(), buffer(name, lib.srgan_h_k4s2_weight_slots(a, b, direction, self.code)),
                      a, b, direction, self.code, stream)
        elif self.kind == 'linear_t':
            # conv_transpose2d weights [Cin][Cout][R][S] applied to a 1 x 1 input: a linear map Cin -> (Cout, R x S); its outputs
            # in the blocked order of a [Cout, R, S] tensor (plane = R * S)
            c_in, c_out = weight.shape[0], weight.shape[1]
            plane = weight.shape[2] * weight.shape[3]
            blocked, c_in_padded = (c_out + 7) // 8 * 8 * plane, (c_in + 7) // 8 * 8
            _call('srgan_h_pack_matrix', weight.data_ptr(), buffer('forward', blocked * c_in_padded // 8), blocked, c_in_padded,
                  c_out * plane, c_in, 1, c_out * plane, plane, 1, self.code, stream)
            _call('srgan_h_pack_matrix', weight.data_ptr(), buffer('transposed', c_in * blocked // 8), c_in, blocked, c_in,
                  c_out * plane, c_out * plane, 1, 1, plane, self.code, stream)
            if self.module.bias is not None:       # bias[co] of every output row (co, pixel), in the blocked order (a copy, no arithmetic)
                groups = (c_out + 7) // 8
                padded = torch.zeros(groups * 8, dtype=torch.float32, device=device)
                padded[:c_out] = self.module.bias.data
                self.bias_rows = padded.view(groups, 1, 8).expand(groups, plane, 8).reshape(-1).contiguous()
        else:
            raise ValueError(self.kind)
        self.key = self._version()

    def _buffer(self, name, slots):
        """The device buffer of one operand form (``slots`` 16-byte slots), allocated on first use and then kept."""
        if getattr(self, name, None) is None:
            setattr(self, name, torch.empty(slots * 4, dtype=torch.int32, device=self.module.weight.device))
        return getattr(self, name).data_ptr()

    def _version(self):
        weight = self.module.weight
        arena = getattr(weight, '_srgan_arena', None)
        bias = getattr(self.module, 'bias', None)
        return (weight.data_ptr(), weight._version, None if bias is None else bias._version,
                None if arena is None else (arena.version, arena.data._version))


def shadow_of(module, kind, code, in_blocked=0, plane=1):
    table = module.__dict__.setdefault('_srgan_shadows', {})
    key = (kind, code, in_blocked, plane)
    found = table.get(key)
    if found is None:
        found = table[key] = Shadow(module, kind, code, in_blocked, plane)
    return found.fresh()


class _PackPlan:
    """The convolution shadows of one arena as ONE launch (``srgan_h_pack_batched``): a device array of jobs, built once from
    the arguments the single-layer packers would get -- the masters live at fixed arena addresses and a shadow keeps its
    buffers, so the table stays valid until the set of shadows changes."""

    def __init__(self, shadows):
        import ctypes
        lib = _lib.library()
        size = lib.srgan_h_pack_job_bytes()
        self.shadows = list(shadows)
        self.signature = tuple((id(s), s.module.weight.data_ptr()) for s in self.shadows)
        host = ctypes.create_string_buffer(size * 5 * max(1, len(self.shadows)))
        count = blocks = 0
        written = ctypes.c_int32()
        device = None
        for shadow in self.shadows:
            weight = shadow.module.weight
            device = weight.device
            if shadow.kind == 'conv3x3':
                k, c, r, s = weight.shape
                for name, transposed, rows, reduced in (('forward', 0, k, c), ('transposed', 1, c, k)):
                    target = shadow._buffer(name, lib.srgan_h_conv_weight_slots(rows, reduced, r, s))
                    taken = lib.srgan_h_pack_job_conv_weights(ctypes.byref(host, count * size), blocks, weight.data_ptr(), target,
                                                              k, c, r, s, transposed, shadow.code)
                    _lib.check(min(taken, 0), 'srgan_h_pack_job_conv_weights')
                    count, blocks = count + 1, blocks + taken
            else:                                           # 'k4s2'
                a, b = weight.shape[0], weight.shape[1]
                for name, direction in (('down', 0), ('up', 1)):
                    target = shadow._buffer(name, lib.srgan_h_k4s2_weight_slots(a, b, direction, shadow.code))
                    taken = lib.srgan_h_pack_job_k4s2_weights(ctypes.byref(host, count * size), blocks, weight.data_ptr(), target,
                                                              a, b, direction, shadow.code, ctypes.byref(written))
                    _lib.check(min(taken, 0), 'srgan_h_pack_job_k4s2_weights')
                    count, blocks = count + written.value, blocks + taken
        self.count, self.blocks = count, blocks
        self.table = None
        if count:
            self.table = torch.frombuffer(bytearray(host.raw[:count * size]), dtype=torch.uint8).to(device)

    def run(self):
        if self.table is not None:
            _call('srgan_h_pack_batched', self.table.data_ptr(), self.count, self.blocks, F._stream())
        for shadow in self.shadows:
            shadow.key = shadow._version()


BATCHED_KINDS = ('conv3x3', 'k4s2')


def refresh(arena):
    """Re-round every shadow of the networks' weights in ``arena`` (called right behind the optimizer update, on its stream:
    whatever orders a later reader behind the update orders it behind the shadows too).  The convolution shadows go as one
    batched launch, the few matrix shadows (linear layers) one by one."""
    shadows = getattr(arena, 'shadows', ())
    if not shadows:
        return
    batched = [s for s in shadows if s.kind in BATCHED_KINDS]
    plan = getattr(arena, '_pack_plan', None)
    if batched and os.environ.get('SRGAN_H_NO_BATCHED_PACK') != '1':
        signature = tuple((id(s), s.module.weight.data_ptr()) for s in batched)
        if plan is None or plan.signature != signature:
            plan = arena._pack_plan = _PackPlan(batched)
        plan.run()
    else:
        for shadow in batched:
            shadow.repack()
    for shadow in shadows:
        if shadow.kind not in BATCHED_KINDS:
            shadow.repack()


# ------------------------------------------------------------------------------------------------ fused layers
def _parameter(p):
    from .nn import parameter_var
    return None if p is None else parameter_var(p)


def conv3x3(x, module, slope=None):
    """``act(conv2d(x, w, b, stride 1, padding 1))`` for a 3x3 ``nn.Conv2d``: slope None = no activation, 0.0 = ReLU (reference
    age/vgg.py:78-82), other = leaky_relu."""
    if tuple(module.kernel_size) != (3, 3) or tuple(module.stride) != (1, 1) or tuple(module.padding) != (1, 1):
        raise ValueError('blocked16.conv3x3: 3x3 / stride 1 / padding 1 convolutions')
    shadow = shadow_of(module, 'conv3x3', x.meta.code)
    epi, slope_value = (1, 1.0 if slope is None else float(slope))
    return _layer(x, module, shadow, False, epi, slope_value, None, True)


def linear(x, module, slope=None):
    """``act(linear(x, w, b))`` on a [N, F] blocked matrix (reference age/vgg.py:33-41,48-53); ``x`` may be the flattened view
    of a [N, C, H, W] tensor: the shadow's columns follow the blocked order."""
    meta = x.meta
    if meta.h != 1 or meta.w != 1:
        raise ValueError('blocked16.linear takes [N, F] (flatten first)')
    shadow = shadow_of(module, 'linear', meta.code, in_blocked=meta.groups * 8, plane=meta.plane)
    return _layer(x, module, shadow, False, 1, 1.0 if slope is None else float(slope), None, True)


def conv4x4s2(x, module, slope=None):
    """``act(conv2d(x, w, b, stride 2, padding 1))`` for a 4x4 ``nn.Conv2d`` (the DCGAN discriminator's stages, reference
    age/models.py:61-73: ``leaky_relu(layer(x), 0.05)``)."""
    if tuple(module.kernel_size) != (4, 4) or tuple(module.stride) != (2, 2) or tuple(module.padding) != (1, 1):
        raise ValueError('blocked16.conv4x4s2: 4x4 / stride 2 / padding 1 convolutions')
    return _layer(x, module, shadow_of(module, 'k4s2', x.meta.code), False, 1, 1.0 if slope is None else float(slope), None, True)


def conv_transpose4x4s2(x, module, slope=None):
    """``act(conv_transpose2d(x, w, b, stride 2, padding 1))`` for a 4x4 ``nn.ConvTranspose2d`` (the DCGAN generator's stages,
    reference age/models.py:37-51)."""
    if tuple(module.kernel_size) != (4, 4) or tuple(module.stride) != (2, 2) or tuple(module.padding) != (1, 1) or \
            tuple(module.output_padding) != (0, 0):
        raise ValueError('blocked16.conv_transpose4x4s2: 4x4 / stride 2 / padding 1 transposed convolutions')
    return _layer(x, module, shadow_of(module, 'k4s2', x.meta.code), False, 1, 1.0 if slope is None else float(slope), None, True)


def seed_conv_transpose(x, module, slope=None):
    """``conv_transpose2d`` of a [N, Cin] code (a 1 x 1 plane) with a full-plane kernel, stride 1, no padding (the generator's
    ``fc``, reference age/models.py:37,47): a linear map onto a [N, Cout, R, S] tensor."""
    if tuple(module.stride) != (1, 1) or tuple(module.padding) != (0, 0) or x.meta.h != 1 or x.meta.w != 1 or x.meta.plane != 1:
        raise ValueError('blocked16.seed_conv_transpose: a stride-1 unpadded transposed convolution of a 1 x 1 plane')
    return _layer(x, module, shadow_of(module, 'linear_t', x.meta.code), False, 1, 1.0 if slope is None else float(slope), None, True)


def _is_transpose(module):
    return isinstance(module, torch.nn.ConvTranspose2d)


def _launch(x, module, shadow, transposed, epi, slope, ref, bias):
    """The contraction launch of ``_layer``: returns (output tensor, its Blocked)."""
    meta, code, stream, device = x.meta, x.meta.code, F._stream(), x.data.device
    bias_ptr = None if bias is None else bias.data.data_ptr()
    if shadow.kind == 'conv3x3':
        k, c = module.weight.shape[0], module.weight.shape[1]
        c_in, c_out = (k, c) if transposed else (c, k)
        if meta.c != c_in:
            raise ValueError(f'blocked16 conv: input has {meta.c} channels, the layer expects {c_in}')
        data = _new(meta.n, c_out, meta.h, meta.w, code, device)
        operand = shadow.transposed if transposed else shadow.forward
        _call('srgan_h_conv3x3', x.data.data_ptr(), operand.data_ptr(), bias_ptr, _ptr(ref), float(slope), epi, data.data_ptr(),
              meta.n, c_in, c_out, c_out, meta.h, meta.w, code, stream)
        return data, Blocked(meta.n, c_out, meta.h, meta.w, code)
    if shadow.kind == 'k4s2':
        a, b = module.weight.shape[0], module.weight.shape[1]
        up = _is_transpose(module) != transposed              # small -> big plane
        if up:
            if meta.c != a:
                raise ValueError(f'blocked16 transposed 4x4 / s2: input has {meta.c} channels, expected {a}')
            data = _new(meta.n, b, 2 * meta.h, 2 * meta.w, code, device)
            _call('srgan_h_conv_transpose4x4s2', x.data.data_ptr(), shadow.up.data_ptr(), bias_ptr, _ptr(ref), float(slope), epi,
                  data.data_ptr(), meta.n, a, b, meta.h, meta.w, code, stream)
            return data, Blocked(meta.n, b, 2 * meta.h, 2 * meta.w, code)
        if meta.c != b or meta.h % 2 or meta.w % 2:
            raise ValueError(f'blocked16 4x4 / s2: input [{meta.c}, {meta.h}, {meta.w}], expected {b} channels on an even plane')
        data = _new(meta.n, a, meta.h // 2, meta.w // 2, code, device)
        _call('srgan_h_conv4x4s2', x.data.data_ptr(), shadow.down.data_ptr(), bias_ptr, _ptr(ref), float(slope), epi,
              data.data_ptr(), meta.n, b, a, meta.h, meta.w, code, stream)
        return data, Blocked(meta.n, a, meta.h // 2, meta.w // 2, code)
    if shadow.kind == 'linear_t':
        c_in, c_out, r, s_ = module.weight.shape
        plane = r * s_
        blocked = (c_out + 7) // 8 * 8 * plane
        if transposed:                                         # [N, Cout, R, S] -> [N, Cin]
            if (meta.c, meta.h, meta.w) != (c_out, r, s_):
                raise ValueError('blocked16 seed transposed convolution (data gradient): unexpected input shape')
            data = _new(meta.n, c_in, 1, 1, code, device)
            _call('srgan_h_gemm', shadow.transposed.data_ptr(), x.data.data_ptr(), None, _ptr(ref), float(slope), epi, data.data_ptr(),
                  c_in, meta.n, blocked, c_in, 0, code, stream)
            return data, Blocked(meta.n, c_in, 1, 1, code)
        if meta.c != c_in:
            raise ValueError(f'blocked16 seed transposed convolution: input has {meta.c} features, expected {c_in}')
        data = _new(meta.n, c_out, r, s_, code, device)
        rows = None if (bias is None or shadow.bias_rows is None) else shadow.bias_rows.data_ptr()
        _call('srgan_h_gemm', shadow.forward.data_ptr(), x.data.data_ptr(), rows, _ptr(ref), float(slope), epi, data.data_ptr(),
              blocked, meta.n, (c_in + 7) // 8 * 8, blocked, blocked, code, stream)
        return data, Blocked(meta.n, c_out, r, s_, code)
    outputs, inputs = module.weight.shape[0], module.weight.numel() // module.weight.shape[0]
    blocked = shadow.in_blocked or (inputs + 7) // 8 * 8
    if transposed:
        if meta.c != outputs:
            raise ValueError(f'blocked16 linear (data gradient): input has {meta.c} features, expected {outputs}')
        data = _new(meta.n, blocked, 1, 1, code, device)
        _call('srgan_h_gemm', shadow.transposed.data_ptr(), x.data.data_ptr(), None, _ptr(ref), float(slope), epi, data.data_ptr(),
              blocked, meta.n, outputs, blocked, 0, code, stream)
        return data, Blocked(meta.n, blocked, 1, 1, code, plane=shadow.plane, real=inputs)
    if meta.c != blocked:
        raise ValueError(f'blocked16 linear: input has {meta.c} (blocked) features, the shadow was built for {blocked}')
    data = _new(meta.n, outputs, 1, 1, code, device)
    _call('srgan_h_gemm', shadow.forward.data_ptr(), x.data.data_ptr(), bias_ptr, _ptr(ref), float(slope), epi, data.data_ptr(),
          outputs, meta.n, blocked, outputs, outputs, code, stream)
    return data, Blocked(meta.n, outputs, 1, 1, code)


def _layer(x, module, shadow, transposed, epi, slope, ref, use_bias):
    """One contraction ``epi(op(x, W) [+ b])`` recorded on the tape.  transposed: the data-gradient direction (x has the
    layer's OUTPUT features).  epi 1: bias + leaky(slope); epi 2: times mask(ref, slope); epi 0: plain."""
    meta = x.meta
    weight = _parameter(module.weight)
    bias = _parameter(module.bias) if (use_bias and module.bias is not None) else None
    data, out_meta = _launch(x, module, shadow, transposed, epi, slope, ref, bias)
    if epi == 1 and slope != 1.0:
        out_meta.mask_ref, out_meta.slope = data, slope            # an activated tensor is its own mask
    elif epi == 2:
        out_meta.mask_ref, out_meta.slope = ref, slope              # a pre-masked gradient: cotangents arrive masked alike

    def backward(s, needs):
        gx = None
        if needs[0]:
            if meta.mask_ref is not None:
                gx = _layer(s, module, shadow, not transposed, 2, meta.slope, meta.mask_ref, False)
            else:
                gx = _layer(s, module, shadow, not transposed, 0, 1.0, None, False)
        if needs[1]:
            if not accumulates_into(weight):
                raise NotImplementedError('the 16-bit path computes weight gradients in plain backward sweeps only '
                                          '(straight into the fp32 gradient arena)')
            x_side, y_side = (s, x) if transposed else (x, s)
            _weight_gradient(shadow, module, x_side, y_side, weight.grad_buffer)
        if needs[2]:
            if not accumulates_into(bias):
                raise NotImplementedError('the 16-bit path computes bias gradients in plain backward sweeps only')
            sm = s.meta
            _call('srgan_h_channel_sums', s.data.data_ptr(), bias.grad_buffer.data_ptr(), sm.n, sm.c, sm.h * sm.w, sm.code,
                  F._stream())
        return gx, None, None
    out = F._out(data, (x, weight, bias), backward, 'h_' + shadow.kind + ('_t' if transposed else ''))
    out.meta = out_meta
    return out


def _weight_gradient(shadow, module, x_side, y_side, into):
    """into (fp32, the arena's gradient view of the layer's weight) += d loss / d W from the tensor at the layer's input side
    and the (pre-masked) gradient at its output side."""
    xm, ym = x_side.meta, y_side.meta
    stream = F._stream()
    if shadow.kind == 'conv3x3':
        _call('srgan_h_conv3x3_wgrad', x_side.data.data_ptr(), y_side.data.data_ptr(), into.data_ptr(), xm.n, xm.c, ym.c, xm.h,
              xm.w, xm.code, stream)
    elif shadow.kind == 'k4s2':
        # conv2d: input side = the big plane; conv_transpose2d: input side = the small plane
        big, small = (y_side, x_side) if _is_transpose(module) else (x_side, y_side)
        _call('srgan_h_k4s2_wgrad', big.data.data_ptr(), small.data.data_ptr(), into.data_ptr(), big.meta.n, big.meta.c, small.meta.c,
              big.meta.h, big.meta.w, 1, xm.code, stream)
    elif shadow.kind == 'linear_t':
        c_in, c_out, r, s_ = module.weight.shape
        plane = r * s_
        blocked = (c_out + 7) // 8 * 8 * plane
        _call('srgan_h_linear_wgrad', y_side.data.data_ptr(), x_side.data.data_ptr(), into.data_ptr(), xm.n, blocked, xm.c,
              c_out * plane, c_in, 1, c_out * plane, plane, 1, xm.code, stream)
    else:
        outputs, inputs = module.weight.shape[0], module.weight.numel() // module.weight.shape[0]
        _call('srgan_h_linear_wgrad', y_side.data.data_ptr(), x_side.data.data_ptr(), into.data_ptr(), xm.n, outputs, xm.c,
              outputs, inputs, inputs, 1, 1, shadow.plane, xm.code, stream)
