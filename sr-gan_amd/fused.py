"""Concat-free DenseNet block: ONE tape node per ``_DenseBlock`` (reference crowd/models.py:335-361), and the
transition's ``norm -> relu -> conv`` as one node on the same kernels (``bn_relu_conv``, crowd/models.py:364-371).

The reference concatenates ``[x, new_features]`` after every layer (quadratic copy traffic) and autograd slices
the gradient back apart.  Here the block owns one ``[N, C0 + L*k, H, W]`` buffer: every layer's 3x3 convolution
writes its ``k`` channels straight into its slice (the conv kernel takes an output batch stride), and the
batch-norm of the next layer reads a channel-slice *view* of the same buffer.  The backward keeps one gradient
buffer of the same shape and walks the layers last-to-first: each layer reads its slice of it, and the gradient
w.r.t. its (view) input is ACCUMULATED into the leading channels by the batch-norm backward epilogue of the
data-gradient kernel (``srgan_conv2d_bwd_data_bnrelu``); weight and batch-norm parameter gradients are accumulated
straight into the network's flat gradient arena.  Neither the normalised / activated tensors nor the gradients
w.r.t. them exist in HBM: the convolution kernels evaluate batch-norm + ReLU in their operand streams (``PROLOGUE``)
and its backward on their way out (``EPILOGUE``).

Second order (the gradient penalty, reference srgan.py:360-375: ``autograd.grad(..., create_graph=True)`` w.r.t.
the block input, then ``.backward()`` of a function of that gradient).  With frozen batch-norm and ReLU the block is
piecewise linear in its input, so its backward B(g) = J^T g is linear in g and depends on the parameters only
through the weights and the batch-norm scales (the ReLU masks are piecewise constant).  The recorded backward
therefore returns a node whose own backward is the *linearised forward* of the block: the adjoint of every
intermediate of B is the corresponding tangent of F, so

  dL/dg       = J v                     (one more concat-free forward pass with mask*scale in place of BN+ReLU)
  dL/dW_conv  = wgrad(tangent at the conv input, first-backward gradient at the conv output)
  dL/dgamma   = inv_std * sum_co W * wgrad(masked, unscaled tangent, ...)        (beta only moves the mask: 0)

and no concatenation, slicing or mask tensors are materialised.  Only the input gradient may be requested from a
recorded backward (exactly the penalty's use); parameter gradients of a recorded backward need the primitive path
(``fused.ENABLED = False``).
"""
import os

import torch

from . import _lib
from . import functional as F
from .tape import Var, Node, grad_enabled, incoming_gradient_is_exclusive, at_sweep_end
from .nn import parameter_var


ENABLED = True      # tests flip this to compare the fused block with the primitive path
# DenseNet transitions evaluated as norm -> relu -> pool -> conv instead of the reference's conv -> pool (crowd/models.py
# _Transition: a 1x1 convolution commutes with the average pooling, so the convolution, its gradients and their
# double-backward forms run on a quarter of the pixels: 8.5 % fewer executed FLOPs per iteration, 75.5 -> 79.2 images/s).
# SRGAN_NO_POOL_FIRST=1 (tests: fused.POOL_FIRST = False) restores the reference order.
POOL_FIRST = not os.environ.get('SRGAN_NO_POOL_FIRST')
PROLOGUE = True     # batch-norm + ReLU evaluated inside the convolution kernels (tests flip this too)
EPILOGUE = True     # batch-norm + ReLU backward evaluated in the epilogue of the data-gradient kernels
# The weight-gradient kernels of a block's backward only feed the optimizer, so they can run on a second stream next to
# the data-gradient chain: the grouped launches of block k then overlap the chain of block k - 1, whose kernels on the small
# planes cannot fill 256 CUs.  The join is deferred to the end of the backward sweep (``tape.at_sweep_end``).  Kernels then
# overlap, so per-kernel timings no longer describe one kernel at a time: bench.py switches this on for the timed region
# (``settings.wgrad_stream`` -> ``fused.WGRAD_STREAM``) and off for the event-bracketed step behind its roofline line.
WGRAD_STREAM = os.environ.get('SRGAN_WGRAD_STREAM', '0') == '1'
_side_streams = {}


def _side_stream(device, main=None):
    """The weight-gradient stream that belongs to ``main`` (the DNN side stream gets its own)."""
    key = (str(device), main.cuda_stream if main is not None else 0)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _launch_on_side_stream(device, tensors, launch):
    """``launch(stream handle)`` on the weight-gradient stream of the current stream, after everything enqueued so far;
    ``tensors`` (read or written by those launches) stay allocated until they have run; the current stream waits for them
    at the end of the backward sweep in progress."""
    main = torch.cuda.current_stream(device)
    side = _side_stream(device, main)
    side.wait_stream(main)
    launch(_lib.stream_handle(side))
    for tensor in tensors:
        if tensor is not None:
            tensor.record_stream(side)
    at_sweep_end(lambda: main.wait_stream(side))


def _ptr(tensor, offset_elements=0):
    return tensor.data_ptr() + 4 * offset_elements


def _desc(n, c, h, w, k, r, s, stride, pad, x_bs=0, y_bs=0):
    oh, ow = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1
    return _lib.ConvDesc(n, c, h, w, k, r, s, stride, stride, pad, pad, oh, ow, x_bs, y_bs, F.COMPUTE_DTYPE)


def _empty(shape, device):
    if F.POISON:
        return torch.full(tuple(shape), float('nan'), dtype=torch.float32, device=device)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _zeros(shape, device):
    """Zero-filled by the library's own kernel (no torch arithmetic on the path; replays correctly inside HIP graphs)."""
    tensor = torch.empty(shape, dtype=torch.float32, device=device)
    F._call('srgan_fill', tensor.data_ptr(), tensor.numel(), 0.0, F._stream())
    return tensor


BATCHED_REDUCE = True   # one batch-norm parameter-sum reduction per block backward instead of one per convolution


def _reduce_plan(layers, n, c0, h, w, growth, buffer_bs, epilogue1, epilogue2, device):
    """Where every fused data-gradient epilogue of the block's backward leaves its per-workgroup batch-norm parameter sums
    inside one scratch tensor, and the device-resident job table of the single reduction that follows.  Cached on the
    block's first layer: the offsets depend only on the geometry, the pointers (frozen statistics, gradient arena) only
    on where the parameters live."""
    key = (n, c0, h, w, epilogue1, epilogue2, str(device),
           layers[0].norm1.weight.grad.data_ptr(), layers[-1].norm2.weight.grad.data_ptr())
    plans = layers[0].__dict__.setdefault('_srgan_reduce_plans', {})      # one per batch size the block is run at
    if key in plans:
        return plans[key]
    lib = _lib.library()
    width = layers[0].conv1.out_channels
    jobs, offsets, total, max_channels, max_tiles = [], [], 0, 0, 0
    for index, layer in enumerate(layers):
        cin = c0 + index * growth
        entry = {}
        for which, enabled, desc, norm, channels in (
                (2, epilogue2, _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs), layer.norm2, width),
                (1, epilogue1, _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0), layer.norm1, cin)):
            if not enabled:
                continue
            tiles = int(lib.srgan_conv2d_bwd_data_bnrelu_tiles(desc))
            if tiles <= 0:
                return None
            inv, _ = norm._inverse_std()
            entry[which] = total
            jobs.append(_lib.BnReduceJob(total, tiles, channels, inv.data.data_ptr(), norm.weight.grad.data_ptr(),
                                         norm.bias.grad.data_ptr()))
            total += 2 * tiles * channels
            max_channels, max_tiles = max(max_channels, channels), max(max_tiles, tiles)
        offsets.append(entry)
    if not jobs:
        return None
    import ctypes
    table = (_lib.BnReduceJob * len(jobs))(*jobs)
    host = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
    plan = dict(key=key, offsets=offsets, total=total, count=len(jobs), max_channels=max_channels, max_tiles=max_tiles,
                jobs=host.to(device), keep=[norm._inverse_std() for layer in layers for norm in (layer.norm1, layer.norm2)])
    plans[key] = plan
    return plan


GROUPED_WGRAD = not os.environ.get('SRGAN_NO_GROUPED_WGRAD')    # all the weight gradients of a block's backward in two launches (one table per kernel size)
IN_PLACE_GRADIENT = not os.environ.get('SRGAN_NO_IN_PLACE_GRADIENT')
# (round 5: 4096 MB -- with the ordered weight gradients a single-problem launch is two launches (workers + finish), so keeping
# the masked tangents of the larger blocks for ONE grouped pair pays at 512 x 512 too: 83.7 vs 83.2 images/s; 800 MB before)
GROUPED_TANGENT_LIMIT = int(os.environ.get('SRGAN_GROUPED_TANGENT_LIMIT_MB', '4096')) << 20   # bytes of masked tangents kept


def _wgrad_plan(layers, n, c0, h, w, growth, buffer_bs, device, tangent=None):
    """Device-resident tables of the two grouped weight-gradient launches of the block's backward (`srgan_wgrad_group_*`):
    slot `index` of the first is layer `index`'s 1x1 convolution (x = the block buffer, gy = its slice of one tensor that
    holds every layer's gradient at conv1's output), of the second its 3x3 convolution (x = its slice of the tensor that
    holds every layer's conv1 output, gy = its channel slice of the gradient buffer).  Offsets are relative to those
    per-step tensors, so the tables are built once per (block, batch size).  None when a geometry has no fused form.

    With ``tangent`` (the plan of `_tangent_plan`): the DOUBLE backward's tables instead -- plain weight gradients of the
    masked tangents (1x1: x = layer's slice of one flat tensor of all u1, gy as above; 3x3: x = its slice of the tensor of all
    u2, gy = its channel slice of the first backward's gradient buffer) that accumulate into the per-step buffer of q at
    the tangent plan's offsets."""
    import ctypes
    key = (n, c0, h, w, str(device), tangent is not None, layers[0].conv1.weight.grad.data_ptr(),
           layers[-1].conv2.weight.grad.data_ptr())
    plans = layers[0].__dict__.setdefault('_srgan_wgrad_plans', {})
    if key in plans:
        return plans[key]
    lib = _lib.library()
    width, hw = layers[0].conv1.out_channels, h * w
    plan = {'keep': []}
    u1_offsets, u1_at = [], 0
    for index in range(len(layers)):
        u1_offsets.append(u1_at)
        u1_at += n * (c0 + index * growth) * hw
    plan['u1_offsets'], plan['u1_total'] = u1_offsets, u1_at
    for size in (1, 3):
        slots = (ctypes.c_byte * (128 * len(layers)))()
        grid_x = grid_y = ragged = 0            # (ragged: the OR of the kernel variants the problems were planned for)
        co_ci_taps = elements = 0
        partial_at = 0                  # the problems' partial-tile regions, back to back in the stream's workspace (floats)
        # what the group holds (sum of CO x CI x taps): the plan shares the launch's workgroups out by work
        weights = sum(width * (c0 + i * growth) if size == 1 else growth * width * 9 for i in range(len(layers)))
        for index, layer in enumerate(layers):
            cin = c0 + index * growth
            if size == 1:
                norm, gw = layer.norm1, layer.conv1.weight.grad
                if tangent is None:
                    desc, x_offset = _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0), 0
                else:
                    desc, x_offset = _desc(n, cin, h, w, width, 1, 1, 1, 0), u1_offsets[index]
                gy_offset = index * n * width * hw
                co_ci_taps += width * cin
                elements += (cin + width) * n * hw
            else:
                desc, norm, gw = _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs), layer.norm2, layer.conv2.weight.grad
                x_offset, gy_offset = index * n * width * hw, cin * hw
                co_ci_taps += growth * width * 9
                elements += (width + growth) * n * hw
            bn = None
            if tangent is None:
                inv, mean = norm._inverse_std()
                plan['keep'].append((inv, mean))
                bn = _lib.BnRelu(mean.data.data_ptr(), inv.data.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr())
            gx, gy, rg, partial = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
            status = lib.srgan_wgrad_group_plan(desc, bn, x_offset, gy_offset, gw.data_ptr() if tangent is None else None,
                                                0 if tangent is None else tangent['offsets'][index][0 if size == 1 else 1],
                                                len(layers), weights, partial_at, ctypes.byref(slots, 128 * index), ctypes.byref(gx),
                                                ctypes.byref(gy), ctypes.byref(rg), ctypes.byref(partial))
            if status != 0:
                plans[key] = None
                return None
            partial_at += partial.value
            grid_x, grid_y, ragged = max(grid_x, gx.value), max(grid_y, gy.value), ragged | rg.value
        table = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(device)
        plan[size] = dict(table=table, count=len(layers), grid_x=grid_x, grid_y=grid_y, ragged=ragged, co_ci_taps=co_ci_taps,
                          pixels=n * hw, elements=elements, partial_floats=partial_at)
    plans[key] = plan
    return plan


def _run_wgrad_group(group, size, fused_bn, x_base, gy_base, gw_base, stream):
    F._call('srgan_wgrad_group_run', group['table'].data_ptr(), group['count'], size, group['grid_x'], group['grid_y'],
            group['ragged'], 1 if fused_bn else 0, x_base.data_ptr(), gy_base.data_ptr(),
            gw_base.data_ptr() if gw_base is not None else None, group['co_ci_taps'], group['pixels'], group['elements'],
            group['partial_floats'], stream)


def _tangent_plan(layers, c0, growth, device):
    """Device-resident table of the grouped tangent-weight launches of the block's double backward: slots 2 * index and
    2 * index + 1 are layer `index`'s 1x1 and 3x3 convolution; `offsets` are their places inside the per-step buffers of
    scaled weights / weight gradients q."""
    import ctypes
    key = (c0, str(device), layers[0].conv1.weight.data_ptr(), layers[-1].conv2.weight.grad.data_ptr())
    plans = layers[0].__dict__.setdefault('_srgan_tangent_plans', {})
    if key in plans:
        return plans[key]
    lib = _lib.library()
    slots = (ctypes.c_byte * (64 * 2 * len(layers)))()
    offsets, at, max_inner, max_co, keep = [], 0, 0, 0, []
    for index, layer in enumerate(layers):
        pair = []
        for which, (conv, norm, taps) in enumerate(((layer.conv1, layer.norm1, 1), (layer.conv2, layer.norm2, 9))):
            inv, _ = norm._inverse_std()
            keep.append(inv)
            co, ci = conv.weight.shape[0], conv.weight.shape[1]
            _lib.check(lib.srgan_bn_conv_tangent_weights_job(conv.weight.data_ptr(), inv.data.data_ptr(), norm.weight.data_ptr(),
                                                             conv.weight.grad.data_ptr(), norm.weight.grad.data_ptr(), at, co,
                                                             ci, taps, ctypes.byref(slots, 64 * (2 * index + which))),
                       'srgan_bn_conv_tangent_weights_job')
            pair.append(at)
            at += conv.weight.numel()
            max_inner, max_co = max(max_inner, ci * taps), max(max_co, co)
        offsets.append(tuple(pair))
    table = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(device)
    plans[key] = dict(table=table, count=2 * len(layers), offsets=offsets, total=at, max_inner=max_inner, max_co=max_co,
                      keep=keep)
    return plans[key]


def _block_parameters(layers):
    """The block's parameters in ``layer.parameters()`` order (the module tree is fixed: enumerated once)."""
    cached = layers[0].__dict__.get('_srgan_block_parameters')
    if cached is None or cached[0] != len(layers):
        cached = layers[0].__dict__['_srgan_block_parameters'] = (len(layers), [p for layer in layers for p in layer.parameters()])
    return cached[1]


def dense_block(x, layers):
    """``layers``: the block's ``_DenseLayer`` modules (norm1, conv1, norm2, conv2)."""
    n, c0, h, w = x.shape
    hw = h * w
    growth = layers[0].conv2.out_channels
    total = c0 + len(layers) * growth
    device = x.data.device
    stream = F._stream()
    buffer = _empty((n, total, h, w), device)
    buffer_bs = total * hw
    saved = []
    parameter_vars = [parameter_var(p) for p in _block_parameters(layers)]
    requires = grad_enabled() and (x.requires_grad or any(v.requires_grad for v in parameter_vars))
    train = requires and any(v.requires_grad for v in parameter_vars)     # t1 / t2 are only read by weight gradients
    width = layers[0].conv1.out_channels
    lib = _lib.library()
    # norm -> relu -> conv with the normalisation evaluated inside the convolution kernels (no t1 / t2 tensors at
    # all) when the block's geometry has the fused forms; otherwise the activations are materialised.
    probe1 = _desc(n, c0, h, w, width, 1, 1, 1, 0, buffer_bs, 0)
    probe2 = _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs)
    prologue = PROLOGUE and all(lib.srgan_conv2d_bnrelu_supported(d, kind) for d in (probe1, probe2) for kind in (0, 2))
    # backward of norm1 -> relu1 -> conv1 w.r.t. the block buffer in one kernel (batch-norm backward in the epilogue
    # of the data gradient, accumulated straight into the gradient buffer)
    epilogue1 = EPILOGUE and prologue and bool(lib.srgan_conv2d_bnrelu_supported(
        _desc(n, c0, h, w, width, 1, 1, 1, 0, buffer_bs, 0), 1))
    epilogue2 = EPILOGUE and prologue and bool(lib.srgan_conv2d_bnrelu_supported(probe2, 1))

    def bn_struct(norm):
        inv, mean = norm._inverse_std()
        return _lib.BnRelu(mean.data.data_ptr(), inv.data.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr())

    # Small planes make the convolutions split K over the grid (fp32 atomics into a zeroed output): one zero-fill for
    # ALL the layers' outputs instead of one fill launch per convolution.
    zero_b1 = zero_new = False
    b1_all = None
    if prologue:
        last_cin = c0 + (len(layers) - 1) * growth
        # (round 5: on a stream with a workspace a K split finishes in a fixed order and STORES whole sums -- no zero-fill)
        atomics = not lib.srgan_split_is_ordered(stream)
        zero_b1 = atomics and lib.srgan_conv2d_fwd_bnrelu_splits(_desc(n, last_cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0)) > 1
        zero_new = atomics and lib.srgan_conv2d_fwd_bnrelu_splits(probe2) > 1
        if zero_new:
            F._call('srgan_fill', buffer.data_ptr(), buffer.numel(), 0.0, stream)
        if zero_b1:
            b1_all = _zeros((len(layers), n, width, h, w), device)
        elif requires and GROUPED_WGRAD:
            # one tensor: the grouped weight gradients index it (peak memory: the slices cannot be freed layer by layer in
            # the backward, L x n x width x h x w x 4 B more than per-layer tensors -- paid only while the grouped form is on)
            b1_all = _empty((len(layers), n, width, h, w), device)
    F._call('srgan_copy_channels', x.data.data_ptr(), c0, 0, buffer.data_ptr(), total, 0, c0, n, hw, 0, stream)
    forward1 = 'srgan_conv2d_fwd_bnrelu_into_zeros' if zero_b1 else 'srgan_conv2d_fwd_bnrelu'
    forward2 = 'srgan_conv2d_fwd_bnrelu_into_zeros' if zero_new else 'srgan_conv2d_fwd_bnrelu'
    for index, layer in enumerate(layers):
        cin = c0 + index * growth
        b1 = b1_all[index] if b1_all is not None else _empty((n, width, h, w), device)
        if prologue:
            F._call(forward1, _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0), buffer.data_ptr(),
                    bn_struct(layer.norm1), layer.conv1.weight.data_ptr(), None, b1.data_ptr(), stream)
            F._call(forward2, _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs), b1.data_ptr(),
                    bn_struct(layer.norm2), layer.conv2.weight.data_ptr(), None, _ptr(buffer, cin * hw), stream)
            if requires:
                saved.append([None, b1, None])
            continue
        inv1, mean1 = layer.norm1._inverse_std()
        inv2, mean2 = layer.norm2._inverse_std()
        t1 = _empty((n, cin, h, w), device)
        F._call('srgan_chan_affine_act_strided', buffer.data_ptr(), mean1.data.data_ptr(), inv1.data.data_ptr(),
                layer.norm1.weight.data_ptr(), layer.norm1.bias.data_ptr(), None, 1, t1.data_ptr(), n, cin, hw,
                buffer_bs, 0, 0, 0, stream)
        F._call('srgan_conv2d_fwd', _desc(n, cin, h, w, width, 1, 1, 1, 0), t1.data_ptr(), layer.conv1.weight.data_ptr(),
                None, b1.data_ptr(), 0, stream)
        t2 = _empty((n, width, h, w), device)
        F._call('srgan_chan_affine_act', b1.data_ptr(), mean2.data.data_ptr(), inv2.data.data_ptr(),
                layer.norm2.weight.data_ptr(), layer.norm2.bias.data_ptr(), None, 1, t2.data_ptr(), n, width, hw, stream)
        F._call('srgan_conv2d_fwd', _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs), t2.data_ptr(),
                layer.conv2.weight.data_ptr(), None, _ptr(buffer, cin * hw), 0, stream)
        if requires:
            saved.append([t1 if train else None, b1, t2 if train else None])

    out = Var(buffer, requires_grad=requires)
    if not requires:
        return out

    def norm_pointers(norm):
        inv, mean = norm._inverse_std()
        return mean.data.data_ptr(), inv.data.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr()

    def backward(g, needs):
        recorded = grad_enabled()
        want_params = any(needs[1:])
        if recorded and want_params:
            raise NotImplementedError('a recorded (create_graph) backward of the fused dense block yields the input '
                                      'gradient only; set srgan_amd.fused.ENABLED = False for parameter gradients of '
                                      'a recorded backward')
        stream = F._stream()
        if IN_PLACE_GRADIENT and not recorded and incoming_gradient_is_exclusive() and g.data.is_contiguous():
            gbuf = g.data                         # nobody else reads it: the layers accumulate into it in place
        else:
            gbuf = _empty(g.shape, device)        # private copy: the incoming gradient may be shared
            F._call('srgan_ew_unary', F.U_COPY, g.data.data_ptr(), gbuf.data_ptr(), gbuf.numel(), 0.0, 0.0, stream)
        kept = [None] * len(layers)               # per layer, for the double backward: (gradient at conv1's output, b1)
        # the batch-norm parameter sums of every fused epilogue land in one scratch tensor, reduced by ONE launch at the end
        plan = scratch = None
        if BATCHED_REDUCE and want_params and (epilogue1 or epilogue2):
            plan = _reduce_plan(layers, n, c0, h, w, growth, buffer_bs, epilogue1, epilogue2, device)
            if plan is not None:
                scratch = _empty((plan['total'],), device)
        # (not while a HIP graph is being captured: a captured fork / join per layer replayed slower and, together with the
        # DNN side stream, crashed the runtime)
        side_ok = WGRAD_STREAM and want_params and prologue and not recorded and not torch.cuda.is_current_stream_capturing()
        grouped = None
        if GROUPED_WGRAD and want_params and prologue and not recorded and b1_all is not None:
            grouped = _wgrad_plan(layers, n, c0, h, w, growth, buffer_bs, device)
        overlap = side_ok and grouped is None     # no grouped form for this geometry: the per-layer launches go to the side stream
        # (recorded: the double backward's grouped weight gradients index the kept gradients through one tensor, too)
        one_tensor = grouped is not None or (recorded and GROUPED_WGRAD and prologue)
        g_b1_all = _empty((len(layers), n, layers[0].conv1.out_channels, h, w), device) if one_tensor else None
        if overlap:
            main = torch.cuda.current_stream(device)
            side = _side_stream(device, main)
            wstream = _lib.stream_handle(side)
            alive = []                            # tensors the side stream still reads
        else:
            wstream = stream
        for index in range(len(layers) - 1, -1, -1):
            layer = layers[index]
            t1, b1, t2 = saved[index]
            cin = c0 + index * growth
            width = layer.conv1.out_channels
            mean1, inv1, gamma1, beta1 = norm_pointers(layer.norm1)
            mean2, inv2, gamma2, beta2 = norm_pointers(layer.norm2)
            g_new = _ptr(gbuf, cin * hw)                                  # [N, growth, H, W] view, batch stride buffer_bs
            desc2 = _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs)
            if overlap:
                side.wait_stream(main)            # this layer's slice of the gradient buffer is final
            if grouped is not None:
                pass                              # every weight gradient of the block: two launches after the loop
            elif want_params and prologue:
                F._call('srgan_conv2d_bwd_weight_bnrelu', desc2, b1.data_ptr(), bn_struct(layer.norm2), g_new,
                        layer.conv2.weight.grad.data_ptr(), 1, wstream)
            elif want_params:
                F._call('srgan_conv2d_bwd_weight', desc2, t2.data_ptr(), g_new, layer.conv2.weight.grad.data_ptr(), 1, 0,
                        stream)
            g_b1 = g_b1_all[index] if g_b1_all is not None else _empty(b1.shape, device)
            if epilogue2 and plan is not None:
                F._call('srgan_conv2d_bwd_data_bnrelu_partials', desc2, g_new, layer.conv2.weight.data_ptr(),
                        bn_struct(layer.norm2), b1.data_ptr(), g_b1.data_ptr(), _ptr(scratch, plan['offsets'][index][2]), 0, stream)
            elif epilogue2:
                F._call('srgan_conv2d_bwd_data_bnrelu', desc2, g_new, layer.conv2.weight.data_ptr(),
                        bn_struct(layer.norm2), b1.data_ptr(), g_b1.data_ptr(),
                        layer.norm2.weight.grad.data_ptr() if want_params else None,
                        layer.norm2.bias.grad.data_ptr() if want_params else None, 0, stream)
            else:
                g_t2 = _empty(b1.shape, device)
                F._call('srgan_conv2d_bwd_data', desc2, g_new, layer.conv2.weight.data_ptr(), None, g_t2.data_ptr(), 0, 0,
                        stream)
                F._call('srgan_bn_act_bwd', g_t2.data_ptr(), b1.data_ptr(), mean2, inv2, gamma2, beta2, 1,
                        g_b1.data_ptr(), layer.norm2.weight.grad.data_ptr() if want_params else None,
                        layer.norm2.bias.grad.data_ptr() if want_params else None, n, width, hw, 0, 0, 0, 0, 0, stream)
                del g_t2
            desc1 = _desc(n, cin, h, w, width, 1, 1, 1, 0)
            if overlap:
                side.wait_stream(main)            # g_b1 is complete
                alive.extend((g_b1, b1))
            if grouped is not None:
                pass
            elif want_params and prologue:
                F._call('srgan_conv2d_bwd_weight_bnrelu', _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0),
                        buffer.data_ptr(), bn_struct(layer.norm1), g_b1.data_ptr(), layer.conv1.weight.grad.data_ptr(), 1,
                        wstream)
            elif want_params:
                F._call('srgan_conv2d_bwd_weight', desc1, t1.data_ptr(), g_b1.data_ptr(),
                        layer.conv1.weight.grad.data_ptr(), 1, 0, stream)
            if epilogue1 and plan is not None:
                F._call('srgan_conv2d_bwd_data_bnrelu_partials', _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0),
                        g_b1.data_ptr(), layer.conv1.weight.data_ptr(), bn_struct(layer.norm1), buffer.data_ptr(),
                        gbuf.data_ptr(), _ptr(scratch, plan['offsets'][index][1]), 1, stream)
            elif epilogue1:
                F._call('srgan_conv2d_bwd_data_bnrelu', _desc(n, cin, h, w, width, 1, 1, 1, 0, buffer_bs, 0),
                        g_b1.data_ptr(), layer.conv1.weight.data_ptr(), bn_struct(layer.norm1), buffer.data_ptr(),
                        gbuf.data_ptr(), layer.norm1.weight.grad.data_ptr() if want_params else None,
                        layer.norm1.bias.grad.data_ptr() if want_params else None, 1, stream)
            else:
                g_t1 = _empty((n, cin, h, w), device)
                F._call('srgan_conv2d_bwd_data', desc1, g_b1.data_ptr(), layer.conv1.weight.data_ptr(), None,
                        g_t1.data_ptr(), 0, 0, stream)
                # batch-norm 1 backward: parameter gradients, and the gradient w.r.t. the layer's (view) input
                # accumulated into the leading channels of the gradient buffer, in one pass
                F._call('srgan_bn_act_bwd', g_t1.data_ptr(), buffer.data_ptr(), mean1, inv1, gamma1, beta1, 1,
                        gbuf.data_ptr(), layer.norm1.weight.grad.data_ptr() if want_params else None,
                        layer.norm1.bias.grad.data_ptr() if want_params else None, n, cin, hw, 0, buffer_bs, buffer_bs,
                        1, 0, stream)
            if recorded:
                # The forward node itself may still be back-propagated later (the penalty also depends on the
                # parameters through the forward activations), so nothing of `saved` is released here; the double
                # backward holds its own references.
                kept[index] = (g_b1, b1)
            else:
                saved[index] = None
        if grouped is not None:
            # (the gradient slices of gbuf the 3x3 group reads are final: later layers only wrote channels below them)
            def run_groups(on):
                _run_wgrad_group(grouped[3], 3, True, b1_all, gbuf, None, on)
                _run_wgrad_group(grouped[1], 1, True, buffer, g_b1_all, None, on)
            if side_ok:
                _launch_on_side_stream(device, (b1_all, gbuf, buffer, g_b1_all), run_groups)
            else:
                run_groups(stream)
        if plan is not None:
            F._call('srgan_bn_partial_reduce_batched', plan['jobs'].data_ptr(), plan['count'], plan['max_channels'],
                    plan['max_tiles'], scratch.data_ptr(), stream)
        if overlap:
            main.wait_stream(side)                # the weight gradients are in the arena before anything consumes it
            del alive
        gx = None
        if needs[0]:
            gx_data = _empty((n, c0, h, w), device)
            F._call('srgan_copy_channels', gbuf.data_ptr(), total, 0, gx_data.data_ptr(), c0, 0, c0, n, hw, 0, stream)
            gx = Var(gx_data)
            if recorded:
                gx.requires_grad = True
                gx.node = Node((g,) + tuple(parameter_vars), lambda v, needs2: double_backward(v, needs2, gbuf, kept, g_b1_all),
                               'dense_block_backward')
        return (gx,) + (None,) * len(parameter_vars)

    def double_backward(v, needs2, gbuf, kept, g_b1_all=None):
        """Backward of the recorded backward: v = dL/d(input gradient).  Returns dL/dg (the linearised forward of the
        block applied to v) and accumulates the weight / batch-norm-scale gradients into the gradient arena."""
        if grad_enabled():
            raise NotImplementedError('third-order differentiation of the fused dense block')
        stream = F._stream()
        want_params = any(needs2[1:])
        vbuf = _empty((n, total, h, w), device)
        F._call('srgan_copy_channels', v.data.data_ptr(), c0, 0, vbuf.data_ptr(), total, 0, c0, n, hw, 0, stream)
        # the weight gradients w.r.t. the scaled weights (q1, q2 of every layer) land in ONE pre-zeroed buffer, so the
        # weight-gradient kernels accumulate into it without a zero-fill launch of their own (2 per layer otherwise)
        # Per weight element: the scaled weights of EVERY layer in one launch up front, the weight / batch-norm-scale
        # gradients from the q of every layer in one launch at the end (`srgan_bn_conv_tangent_weights_grouped`); the
        # weight gradients q themselves accumulate into ONE pre-zeroed buffer (no zero-fill launch of their own).
        tangent = _tangent_plan(layers, c0, growth, device)
        scaled_all = _empty((tangent['total'],), device)
        q_all = _zeros((tangent['total'],), device) if want_params else None
        F._call('srgan_bn_conv_tangent_weights_grouped', tangent['table'].data_ptr(), tangent['count'], tangent['max_inner'],
                tangent['max_co'], scaled_all.data_ptr(), None, stream)
        # the plain weight gradients q of every layer: two grouped launches at the end, reading the masked tangents of all
        # layers from two tensors (the per-layer launches otherwise)
        grouped = None
        if GROUPED_WGRAD and want_params and g_b1_all is not None:
            grouped = _wgrad_plan(layers, n, c0, h, w, growth, buffer_bs, device, tangent)
            # (keeping every layer's u1 costs a second trip through HBM for it: worth it while the launches, not the
            # bytes, bound the pass -- measured: a gain at 224 x 224, a loss for the larger blocks of 512 x 512)
            if grouped is not None and 4 * grouped['u1_total'] > GROUPED_TANGENT_LIMIT:
                grouped = None
        u1_all = _empty((grouped['u1_total'],), device) if grouped is not None else None
        u2_all = _empty((len(layers), n, layers[0].conv1.out_channels, h, w), device) if grouped is not None else None
        for index, layer in enumerate(layers):
            g_b1, b1 = kept[index]
            cin = c0 + index * growth
            width = layer.conv1.out_channels
            mean1, inv1, gamma1, beta1 = norm_pointers(layer.norm1)
            mean2, inv2, gamma2, beta2 = norm_pointers(layer.norm2)
            at1, at2 = tangent['offsets'][index]
            desc1 = _desc(n, cin, h, w, width, 1, 1, 1, 0)
            desc2 = _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs)
            # ---- batch-norm 1 + ReLU, linearised: masked (unscaled) tangent; the scale is in the weights
            u1 = _empty((n, cin, h, w), device) if grouped is None else None
            u1_ptr = u1.data_ptr() if grouped is None else _ptr(u1_all, grouped['u1_offsets'][index])
            F._call('srgan_bn_act_bwd', vbuf.data_ptr(), buffer.data_ptr(), mean1, inv1, gamma1, beta1, 1, u1_ptr,
                    None, None, n, cin, hw, buffer_bs, buffer_bs, 0, 0, 1, stream)
            if want_params and grouped is None:
                F._call('srgan_conv2d_bwd_weight', desc1, u1_ptr, g_b1.data_ptr(), _ptr(q_all, at1), 1, 0, stream)
            b1_tangent = _empty(b1.shape, device)
            F._call('srgan_conv2d_fwd', desc1, u1_ptr, _ptr(scaled_all, at1), None, b1_tangent.data_ptr(), 0, stream)
            del u1
            # ---- batch-norm 2 + ReLU, linearised
            u2 = u2_all[index] if grouped is not None else _empty(b1.shape, device)
            F._call('srgan_bn_act_bwd', b1_tangent.data_ptr(), b1.data_ptr(), mean2, inv2, gamma2, beta2, 1, u2.data_ptr(),
                    None, None, n, width, hw, 0, 0, 0, 0, 1, stream)
            if want_params and grouped is None:
                F._call('srgan_conv2d_bwd_weight', desc2, u2.data_ptr(), _ptr(gbuf, cin * hw), _ptr(q_all, at2), 1, 0, stream)
            F._call('srgan_conv2d_fwd', desc2, u2.data_ptr(), _ptr(scaled_all, at2), None, _ptr(vbuf, cin * hw), 0, stream)
            kept[index] = None
        def run_parameter_part(on):
            if grouped is not None:
                _run_wgrad_group(grouped[1], 1, False, u1_all, g_b1_all, q_all, on)
                _run_wgrad_group(grouped[3], 3, False, u2_all, gbuf, q_all, on)
            # dL/dW += q * a;  dL/dgamma += inv_std * sum_co W * q   (a = inv_std * gamma)
            F._call('srgan_bn_conv_tangent_weights_grouped', tangent['table'].data_ptr(), tangent['count'],
                    tangent['max_inner'], tangent['max_co'], None, q_all.data_ptr(), on)
        if want_params and grouped is not None and WGRAD_STREAM and not torch.cuda.is_current_stream_capturing():
            # nothing in the rest of the sweep reads q or the arena: the whole parameter part runs next to the chain
            _launch_on_side_stream(device, (u1_all, u2_all, g_b1_all, gbuf, q_all, scaled_all), run_parameter_part)
        elif want_params:
            run_parameter_part(stream)
        return (Var(vbuf) if needs2[0] else None,) + (None,) * len(parameter_vars)

    out.node = Node((x,) + tuple(parameter_vars), backward, 'dense_block')
    return out


def bn_relu_conv(x, norm, conv):
    """``conv(relu(norm(x)))`` for a 1x1 convolution as ONE tape node on the fused kernels of the dense layers (the
    DenseNet transitions, reference crowd/models.py:364-371): the activated tensor is never materialised in either
    direction -- forward with the normalisation in the operand stream, weight gradient likewise, data gradient with the
    batch-norm backward in its epilogue.  First order, recorded backward (input gradient only) and its double backward
    (linearised forward on the masked tangent) follow the dense block above.  Returns None when the geometry has no fused
    form (the caller then composes the primitive ops)."""
    n, cin, h, w = x.shape
    if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.bias is not None:
        return None
    cout, hw = conv.out_channels, h * w
    lib = _lib.library()
    desc = _desc(n, cin, h, w, cout, 1, 1, 1, 0)
    if not (PROLOGUE and EPILOGUE and all(lib.srgan_conv2d_bnrelu_supported(desc, kind) for kind in (0, 1, 2))):
        return None
    device = x.data.device
    parameter_vars = [parameter_var(p) for p in (norm.weight, norm.bias, conv.weight)]
    requires = grad_enabled() and (x.requires_grad or any(v.requires_grad for v in parameter_vars))

    def bn_struct():
        inv, mean = norm._inverse_std()
        return _lib.BnRelu(mean.data.data_ptr(), inv.data.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr())

    y = _empty((n, cout, h, w), device)
    F._call('srgan_conv2d_fwd_bnrelu', desc, x.data.data_ptr(), bn_struct(), conv.weight.data_ptr(), None, y.data_ptr(),
            F._stream())
    out = Var(y, requires_grad=requires)
    if not requires:
        return out

    def backward(g, needs):
        recorded = grad_enabled()
        want_params = any(needs[1:])
        if recorded and want_params:
            raise NotImplementedError('a recorded backward of the fused norm -> relu -> conv yields the input gradient only')
        stream = F._stream()
        if want_params:
            F._call('srgan_conv2d_bwd_weight_bnrelu', desc, x.data.data_ptr(), bn_struct(), g.data.data_ptr(),
                    conv.weight.grad.data_ptr(), 1, stream)
        gx = None
        if needs[0] or want_params:
            gx_data = _empty((n, cin, h, w), device)
            F._call('srgan_conv2d_bwd_data_bnrelu', desc, g.data.data_ptr(), conv.weight.data_ptr(), bn_struct(),
                    x.data.data_ptr(), gx_data.data_ptr(), norm.weight.grad.data_ptr() if want_params else None,
                    norm.bias.grad.data_ptr() if want_params else None, 0, stream)
            if needs[0]:
                gx = Var(gx_data)
                if recorded:
                    gx.requires_grad = True
                    gx.node = Node((g,) + tuple(parameter_vars), lambda v, needs2: double_backward(v, needs2, g),
                                   'bn_relu_conv_backward')
        return (gx,) + (None,) * len(parameter_vars)

    def double_backward(v, needs2, g):
        """v = dL/d(input gradient): dL/dg = conv(mask * v, W * a); dL/dW += wgrad(mask * v, g) * a; dL/dgamma +=
        inv_std * sum_co W * wgrad(mask * v, g)."""
        if grad_enabled():
            raise NotImplementedError('third-order differentiation of the fused norm -> relu -> conv')
        stream = F._stream()
        want_params = any(needs2[1:])
        inv, mean = norm._inverse_std()
        u = _empty((n, cin, h, w), device)
        F._call('srgan_bn_act_bwd', v.data.data_ptr(), x.data.data_ptr(), mean.data.data_ptr(), inv.data.data_ptr(),
                norm.weight.data_ptr(), norm.bias.data_ptr(), 1, u.data_ptr(), None, None, n, cin, hw, 0, 0, 0, 0, 1, stream)
        q = None
        if want_params:
            q = _zeros(tuple(conv.weight.shape), device)
            F._call('srgan_conv2d_bwd_weight', desc, u.data_ptr(), g.data.data_ptr(), q.data_ptr(), 1, 0, stream)
        scaled = _empty(conv.weight.shape, device)
        F._call('srgan_bn_conv_tangent_weights', conv.weight.data_ptr(), q.data_ptr() if want_params else None,
                inv.data.data_ptr(), norm.weight.data_ptr(), scaled.data_ptr(),
                conv.weight.grad.data_ptr() if want_params else None,
                norm.weight.grad.data_ptr() if want_params else None, cout, cin, 1, stream)
        tangent = None
        if needs2[0]:
            tangent = _empty((n, cout, h, w), device)
            F._call('srgan_conv2d_fwd', desc, u.data_ptr(), scaled.data_ptr(), None, tangent.data_ptr(), 0, stream)
            tangent = Var(tangent)
        return (tangent,) + (None,) * len(parameter_vars)

    out.node = Node((x,) + tuple(parameter_vars), backward, 'bn_relu_conv')
    return out
