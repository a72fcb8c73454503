"""Concat-free DenseNet block: ONE tape node per ``_DenseBlock`` (reference crowd/models.py:335-361).

The reference concatenates ``[x, new_features]`` after every layer (quadratic copy traffic) and autograd slices
the gradient back apart.  Here the block owns one ``[N, C0 + L*k, H, W]`` buffer: every layer's 3x3 convolution
writes its ``k`` channels straight into its slice (the conv kernel takes an output batch stride), and the
batch-norm of the next layer reads a channel-slice *view* of the same buffer.  The backward keeps one gradient
buffer of the same shape and walks the layers last-to-first: each layer reads its slice of it, and the gradient
w.r.t. its (view) input is ACCUMULATED into the leading channels by the batch-norm backward kernel; weight and
batch-norm parameter gradients are accumulated straight into the network's flat gradient arena.

This node is first-order only.  When a forward is going to be differentiated twice (the gradient penalty's
discriminator pass) the modules are run under ``tape.higher_order()`` and use the primitive ops instead.
"""
import torch

from . import _lib
from . import functional as F
from .tape import Var, Node, grad_enabled, higher_order_enabled
from .nn import parameter_var


ENABLED = True      # tests flip this to compare the fused block with the primitive path


def _ptr(tensor, offset_elements=0):
    return tensor.data_ptr() + 4 * offset_elements


def _desc(n, c, h, w, k, r, s, stride, pad, x_bs=0, y_bs=0):
    oh, ow = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1
    return _lib.ConvDesc(n, c, h, w, k, r, s, stride, stride, pad, pad, oh, ow, x_bs, y_bs)


def dense_block(x, layers):
    """``layers``: the block's ``_DenseLayer`` modules (norm1, conv1, norm2, conv2)."""
    n, c0, h, w = x.shape
    hw = h * w
    growth = layers[0].conv2.out_channels
    total = c0 + len(layers) * growth
    device = x.data.device
    stream = F._stream()
    lib = _lib.library()
    buffer = torch.empty((n, total, h, w), dtype=torch.float32, device=device)
    buffer_bs = total * hw
    F._call('srgan_copy_channels', x.data.data_ptr(), c0, 0, buffer.data_ptr(), total, 0, c0, n, hw, 0, stream)
    saved = []
    train = parameter_var(layers[0].conv1.weight).requires_grad and grad_enabled()
    for index, layer in enumerate(layers):
        cin = c0 + index * growth
        inv1, mean1 = layer.norm1._inverse_std()
        inv2, mean2 = layer.norm2._inverse_std()
        width = layer.conv1.out_channels
        t1 = torch.empty((n, cin, h, w), dtype=torch.float32, device=device)
        F._call('srgan_chan_affine_act_strided', buffer.data_ptr(), mean1.data.data_ptr(), inv1.data.data_ptr(),
                layer.norm1.weight.data_ptr(), layer.norm1.bias.data_ptr(), None, 1, t1.data_ptr(), n, cin, hw,
                buffer_bs, 0, 0, 0, stream)
        b1 = torch.empty((n, width, h, w), dtype=torch.float32, device=device)
        F._call('srgan_conv2d_fwd', _desc(n, cin, h, w, width, 1, 1, 1, 0), t1.data_ptr(), layer.conv1.weight.data_ptr(),
                None, b1.data_ptr(), 0, stream)
        t2 = torch.empty((n, width, h, w), dtype=torch.float32, device=device)
        F._call('srgan_chan_affine_act', b1.data_ptr(), mean2.data.data_ptr(), inv2.data.data_ptr(),
                layer.norm2.weight.data_ptr(), layer.norm2.bias.data_ptr(), None, 1, t2.data_ptr(), n, width, hw, stream)
        F._call('srgan_conv2d_fwd', _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs), t2.data_ptr(),
                layer.conv2.weight.data_ptr(), None, _ptr(buffer, cin * hw), 0, stream)
        saved.append((t1, b1, t2) if (grad_enabled()) else None)

    parameter_vars = [parameter_var(p) for layer in layers for p in layer.parameters()]
    requires = grad_enabled() and (x.requires_grad or train)
    out = Var(buffer, requires_grad=requires)
    if not requires:
        return out

    def backward(g, needs):
        if grad_enabled():
            raise RuntimeError('the fused dense block is first-order only: build forwards that will be differentiated '
                               'twice under tape.higher_order()')
        stream = F._stream()
        gbuf = torch.empty_like(g.data)          # private copy: the incoming gradient may be shared
        F._call('srgan_ew_unary', F.U_COPY, g.data.data_ptr(), gbuf.data_ptr(), gbuf.numel(), 0.0, 0.0, stream)
        for index in range(len(layers) - 1, -1, -1):
            layer = layers[index]
            t1, b1, t2 = saved[index]
            cin = c0 + index * growth
            width = layer.conv1.out_channels
            inv1, mean1 = layer.norm1._inverse_std()
            inv2, mean2 = layer.norm2._inverse_std()
            g_new = _ptr(gbuf, cin * hw)                                  # [N, growth, H, W] view, batch stride buffer_bs
            desc2 = _desc(n, width, h, w, growth, 3, 3, 1, 1, 0, buffer_bs)
            if train:
                F._call('srgan_conv2d_bwd_weight', desc2, t2.data_ptr(), g_new, layer.conv2.weight.grad.data_ptr(), 1, 0,
                        stream)
            g_t2 = torch.empty_like(t2)
            F._call('srgan_conv2d_bwd_data', desc2, g_new, layer.conv2.weight.data_ptr(), None, g_t2.data_ptr(), 0, 0,
                    stream)
            g_b1 = torch.empty_like(b1)
            F._call('srgan_bn_act_bwd', g_t2.data_ptr(), b1.data_ptr(), mean2.data.data_ptr(), inv2.data.data_ptr(),
                    layer.norm2.weight.data_ptr(), layer.norm2.bias.data_ptr(), 1, g_b1.data_ptr(),
                    layer.norm2.weight.grad.data_ptr() if train else None,
                    layer.norm2.bias.grad.data_ptr() if train else None, n, width, hw, 0, 0, 0, stream)
            desc1 = _desc(n, cin, h, w, width, 1, 1, 1, 0)
            if train:
                F._call('srgan_conv2d_bwd_weight', desc1, t1.data_ptr(), g_b1.data_ptr(),
                        layer.conv1.weight.grad.data_ptr(), 1, 0, stream)
            g_t1 = torch.empty_like(t1)
            F._call('srgan_conv2d_bwd_data', desc1, g_b1.data_ptr(), layer.conv1.weight.data_ptr(), None, g_t1.data_ptr(),
                    0, 0, stream)
            # batch-norm 1 backward: parameter gradients, and the gradient w.r.t. the layer's (view) input
            # accumulated into the leading channels of the gradient buffer, in one pass
            F._call('srgan_bn_act_bwd', g_t1.data_ptr(), buffer.data_ptr(), mean1.data.data_ptr(), inv1.data.data_ptr(),
                    layer.norm1.weight.data_ptr(), layer.norm1.bias.data_ptr(), 1, gbuf.data_ptr(),
                    layer.norm1.weight.grad.data_ptr() if train else None,
                    layer.norm1.bias.grad.data_ptr() if train else None, n, cin, hw, buffer_bs, buffer_bs, 1, stream)
            saved[index] = None
        gx = None
        if needs[0]:
            gx_data = torch.empty((n, c0, h, w), dtype=torch.float32, device=device)
            F._call('srgan_copy_channels', gbuf.data_ptr(), total, 0, gx_data.data_ptr(), c0, 0, c0, n, hw, 0, stream)
            gx = Var(gx_data)
        return (gx,) + (None,) * len(parameter_vars)

    out.node = Node((x,) + tuple(parameter_vars), backward, 'dense_block')
    return out
