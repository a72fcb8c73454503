"""Layer containers.  They subclass the torch.nn modules the reference uses, so construction order,
default initialisation (same random stream) and ``state_dict`` keys are the reference's; only ``forward``
is replaced: it takes and returns tape ``Var``s and launches HIP kernels.  torch.nn here is a parameter
container / checkpoint format, not a compute path.
"""
import torch
from torch import nn

from . import functional as F
from .tape import Var


def parameter_var(parameter):
    """The persistent graph leaf of a parameter (rebuilt if the parameter's storage moved)."""
    var = getattr(parameter, '_srgan_var', None)
    if var is None or var.data.data_ptr() != parameter.data.data_ptr():
        var = F.leaf(parameter.data, requires_grad=parameter.requires_grad)
        if var.data.data_ptr() != parameter.data.data_ptr():
            raise RuntimeError('parameters must be contiguous fp32 device tensors (call flatten_parameters first)')
        parameter._srgan_var = var
    var.grad_buffer = parameter.grad
    return var


P = parameter_var


class Conv2d(nn.Conv2d):
    def forward(self, x):
        return F.conv2d(x, P(self.weight), P(self.bias) if self.bias is not None else None, self.stride, self.padding)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        return F.conv_transpose2d(x, P(self.weight), P(self.bias) if self.bias is not None else None, self.stride,
                                  self.padding)


class Linear(nn.Linear):
    def forward(self, x):
        return F.linear(x, P(self.weight), P(self.bias) if self.bias is not None else None)


class BatchNorm2d(nn.BatchNorm2d):
    """Always evaluated with the running statistics: the reference freezes every batch-norm layer at each
    step (srgan.py:261,276,538-542), so training mode never updates or uses batch statistics.  gamma / beta
    are still trained.  One fused kernel: y = (x - mean) * rsqrt(var + eps) * gamma + beta."""

    def _inverse_std(self):
        key = (id(self.running_var), self.running_var._version, id(self.running_mean), self.running_mean._version)
        cached = getattr(self, '_inv_std_cache', None)
        if cached is None or cached[0] != key:
            variance = self.running_var.detach().contiguous()
            if cached is not None and cached[1].data.shape == variance.shape and cached[1].data.device == variance.device:
                # New statistics (load_state_dict, load_models): refreshed IN PLACE.  The grouped launch tables of
                # fused.py and a captured HIP graph hold the raw device pointers of these two tensors.
                inv, mean = cached[1], cached[2]
                F._unary_raw(F.U_RSQRT, F._unary_raw(F.U_AFFINE, variance, 1.0, self.eps, out=inv.data), out=inv.data)
                mean.data.copy_(self.running_mean.detach())
            else:
                inv = Var(F._unary_raw(F.U_RSQRT, F._unary_raw(F.U_AFFINE, variance, 1.0, self.eps)))
                mean = Var(self.running_mean.detach().clone().contiguous())      # a copy we own: its address never moves
            cached = (key, inv, mean)
            self._inv_std_cache = cached
        return cached[1], cached[2]

    def forward(self, x, relu=False):
        inv_std, mean = self._inverse_std()
        return F.batch_norm_eval(x, mean, inv_std, P(self.weight), P(self.bias), relu=relu)


class ReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return F.relu(x)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding

    def forward(self, x):
        return F.max_pool2d(x, self.kernel_size, self.stride, self.padding)


class AvgPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride or kernel_size

    def forward(self, x):
        return F.avg_pool2d(x, self.kernel_size, self.stride)


Sequential = nn.Sequential
Module = nn.Module
ModuleList = nn.ModuleList


class ParameterArena:
    """All parameters of one network in ONE contiguous device buffer, with a parallel gradient buffer.

    MI355X-first layout: Adam is a single streaming kernel over the arena, ``zero_grad`` one memset, and the
    data-parallel gradient exchange a handful of large RCCL all-reduces instead of one per tensor.  The
    module's Parameters become views into the arena (``state_dict`` is unaffected)."""

    def __init__(self, module, device):
        parameters = [p for p in module.parameters()]
        self.sizes = [p.numel() for p in parameters]
        # keep every tensor 16-byte aligned inside the arena (float4 kernels)
        self.offsets, total = [], 0
        for size in self.sizes:
            self.offsets.append(total)
            total += (size + 3) // 4 * 4
        self.numel = total
        self.data = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.parameters = parameters
        self.version = 0          # bumped by every optimizer update (raw-pointer writes torch's version counters do not see)
        self.shadows = []         # blocked16.Shadow objects: 16-bit operand forms of these weights, refreshed behind an update
        with torch.no_grad():
            for p, offset, size in zip(parameters, self.offsets, self.sizes):
                view = self.data[offset:offset + size].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[offset:offset + size].view(p.shape)
                p._srgan_var = None
                p._srgan_arena = self
        for buffer_owner in module.modules():      # move buffers (batch-norm statistics) as well
            for name, buffer in list(buffer_owner._buffers.items()):
                if buffer is not None:
                    buffer_owner._buffers[name] = buffer.to(device)

    def zero_grad(self):
        F.fill_(self.grad, 0.0)

    def gradients_into_alternate(self):
        """Context manager: while it is active, every gradient of this network accumulates into a SECOND buffer of the
        arena's size (returned; allocated on first use, zeroed by the caller) instead of ``self.grad`` -- for a backward
        chain that runs on another stream concurrently with one that accumulates into ``self.grad`` (the accumulations are
        read-modify-write kernels, not atomics).  The caller adds the buffer to ``self.grad`` after joining the streams."""
        import contextlib
        if getattr(self, '_alternate', None) is None:
            self._alternate = torch.empty_like(self.grad)
            self._alternate_views = [self._alternate[offset:offset + size].view(p.shape)
                                     for p, offset, size in zip(self.parameters, self.offsets, self.sizes)]
            self._main_views = [self.grad[offset:offset + size].view(p.shape)
                                for p, offset, size in zip(self.parameters, self.offsets, self.sizes)]

        def point_at(views):
            for p, view in zip(self.parameters, views):
                p.grad = view
                var = getattr(p, '_srgan_var', None)
                if var is not None:
                    var.grad_buffer = view

        @contextlib.contextmanager
        def redirected():
            point_at(self._alternate_views)
            try:
                yield self._alternate
            finally:
                point_at(self._main_views)
        return redirected()

    def rebind(self):
        """Re-point ``p.grad`` at the arena (torch utilities such as zero_grad(set_to_none) may drop it)."""
        for p, offset, size in zip(self.parameters, self.offsets, self.sizes):
            p.grad = self.grad[offset:offset + size].view(p.shape)


def flatten_parameters(module, device):
    arena = ParameterArena(module, device)
    module._srgan_arena = arena
    return arena


class frozen_parameters:
    """Context manager: the module's parameters do not receive gradients (used for the discriminator during
    the generator update -- the reference computes and then discards those gradients, srgan.py:304,278)."""

    def __init__(self, module):
        self.vars = [parameter_var(p) for p in module.parameters()]

    def __enter__(self):
        self.previous = [v.requires_grad for v in self.vars]
        for v in self.vars:
            v.requires_grad = False

    def __exit__(self, *exc):
        for v, flag in zip(self.vars, self.previous):
            v.requires_grad = flag
