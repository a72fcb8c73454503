"""Fused data gradient + batch-norm backward (srgan_conv2d_bwd_data_bnrelu) against the two-kernel form, on the
conv1 shapes of the DenseNet blocks (gx accumulated into a wider gradient buffer).  usage: bench_epilogue.py [n,c,h,w ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srgan_amd  # noqa: F401
from srgan_amd import _lib

shapes = [(16, 160, 128, 128), (16, 320, 64, 64), (16, 512, 64, 64), (16, 512, 32, 32), (16, 1280, 32, 32), (48, 1024, 16, 16),
          (48, 320, 64, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
lib = _lib.library()
stream = _lib.stream_handle()          # (registers the split-K / partial-sum workspace for the stream)
k = 128
for (n, c, h, w) in shapes:
    total = c + 32
    hw = h * w
    buffer = torch.randn(n, total, h, w, device='cuda')
    gbuf = torch.zeros(n, total, h, w, device='cuda')
    gy = torch.randn(n, k, h, w, device='cuda')
    weight = torch.randn(k, c, device='cuda') / c ** 0.5
    mean, inv, gamma, beta = (torch.randn(c, device='cuda') * 0.2, torch.rand(c, device='cuda') + 0.5,
                              torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.2)
    g_gamma, g_beta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    bn = _lib.BnRelu(mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    wide = _lib.ConvDesc(n, c, h, w, k, 1, 1, 1, 1, 0, 0, h, w, total * hw, 0)
    dense = _lib.ConvDesc(n, c, h, w, k, 1, 1, 1, 1, 0, 0, h, w, 0, 0)
    temp = torch.empty(n, c, h, w, device='cuda')

    def fused(params=True):
        _lib.check(lib.srgan_conv2d_bwd_data_bnrelu(wide, gy.data_ptr(), weight.data_ptr(), bn, buffer.data_ptr(),
                                                    gbuf.data_ptr(), g_gamma.data_ptr() if params else None,
                                                    g_beta.data_ptr() if params else None, 1, stream), 'fused')

    def two_step(params=True):
        _lib.check(lib.srgan_conv2d_bwd_data(dense, gy.data_ptr(), weight.data_ptr(), None, temp.data_ptr(), 0, 0, stream), 'a')
        _lib.check(lib.srgan_bn_act_bwd(temp.data_ptr(), buffer.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), 1, gbuf.data_ptr(), g_gamma.data_ptr() if params else None,
                                        g_beta.data_ptr() if params else None, n, c, hw, 0, total * hw, total * hw, 1, 0,
                                        stream), 'b')

    def timed(fn, *args):
        for _ in range(3):
            fn(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3

    elements = n * c * hw
    print(f'{k}->{c} @{h}x{w} B{n}: two-step {timed(two_step):7.1f} us | fused {timed(fused):7.1f} us '
          f'({12 * elements / timed(fused) / 1e3:6.0f} GB/s on 12 B/element) | without parameter gradients: two-step '
          f'{timed(two_step, False):7.1f}  fused {timed(fused, False):7.1f} us', flush=True)
