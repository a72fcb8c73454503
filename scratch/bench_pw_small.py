"""Event-timed loop over the fused 1x1 forward (norm -> relu -> conv, C -> 128) on the small planes of 224 x 224."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srgan_amd
from srgan_amd import _lib
lib = _lib.library(); stream = _lib.stream_handle()
def time_case(n, c, h, w, k=128, reps=300, into_zeros=True):
    total = c + 32
    x = torch.randn(n, total, h, w, device='cuda')
    bn = [torch.randn(c, device='cuda') * 0.1, torch.rand(c, device='cuda') + 0.5, torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1]
    weight = torch.randn(k, c, 1, 1, device='cuda') * 0.03
    y = torch.zeros(n, k, h, w, device='cuda')
    desc = _lib.ConvDesc(n, c, h, w, k, 1, 1, 1, 1, 0, 0, h, w, total * h * w, 0)
    struct = _lib.BnRelu(*(t.data_ptr() for t in bn))
    fn = lib.srgan_conv2d_fwd_bnrelu_into_zeros if into_zeros else lib.srgan_conv2d_fwd_bnrelu
    split = lib.srgan_conv2d_fwd_bnrelu_splits(desc)
    for _ in range(20):
        fn(desc, x.data_ptr(), struct, weight.data_ptr(), None, y.data_ptr(), stream)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn(desc, x.data_ptr(), struct, weight.data_ptr(), None, y.data_ptr(), stream)
    stop.record(); torch.cuda.synchronize()
    us = start.elapsed_time(stop) / reps * 1e3
    print(f'n={n:3d} c={c:5d} {h}x{w} split={split:2d}: {us:7.1f} us  {2 * k * c * n * h * w / us / 1e6:6.1f} TF/s')
for (n, h) in ((16, 14), (48, 14), (16, 7), (16, 28)):
    for c in (256, 1024, 1792):
        time_case(n, c, h, h)
