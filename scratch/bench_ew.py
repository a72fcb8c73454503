"""Bandwidth of the HBM-bound kernels on DenseNet-shaped tensors (GB/s of algorithmic bytes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srgan_amd
from srgan_amd import functional as F, _lib
lib = _lib.library()
stream = torch.cuda.current_stream().cuda_stream

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

shapes = [(16, 128, 128, 128), (16, 256, 128, 128), (16, 128, 64, 64), (16, 512, 64, 64), (16, 128, 32, 32), (16, 1024, 32, 32),
          (16, 128, 16, 16), (16, 1024, 16, 16)]
for (n, c, h, w) in shapes:
    hw = h * w
    total = c + 64
    wide = torch.randn(n, total, h, w, device='cuda')
    x = torch.randn(n, c, h, w, device='cuda'); y = torch.empty_like(x); g = torch.randn_like(x)
    mean, inv, gamma, beta = (torch.randn(c, device='cuda') for _ in range(4))
    gg, gb = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    nbytes = x.numel() * 4
    ms = timed(lambda: lib.srgan_chan_affine_act(x.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 1, y.data_ptr(), n, c, hw, stream))
    r = [f'bnrelu_fwd {2*nbytes/ms/1e6:6.0f}']
    ms = timed(lambda: lib.srgan_chan_affine_act_strided(wide.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 1, y.data_ptr(), n, c, hw, total*hw, 0, 0, 0, stream))
    r.append(f'fwd_strided {2*nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: lib.srgan_bn_act_bwd(g.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, y.data_ptr(), gg.data_ptr(), gb.data_ptr(), n, c, hw, 0, 0, 0, 0, 0, stream))
    r.append(f'bn_bwd {3*nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: lib.srgan_bn_act_bwd(g.data_ptr(), wide.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, wide.data_ptr(), gg.data_ptr(), gb.data_ptr(), n, c, hw, 0, total*hw, total*hw, 1, 0, stream))
    r.append(f'bn_bwd_acc {4*nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: lib.srgan_ew_binary(0, x.data_ptr(), g.data_ptr(), y.data_ptr(), x.numel(), 0.0, stream))
    r.append(f'add {3*nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: lib.srgan_copy_channels(x.data_ptr(), c, 0, wide.data_ptr(), total, 0, c, n, hw, 0, stream))
    r.append(f'copy_ch {2*nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: lib.srgan_fill(y.data_ptr(), y.numel(), 0.0, stream))
    r.append(f'memset {nbytes/ms/1e6:6.0f}')
    ms = timed(lambda: y.copy_(x))
    r.append(f'torch_copy {2*nbytes/ms/1e6:6.0f}')
    print(f'[{n},{c},{h},{w}] {nbytes/1e6:6.1f} MB  GB/s: ' + '  '.join(r), flush=True)
