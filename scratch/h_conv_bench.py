"""Times the 16-bit 3x3 kernels on VGG-16's shapes (batch 128 and the stacked 384):  python scratch/h_conv_bench.py [fwd|wgrad]
Environment switches of csrc/blocked16.hip apply (SRGAN_H_DMA_RING, SRGAN_H_CONV_NI, SRGAN_H_EXPERIMENT, SRGAN_H_WGRAD_64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import srgan_amd  # noqa: E402,F401
from srgan_amd import blocked16 as B, functional as F  # noqa: E402
from srgan_amd.tape import no_grad  # noqa: E402

SHAPES = [(128, 64, 64, 64, 64), (384, 64, 64, 64, 64), (128, 128, 128, 32, 32), (384, 128, 128, 32, 32), (128, 256, 256, 16, 16),
          (384, 256, 256, 16, 16), (128, 512, 512, 8, 8), (384, 512, 512, 8, 8), (128, 512, 512, 4, 4), (384, 512, 512, 4, 4)]
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
if len(sys.argv) > 2 and sys.argv[2] == 'first':        # the layers with few channels: conv1_1 (3 -> 64) and conv1_2 (64 -> 64)
    SHAPES = [(128, 3, 64, 64, 64), (384, 3, 64, 64, 64), (128, 64, 64, 64, 64), (384, 64, 64, 64, 64), (128, 64, 3, 64, 64)]
if len(sys.argv) > 2 and sys.argv[2] == 'channels':
    SHAPES = [(128, c, 64, 64, 64) for c in (3, 8, 16, 24, 32, 64)] + [(128, 16, k, 64, 64) for k in (8, 16, 32, 64, 128)]
reps = 20
print(f'{what}: environment', {k: v for k, v in os.environ.items() if k.startswith('SRGAN_H_')})
with no_grad():
    for n, c, k, h, w in SHAPES:
        layer = torch.nn.Conv2d(c, k, 3, padding=1, bias=os.environ.get('BENCH_NO_BIAS') is None).cuda()
        x = B.pack(F.leaf(torch.randn(n, c, h, w, device='cuda')), 1)
        s = B.pack(F.leaf(torch.randn(n, k, h, w, device='cuda')), 1)
        shadow = B.shadow_of(layer, 'conv3x3', 1)
        into = torch.zeros_like(layer.weight.data)
        if what == 'fwd':
            run = lambda: B.conv3x3(x, layer, slope=0.0)
        elif what == 'bwd':          # the data gradient with the mask epilogue (epi 2): rows = c, reduced = k, ref = the layer's input
            run = lambda: B._layer(s, layer, shadow, True, 2, 0.0, x.data, False)
        else:
            run = lambda: B._weight_gradient(shadow, layer, x, s, into)
        for _ in range(3):
            run()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            run()
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / reps
        flops = 2.0 * n * h * w * c * k * 9
        print(f'N {n:4d} C {c:4d} K {k:4d} {h:3d}x{w:<3d}  {1e3 * ms:8.1f} us  {flops / ms / 1e9:8.1f} TF/s', flush=True)
