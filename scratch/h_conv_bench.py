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
reps = 20
print(f'{what}: environment', {k: v for k, v in os.environ.items() if k.startswith('SRGAN_H_')})
with no_grad():
    for n, c, k, h, w in SHAPES:
        layer = torch.nn.Conv2d(c, k, 3, padding=1).cuda()
        x = B.pack(F.leaf(torch.randn(n, c, h, w, device='cuda')), 1)
        s = B.pack(F.leaf(torch.randn(n, k, h, w, device='cuda')), 1)
        shadow = B.shadow_of(layer, 'conv3x3', 1)
        into = torch.zeros_like(layer.weight.data)
        run = (lambda: B.conv3x3(x, layer, slope=0.0)) if what == 'fwd' else (lambda: B._weight_gradient(shadow, layer, x, s, into))
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
