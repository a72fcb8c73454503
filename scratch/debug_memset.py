"""Which zero-fill is not replayed?  Each case: (zero a tensor allocated INSIDE the capture, then add ones into it)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import srgan_amd
from srgan_amd import functional as F, _lib
lib = _lib.library()
ones = {}
def case(n, how):
    y = torch.empty(n, device='cuda')
    if how == 'memset':
        F.fill_(y, 0.0)
    elif how == 'torch':
        y.zero_()
    elif how == 'kernel':
        F.fill_(y, 1.0); F._unary_raw(F.U_AFFINE, y, 0.0, 0.0, out=y)
    F._binary_raw(F.B_ADD, y, ones[n], out=y)
    return y
sizes = (1, 100, 4096, 100000, 1 << 22)
for n in sizes:
    ones[n] = torch.ones(n, device='cuda')
for how in ('memset', 'torch', 'kernel'):
    for n in sizes: case(n, how)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
results = {}
with torch.cuda.graph(graph):
    for how in ('memset', 'torch', 'kernel'):
        for n in sizes:
            results[(how, n)] = case(n, how)
for trial in range(3):
    graph.replay()
    torch.cuda.synchronize()
    print('replay', trial, {key: (float(v.min()), float(v.max())) for key, v in results.items() if float(v.min()) != 1.0 or float(v.max()) != 1.0})
