import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srgan_amd
from srgan_amd import functional as F
shapes = [(16,128,128,128,32,3), (16,128,64,64,32,3), (16,128,32,32,32,3), (16,32,128,128,128,3), (16,32,32,32,128,3),
          (16,256,128,128,128,1), (16,1024,32,32,128,1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (n,c,h,w,k,r) in shapes:
    x = F.leaf(torch.randn(n,c,h,w).cuda()); wt = F.leaf((torch.randn(k,c,r,r)/ (c*r*r)**0.5).cuda())
    for _ in range(3): y = F.conv2d(x, wt, None, 1, r//2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): y = F.conv2d(x, wt, None, 1, r//2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    fl = 2.0*n*k*c*r*r*h*w
    print(f'conv {c}->{k} k{r} @{h}x{w} B{n}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s', flush=True)
    if os.environ.get('WGRAD'):
        gy = F.leaf(torch.randn(n,k,h,w).cuda())
        for _ in range(3): g = F.conv2d_backward_weight(x, gy, wt.shape, (1, 1), (r//2, r//2))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps): g = F.conv2d_backward_weight(x, gy, wt.shape, (1, 1), (r//2, r//2))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/reps
        print(f'   wgrad {c}->{k} k{r} @{h}x{w} B{n}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s', flush=True)
    if os.environ.get('BWD'):
        gy = F.leaf(torch.randn(n,k,h,w).cuda())
        for _ in range(3): g = F.conv2d_backward_data(gy, wt, x.shape, (1, 1), (r//2, r//2))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps): g = F.conv2d_backward_data(gy, wt, x.shape, (1, 1), (r//2, r//2))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/reps
        print(f'   bwd_data {k}->{c} k{r} @{h}x{w} B{n}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s', flush=True)
