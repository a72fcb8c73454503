"""Per-layer: is a captured conv_transpose2d forward + backward replayed identically?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import srgan_amd
from srgan_amd import functional as F
from srgan_amd.tape import backward
torch.manual_seed(0)
cases = [(2, 256, 1, 1, 512, 4, 1, 0), (2, 512, 4, 4, 256, 4, 2, 1), (2, 256, 8, 8, 128, 4, 2, 1), (2, 128, 16, 16, 64, 4, 2, 1),
         (2, 64, 32, 32, 3, 4, 2, 1)]
for (n, cin, h, w, cout, k, stride, pad) in cases:
    x = F.leaf(torch.randn(n, cin, h, w, device='cuda'), requires_grad=True)
    weight = F.leaf(torch.randn(cin, cout, k, k, device='cuda') * 0.05, requires_grad=True)
    bias = F.leaf(torch.randn(cout, device='cuda'), requires_grad=True)
    def work():
        for v in (x, weight, bias):
            v.grad = None
        y = F.conv_transpose2d(x, weight, bias, (stride, stride), (pad, pad))
        loss = F.sum_all(F.square(y))
        backward(loss)
        return y, x.grad, weight.grad, bias.grad
    for _ in range(2):
        reference = [t.data.clone() for t in work()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outputs = work()
    for trial in range(3):
        graph.replay()
        torch.cuda.synchronize()
        print((n, cin, h, w, cout, k, stride), 'replay', trial,
              ['%.2e' % float((o.data - r).abs().max() / r.abs().max()) for o, r in zip(outputs, reference)])
