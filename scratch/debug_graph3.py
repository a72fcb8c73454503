"""Replay a captured forward+backward of the crowd discriminator with other work between the replays."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import srgan_amd
from srgan_amd import functional as F, nn
from srgan_amd.tape import backward, no_grad
from srgan_amd.crowd.models import KnnDenseNetCat, DCGenerator
which = sys.argv[1]
torch.manual_seed(0)
size = 64
D = KnnDenseNetCat(image_size=size)
G = DCGenerator(image_size=size)
for m in (D, G):
    nn.flatten_parameters(m, torch.device('cuda'))
x = torch.randn(2, 3, size, size, device='cuda')
z = torch.randn(2, 100, device='cuda')
def work():
    if which == 'D':
        D._srgan_arena.zero_grad()
        density, count, maps = D(F.constant(x))
        loss = F.add(F.sum_all(F.square(count)), F.sum_all(F.square(D.features)))
        backward(loss)
        return loss, D._srgan_arena.grad
    if which == 'G':
        G._srgan_arena.zero_grad()
        fake = G(F.constant(z))
        loss = F.sum_all(F.square(fake))
        backward(loss)
        return loss, G._srgan_arena.grad
    if which == 'GD':
        G._srgan_arena.zero_grad()
        fake = G(F.constant(z))
        with nn.frozen_parameters(D):
            D(fake)
            loss = F.sum_all(F.square(D.features))
        backward(loss)
        return loss, G._srgan_arena.grad
for _ in range(2):
    loss, grad = work()
torch.cuda.synchronize()
reference = (float(loss.item()), grad.clone())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss, grad = work()
for trial in range(4):
    if trial >= 2:
        junk = torch.randn(1 << 22, device='cuda').abs().max()       # torch kernels between replays
    graph.replay()
    torch.cuda.synchronize()
    print(which, 'replay', trial, 'loss', float(loss.item()), 'vs', reference[0], 'grad max diff', float((grad - reference[1]).abs().max()),
          'grad max', float(reference[1].abs().max()))
