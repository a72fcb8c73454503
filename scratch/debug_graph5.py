import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import srgan_amd
from srgan_amd import functional as F, nn
from srgan_amd.tape import backward, no_grad
from srgan_amd.crowd.models import DCGenerator
mode = sys.argv[1]
torch.manual_seed(0)
G = DCGenerator(image_size=64)
nn.flatten_parameters(G, torch.device('cuda'))
z = torch.randn(2, 256, device='cuda')
def work():
    if mode == 'forward':
        with no_grad():
            return [G(F.constant(z))]
    if mode == 'stages':
        with no_grad():
            out = G.fc(F.view(F.constant(z), (2, 256, 1, 1)))
            outs = [out]
            for stage in (G.layer1, G.layer2, G.layer3):
                pre = stage(out); outs.append(pre)
                out = F.leaky_relu(pre, 0.05); outs.append(out)
            outs.append(G.layer4(out))
            return outs
    G._srgan_arena.zero_grad()
    fake = G(F.constant(z))
    loss = F.sum_all(F.square(fake))
    backward(loss)
    return [fake, loss, Var(G._srgan_arena.grad)] if False else [fake, loss]
for _ in range(2):
    reference = [t.data.clone() for t in work()]
    ref_grad = G._srgan_arena.grad.clone()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    outputs = work()
for trial in range(4):
    graph.replay()
    torch.cuda.synchronize()
    print(mode, 'replay', trial, ['%.1e' % float((o.data - r).abs().max() / r.abs().max()) for o, r in zip(outputs, reference)],
          'grad %.1e' % float((G._srgan_arena.grad - ref_grad).abs().max() / ref_grad.abs().max()) if mode == 'full' else '')
