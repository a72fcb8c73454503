import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import srgan_amd
from srgan_amd import functional as F, nn
from srgan_amd.tape import backward, no_grad
from srgan_amd.coefficient.models import MLP, Generator
from helpers import load_golden, golden_state, make_settings
from oracle import models as OM, functional as OF
from test_steps_gpu import make_experiment, finish_setup

g = load_golden('g3b_coefficient_srgan_gp_active')
B = int(g['batch_size'])
# oracle
D = OM.CoefficientMLP(10); G = OM.CoefficientGenerator(10)
D.load_state_dict(golden_state(g, 'init/D')); G.load_state_dict(golden_state(g, 'init/G'))
x, y, u = (torch.from_numpy(g[f's0/{k}']) for k in ('x', 'y', 'u'))
zd, alpha = torch.from_numpy(g['s0/z_d']), torch.from_numpy(g['s0/alpha'])
fake = G(zd).detach()
def ograd(loss):
    D.zero_grad(); loss.backward(); return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in D.named_parameters()}
p = D(x); fx = D.features; lab = OF.labeled_loss(p, y)
o_lab = ograd(lab)
D(x); fx = D.features; D(u); fu = D.features
o_unl = ograd(OF.feature_distance_loss(fu, fx, OF.abs_mean))
D(u); fu = D.features; D(fake); ff = D.features
o_fake = ograd(OF.feature_distance_loss(fu, ff, OF.abs_plus_one_sqrt_mean_neg))
interp = (alpha * u + (1 - alpha) * fake).requires_grad_()
D(interp); f = D.features.norm(dim=1)
gr = torch.autograd.grad(f, interp, torch.ones_like(f), create_graph=True)[0]
gn = gr.view(B, -1).norm(dim=1)
gp = (torch.relu(gn - 1) ** 2).mean() * 10
o_gp = ograd(gp)

exp = make_experiment(lambda: (Generator(10), MLP(10), MLP(10)), dict(batch_size=B))
exp.D.load_state_dict(golden_state(g, 'init/D')); exp.G.load_state_dict(golden_state(g, 'init/G'))
finish_setup(exp)
xv, yv, uv = (F.leaf(t.cuda()) for t in (x, y, u))
fakev = F.leaf(fake.cuda())
arena = exp.D._srgan_arena
def pgrad(loss):
    arena.zero_grad(); backward(loss); return {n: p.grad.detach().cpu().clone() for n, p in exp.D.named_parameters()}
def cmp(tag, a, b):
    for n in a:
        e = (a[n] - b[n]).abs().max().item(); m = b[n].abs().max().item()
        print(f'{tag:8s} {n:16s} err {e:.3e} max {m:.3e} rel {e/max(m,1e-30):.2e}')
pr = exp.D(xv); lv = exp.labeled_loss_function(pr, yv, order=2)
print('labeled', lv.item(), lab.item()); cmp('labeled', pgrad(lv), o_lab)
exp.D(xv); fxv = exp.D.features; exp.D(uv); fuv = exp.D.features
l2 = exp.feature_distance_loss(fuv, fxv); print('unl', l2.item()); cmp('unl', pgrad(l2), o_unl)
exp.D(uv); fuv = exp.D.features; exp.D(fakev); ffv = exp.D.features
l3 = exp.feature_distance_loss(fuv, ffv, distance_function=exp.settings.contrasting_distance_function)
print('fake', l3.item()); cmp('fake', pgrad(l3), o_fake)
exp.injected_draws = {'alpha': alpha}
exp.settings.gradient_penalty_multiplier = 10.0
l4 = exp.gradient_penalty_calculation(fakev, uv); print('gp', l4.item(), gp.item())
print('gn err', (exp.gradient_norm.cpu() - gn.detach()).abs().max().item())
cmp('gp', pgrad(l4), o_gp)
