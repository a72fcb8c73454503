"""Which library entry points does one iteration call, how often, and from where?  (host-side histogram of F._call / fused F._call)
    python scratch/count_calls.py [entry point substring]"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from srgan_amd import functional as F
want = sys.argv[1] if len(sys.argv) > 1 else None
sys.argv = ['bench.py', '--no-cpu-baseline', '--single-stream']
args = bench.parse()
exp = bench.build_experiment(args, None)
labeled = exp.infinite_iter(exp.train_dataset_loader); unlabeled = exp.infinite_iter(exp.unlabeled_dataset_loader)
bench.one_step(exp, labeled, unlabeled, 0)
names, sites = collections.Counter(), collections.Counter()
real = F._call
def counting(name, *a):
    names[name] += 1
    if want and want in name:
        stack = traceback.extract_stack(limit=7)[:-1]
        sites[(name, ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(stack[-5:])))] += 1
    return real(name, *a)
F._call = counting
bench.one_step(exp, labeled, unlabeled, 1)
torch.cuda.synchronize()
for name, n in names.most_common(60):
    print(f'{n:5d} {name}')
for (name, where), n in sites.most_common(30):
    print(f'{n:5d} {name:28s} {where}')
