"""Who launches the small elementwise kernels of one iteration?  (call-site histogram of a few functional ops)"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from srgan_amd import functional as F
args = bench.parse(); args.no_cpu_baseline = True
exp = bench.build_experiment(args, None)
labeled = exp.infinite_iter(exp.train_dataset_loader); unlabeled = exp.infinite_iter(exp.unlabeled_dataset_loader)
bench.one_step(exp, labeled, unlabeled, 0)
counts = collections.Counter()
def wrap(name):
    real = getattr(F, name)
    def wrapped(*a, **k):
        stack = traceback.extract_stack(limit=6)[:-1]
        counts[(name, ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(stack[-4:])))] += 1
        return real(*a, **k)
    setattr(F, name, wrapped)
for name in ('add', 'mask_mul', '_unary_raw', '_binary_raw', 'chan_reduce', 'chan_affine', 'accumulate_'):
    wrap(name)
bench.one_step(exp, labeled, unlabeled, 1)
torch.cuda.synchronize()
for (name, where), n in counts.most_common(40):
    print(f'{n:5d} {name:12s} {where}')
