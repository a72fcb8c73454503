"""Which host call sites issue large torch copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) during one training iteration?
Wraps Tensor.copy_ / clone / contiguous / to and prints the call stack of every call that moves >= 1M elements."""
import argparse
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline'] + sys.argv[1:]
args = bench.parse()
bench.ensure_library()
torch.cuda.set_device(0)
experiment = bench.build_experiment(args, None)
labeled = experiment.infinite_iter(experiment.train_dataset_loader)
unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
bench.one_step(experiment, labeled, unlabeled, 0)
experiment.join_dnn_stream()
torch.cuda.synchronize()
seen = {}


def wrap(name):
    original = getattr(torch.Tensor, name)

    def wrapper(self, *a, **k):
        if self.numel() >= (1 << 20) and self.is_cuda:
            stack = ''.join(traceback.format_stack(limit=7)[:-1])
            key = (name, stack)
            seen[key] = seen.get(key, 0) + 1
        return original(self, *a, **k)
    setattr(torch.Tensor, name, wrapper)


for name in ('copy_', 'clone', 'contiguous', 'to', 'zero_', 'fill_'):
    wrap(name)
bench.one_step(experiment, labeled, unlabeled, 1)
experiment.join_dnn_stream()
torch.cuda.synchronize()
for (name, stack), count in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f'==== {name} x {count}\n{stack}')
