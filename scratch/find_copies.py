"""Which host call sites run torch (aten) operations on large tensors during one training iteration?  The library's own
launches do not go through aten; what shows here is torch arithmetic / copies left on the hot path (the three image-sized
__amd_rocclr_copyBuffer launches per iteration in the kernel trace).  A TorchDispatchMode logs every aten op whose largest
tensor argument has >= 1M elements, with the Python stack."""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

THRESHOLD = 1 << 20
if '--threshold' in sys.argv:
    at = sys.argv.index('--threshold')
    THRESHOLD = int(sys.argv[at + 1])
    del sys.argv[at:at + 2]
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


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        flat, _ = tree_flatten((args, kwargs or {}))
        largest = max([t.numel() for t in flat if isinstance(t, torch.Tensor)] or [0])
        name = str(func)
        if largest >= THRESHOLD and not any(k in name for k in ('aten.view', 'aten.detach', 'aten.slice', 'aten.select', 'aten.alias',
                                                                   'aten._unsafe_view', 'aten.as_strided', 'aten.expand', 'aten.t.')):
            stack = ''.join(traceback.format_stack(limit=9)[:-1]) if THRESHOLD else ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(traceback.extract_stack(limit=7)[:-1]))
            key = (name, largest, stack)
            seen[key] = seen.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


with Log():
    bench.one_step(experiment, labeled, unlabeled, 1)
    experiment.join_dnn_stream()
torch.cuda.synchronize()
for (name, largest, stack), count in sorted(seen.items(), key=lambda kv: (-kv[1], -kv[0][1])):
    print(f'==== {name} x {count}, largest tensor {largest} elements\n{stack}')
