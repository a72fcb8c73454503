"""40 training iterations at the benchmark configuration: losses stay finite, device memory does not grow."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse()
exp = bench.build_experiment(args, None)
labeled = exp.infinite_iter(exp.train_dataset_loader); unlabeled = exp.infinite_iter(exp.unlabeled_dataset_loader)
marks = []
for i in range(40):
    bench.one_step(exp, labeled, unlabeled, i)
    if i % 10 == 9:
        torch.cuda.synchronize()
        losses = {k: float(v.item()) for k, v in exp.last_losses.items() if v is not None}
        marks.append((i + 1, torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30,
                      torch.cuda.memory_reserved() / 2**30, losses))
for step, allocated, peak, reserved, losses in marks:
    print(f'step {step}: allocated {allocated:.2f} GiB, peak {peak:.2f} GiB, reserved by the caching allocator {reserved:.2f} GiB, '
          + ', '.join(f'{k} {v:.4g}' for k, v in losses.items()))
