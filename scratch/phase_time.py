"""GPU time of the phases of one training iteration at the benchmark configuration (events around the sub-steps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse()
args.no_cpu_baseline = True
experiment = bench.build_experiment(args, None)
labeled = experiment.infinite_iter(experiment.train_dataset_loader)
unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
marks = []


def mark(name):
    event = torch.cuda.Event(enable_timing=True)
    event.record()
    marks.append((name, event))


def wrap(owner, attribute, before, after):
    original = getattr(owner, attribute)

    def wrapped(*a, **k):
        mark(before)
        result = original(*a, **k)
        mark(after)
        return result
    setattr(owner, attribute, wrapped)


wrap(experiment, 'discriminator_losses_shared_forwards', 'D losses: stacked forward + backward', 'after D losses')
wrap(experiment, 'gradient_penalty_calculation', 'gradient penalty: forward + recorded backward (graph only)', 'after GP graph')
wrap(experiment.d_optimizer, 'step', 'D Adam', 'after D Adam')
wrap(experiment, 'generator_loss_calculation', 'G loss: D(fake) forward', 'after G loss forward')
wrap(experiment.g_optimizer, 'step', 'G Adam', 'after G Adam')
for i in range(2):
    bench.one_step(experiment, labeled, unlabeled, i)
torch.cuda.synchronize()
totals = {}
steps = 3
for i in range(steps):
    x, heads, knn = next(labeled)
    u = next(unlabeled)[0]
    marks.clear()
    mark('DNN step')
    experiment.dnn_training_step(x, (heads, knn), 10 + i)
    mark('gan step start (G(z_d) etc.)')
    experiment.gan_training_step(x, (heads, knn), u, 10 + i)
    mark('end')
    torch.cuda.synchronize()
    for (name, start), (_, stop) in zip(marks[:-1], marks[1:]):
        totals[name] = totals.get(name, 0.0) + start.elapsed_time(stop) / steps
for name, ms in totals.items():
    print(f'{ms:8.2f} ms  {name}')
print(f'{sum(totals.values()):8.2f} ms  total')
