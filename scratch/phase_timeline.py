"""When do the chains of one training iteration run?  Events on each stream at the phase boundaries of the four-stream
schedule (no profiler: its per-dispatch bookkeeping distorts the overlap), relative to the start of the iteration's GAN step.
    python scratch/phase_timeline.py [bench arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from srgan_amd import srgan as S

sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline'] + sys.argv[1:]
args = bench.parse()
experiment = bench.build_experiment(args, None)
labeled = experiment.infinite_iter(experiment.train_dataset_loader)
unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
marks = []


def mark(name):
    event = torch.cuda.Event(enable_timing=True)
    event.record(torch.cuda.current_stream())
    marks.append((name, event))


def wrap(owner, name, before, after):
    real = getattr(owner, name)

    def wrapped(*a, **k):
        mark(before)
        result = real(*a, **k)
        mark(after)
        return result
    setattr(owner, name, wrapped)


wrap(experiment, '_dnn_training_step', 'DNN step start', 'DNN step end (Adam enqueued)')
wrap(experiment, 'gradient_penalty_calculation', 'penalty chain: forward + recorded backward start', 'penalty chain: forward + recorded backward end')
wrap(experiment, 'discriminator_losses_shared_forwards', 'stacked pass forward start', 'stacked pass forward end')
wrap(experiment, 'generator_loss_calculation', 'generator step: D(fake) / D(u) forward start', 'generator step: D forwards end')
real_gan = experiment.gan_training_step


def gan(*a, **k):
    mark('GAN step start (main)')
    result = real_gan(*a, **k)
    mark('GAN step end (generator Adam enqueued, main)')
    return result


experiment.gan_training_step = gan
real_penalty = experiment._gradient_penalty_on_its_own_stream


def penalty(stream, *a, **k):
    result = real_penalty(stream, *a, **k)
    with torch.cuda.stream(stream):
        mark('penalty chain end (its own backward done)')
    mark('stacked chain start (main, after the penalty chain was enqueued)')
    return result


experiment._gradient_penalty_on_its_own_stream = penalty
real_update = experiment.start_update


def update(name, *a, **k):
    if name == 'D':
        mark('chains joined, gradient buffers added (main)')
    return real_update(name, *a, **k)


experiment.start_update = update
real_exchange = experiment.gradient_exchange


def exchange(module):
    if module is experiment.D:
        mark('stacked chain end (its backward done, main)')
    return real_exchange(module)


experiment.gradient_exchange = exchange

for step in range(4):
    bench.one_step(experiment, labeled, unlabeled, step)
experiment.join_dnn_stream(); torch.cuda.synchronize()
totals, count = {}, 6
for step in range(count):
    marks.clear()
    bench.one_step(experiment, labeled, unlabeled, 4 + step)
    experiment.join_dnn_stream(); torch.cuda.synchronize()
    origin = dict(marks)['GAN step start (main)']
    for name, event in marks:
        totals.setdefault(name, []).append(origin.elapsed_time(event))
print(f'{"ms after the GAN step started":>30s}   (mean of {count} iterations, one iteration at a time: the DNN stream cannot run ahead)')
for name, values in sorted(totals.items(), key=lambda kv: sum(kv[1])):
    print(f'{sum(values) / len(values):30.1f}   {name}')
