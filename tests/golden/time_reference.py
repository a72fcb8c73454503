"""SURVEY.md 8(d) cross-check, run ONCE in the build container (needs /root/reference; test infrastructure only): one
training iteration of the UNMODIFIED reference classes (size-generalised by subclassing, as in make_goldens.py) timed next
to the CPU oracle that ``bench.py``'s ``cpu_baseline`` leg times on the GPU box -- same shape, batch 1, same weights scale,
same thread count.  Expected: identical losses, the same images/s within noise.  Output kept in profiles/.

    python tests/golden/time_reference.py [image size, default 512] [threads, default all]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_goldens as G  # noqa: E402  (installs the stubs and imports the reference)
import torch  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
torch.set_num_threads(threads)
batch, warmup, timed, scale = 1, 1, 2, 1.27

reference = G._image_experiment(G._crowd_builders(size), batch, G.CROWD_MULTIPLIERS, crowd=True)
with torch.no_grad():
    for module in reference.D.modules():
        if isinstance(module, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            module.weight.mul_(scale)

from oracle import functional as OF, models as OM  # noqa: E402
from oracle.experiment import OracleExperiment  # noqa: E402
from types import SimpleNamespace  # noqa: E402
settings = SimpleNamespace(batch_size=batch, learning_rate=1e-4, weight_decay=0, labeled_loss_multiplier=1.0,
                           matching_loss_multiplier=1e3, contrasting_loss_multiplier=1e2, srgan_loss_multiplier=1.0,
                           gradient_penalty_multiplier=1e2, mean_offset=0, labeled_loss_order=2,
                           generator_training_step_period=1, normalize_feature_norm=False,
                           contrasting_distance_function=OF.abs_plus_one_sqrt_mean_neg,
                           matching_distance_function=OF.abs_mean, map_multiplier=1e-3)
OF.seed_all(0)
g = OM.DCGANGenerator(image_size=size)
d, dnn = OM.KnnDenseNetCat(image_size=size), OM.KnnDenseNetCat(image_size=size)
for ours, theirs in ((g, reference.G), (d, reference.D), (dnn, reference.DNN)):     # identical weights
    ours.load_state_dict(theirs.state_dict())
oracle = OracleExperiment(settings, d, dnn, g,
                          labeled_loss_function=lambda p, y, order: OF.crowd_labeled_loss(p, y, order, 1e-3))

generator = torch.Generator().manual_seed(0)
x, y, u = G._crowd_inputs(generator, batch, size)


def run(name, step):
    seconds = []
    for iteration in range(warmup + timed):
        torch.manual_seed(iteration)
        import numpy as np
        np.random.seed(iteration)
        start = time.perf_counter()
        losses = step(iteration)
        seconds.append(time.perf_counter() - start)
    rate = batch * timed / sum(seconds[warmup:])
    print(f'{name}: {rate:.4f} images/s ({" / ".join(f"{t:.2f}" for t in seconds[warmup:])} s; warm-up {seconds[0]:.2f} s), '
          f'last losses {losses}', flush=True)
    return rate


def reference_step(iteration):
    reference.dnn_training_step(x, y, iteration)
    reference.gan_training_step(x, y, u, iteration)
    scalars = G.last_scalars(reference.gan_summary_writer)
    return {G.GAN_TAGS[k]: round(float(v), 4) for k, v in scalars.items() if k in G.GAN_TAGS}


def oracle_step(iteration):
    oracle.dnn_training_step(x, y)
    result = oracle.gan_training_step(x, y, u, iteration)
    return {k: round(float(v), 4) for k, v in result.items() if isinstance(v, (int, float))}


print(f'crowd {size}x{size}, batch {batch}, {threads} threads, torch {torch.__version__}, D conv weights x{scale}')
a = run('reference (unmodified classes, size-generalised by subclassing)', reference_step)
b = run('CPU oracle (what bench.py cpu_baseline times)', oracle_step)
print(f'oracle / reference = {b / a:.3f}')
