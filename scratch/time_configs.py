"""Times dnn_training_step + gan_training_step at the other BASELINE.json configurations' shapes (they are parity
cases, not bench lines): age VGG-16 D @64x64 batch 128, age DCGAN @128x128 batch 128, driving DCGAN @64x192 batch 128,
coefficient MLP batch 256."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srgan_amd  # noqa: F401
from srgan_amd.settings import Settings
from srgan_amd.utility import SummaryWriter, seed_all
import srgan_amd.age.srgan as age
from srgan_amd.driving.srgan import DrivingExperiment
from srgan_amd.coefficient.srgan import CoefficientExperiment


def timed(name, experiment, x, y, u, steps=5):
    experiment.dnn_summary_writer, experiment.gan_summary_writer = SummaryWriter(), SummaryWriter()
    experiment.gpu_mode()
    experiment.prepare_optimizers()
    experiment.train_mode()
    for step in range(2):
        experiment.dnn_training_step(x, y, step)
        experiment.gan_training_step(x, y, u, step)
    torch.cuda.synchronize()
    start = time.time()
    for step in range(2, 2 + steps):
        experiment.dnn_training_step(x, y, step)
        experiment.gan_training_step(x, y, u, step)
    torch.cuda.synchronize()
    ms = (time.time() - start) / steps * 1e3
    losses = {k: round(float(v.item()), 4) for k, v in experiment.last_losses.items() if v is not None}
    print(f'{name}: {ms:8.2f} ms/step  {x.shape[0] / ms * 1e3:9.1f} examples/s  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  {losses}',
          flush=True)


def images(batch, height, width):
    generator = torch.Generator().manual_seed(1)
    x = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    u = torch.rand(batch, 3, height, width, generator=generator) * 2 - 1
    y = torch.rand(batch, generator=generator) * 85 + 10
    return x.cuda(), y.cuda(), u.cuda()


def build(cls, batch, **attributes):
    settings = Settings()
    settings.batch_size = batch
    experiment = cls(settings)
    for key, value in attributes.items():
        setattr(experiment, key, value)
    seed_all(0)
    experiment.model_setup()
    return experiment


age.model_architecture = 'vgg'
timed('age VGG-16 64x64 B128', build(age.AgeExperiment, 128, image_size=64), *images(128, 64, 64))
age.model_architecture = 'dcgan'
timed('age DCGAN 128x128 B128', build(age.AgeExperiment, 128), *images(128, 128, 128))
timed('driving DCGAN 64x192 B128', build(DrivingExperiment, 128, image_size=(64, 192)), *images(128, 64, 192))
generator = torch.Generator().manual_seed(1)
timed('coefficient MLP B256', build(CoefficientExperiment, 256), torch.randn(256, 50, generator=generator).cuda(),
      torch.randn(256, generator=generator).cuda(), torch.randn(256, 50, generator=generator).cuda(), steps=50)
