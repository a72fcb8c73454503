"""DNN-only training (surface of reference dnn.py:15-100): the same ``dnn_training_step`` as the SRGAN experiment
with only the DNN network, its Adam state, its summary writer and a ``{'DNN', 'dnn_optimizer', 'step'}`` checkpoint.
Application classes combine it with their SRGAN experiment, as the reference does
(``class CrowdDnnExperiment(DnnExperiment, CrowdExperiment)``, crowd/dnn.py:20)."""
import datetime
import os
from abc import ABC

from . import nn
from .optim import Adam
from .srgan import Experiment
from .utility import SummaryWriter, current_device


class DnnExperiment(Experiment, ABC):
    """A trial with only a DNN."""

    def prepare_summary_writers(self):
        self.dnn_summary_writer = SummaryWriter(self.summary_directory('DNN'))
        self.dnn_summary_writer.summary_period = self.settings.summary_step_period
        self.dnn_summary_writer.steps_to_run = self.settings.steps_to_run

    def gpu_mode(self):
        if getattr(self.DNN, '_srgan_arena', None) is None:
            nn.flatten_parameters(self.DNN, current_device())

    def eval_mode(self):
        self.DNN.eval()

    def train_mode(self):
        self.DNN.train()

    def prepare_optimizers(self):
        self.gpu_mode()
        self.dnn_optimizer = Adam(self.DNN._srgan_arena, lr=self.settings.learning_rate,
                                  weight_decay=self.settings.weight_decay)

    CHECKPOINT_PARTS = (('DNN', 'dnn_optimizer'),)       # model_<step>.pth = {'DNN', 'dnn_optimizer', 'step'}

    def training_loop(self):
        """reference dnn.py:78-100: the SRGAN loop without the unlabeled stream and the GAN step."""
        train_dataset_generator = self.infinite_iter(self.train_dataset_loader)
        step_time_start = datetime.datetime.now()
        for step in range(self.starting_step, self.settings.steps_to_run):
            self.adjust_learning_rate(step)
            labeled_examples, labels = self.unpack_labeled(next(train_dataset_generator))
            self.dnn_training_step(labeled_examples, labels, step)
            step_time_start = self.end_of_step(step, self.dnn_summary_writer, step_time_start)
