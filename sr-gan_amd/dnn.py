"""DNN-only training (surface of reference dnn.py:15-100): the same ``dnn_training_step`` as the SRGAN experiment
with only the DNN network, its Adam state, its summary writer and a ``{'DNN', 'dnn_optimizer', 'step'}`` checkpoint.
Application classes combine it with their SRGAN experiment, as the reference does
(``class CrowdDnnExperiment(DnnExperiment, CrowdExperiment)``, crowd/dnn.py:20)."""
import datetime
import os
import re
from abc import ABC

import torch

from . import nn
from .optim import Adam
from .srgan import Experiment, as_var
from .tape import no_grad
from .utility import SummaryWriter, current_device


class DnnExperiment(Experiment, ABC):
    """A trial with only a DNN."""

    def prepare_summary_writers(self):
        self.dnn_summary_writer = SummaryWriter(os.path.join(self.trial_directory, 'DNN'))
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

    def save_models(self, step):
        self.join_dnn_stream()
        if self.dp is not None and self.dp.rank != 0:
            return
        model = {'DNN': self.DNN.state_dict(), 'dnn_optimizer': self.dnn_optimizer.state_dict(), 'step': step}
        torch.save(model, os.path.join(self.trial_directory, f'model_{step}.pth'))

    def load_models(self, with_optimizers=True):
        if not self.settings.load_model_path:
            return
        latest_model = None
        for file_name in os.listdir(self.settings.load_model_path):
            match = re.search(r'model_?(\d+)?\.pth', file_name)
            if match:
                latest_model = self.compare_model_path_for_latest(latest_model, match)
        if latest_model is None:
            return
        model_path = os.path.join(self.settings.load_model_path, latest_model.group(0))
        loaded_model = torch.load(model_path, map_location='cpu')
        self.DNN.load_state_dict(loaded_model['DNN'])
        if with_optimizers:
            self.dnn_optimizer.load_state_dict(loaded_model['dnn_optimizer'])
        print('Model loaded from `{}`.'.format(model_path))
        if self.settings.continue_existing_experiments:
            self.starting_step = loaded_model['step'] + 1
            print(f'Continuing from step {self.starting_step}')

    def training_loop(self):
        """reference dnn.py:78-100: the SRGAN loop without the GAN step."""
        train_dataset_generator = self.infinite_iter(self.train_dataset_loader)
        step_time_start = datetime.datetime.now()
        for step in range(self.starting_step, self.settings.steps_to_run):
            self.adjust_learning_rate(step)
            samples = next(train_dataset_generator)
            if len(samples) == 2:
                labeled_examples, labels = samples
            else:
                labeled_examples, primary_labels, secondary_labels = samples
                labels = (primary_labels, secondary_labels)
            self.dnn_training_step(as_var(labeled_examples), as_var(labels), step)
            if self.dnn_summary_writer.is_summary_step() or step == self.settings.steps_to_run - 1:
                print('\rStep {}, {}...'.format(step, datetime.datetime.now() - step_time_start), end='')
                step_time_start = datetime.datetime.now()
                self.join_dnn_stream()
                self.eval_mode()
                with no_grad():
                    self.validation_summaries(step)
                self.train_mode()
            self.handle_user_input(step)
            if self.settings.save_step_period and step % self.settings.save_step_period == 0 and step != 0:
                self.save_models(step=step)
