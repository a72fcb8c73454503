"""Driving (steering-angle) application (surface of reference driving/srgan.py:17-46): DCGAN D / G.  The
reference resizes frames to 128x128 (driving/data.py:56,123); ``image_size`` may be set to (64, 192) for the
66x200-derived rectangular shape of BASELINE config 5."""
import math

from ..srgan import Experiment
from ..synthetic import SyntheticLoader
from .models import Generator, Discriminator


class DrivingExperiment(Experiment):
    image_size = 128

    def dataset_setup(self):
        settings = self.settings
        angle = math.pi / 2
        self.train_dataset_loader = SyntheticLoader.images(settings.batch_size, self.image_size, (-angle, angle),
                                                           seed=settings.labeled_dataset_seed, dp=self.dp)
        self.unlabeled_dataset_loader = SyntheticLoader.images(settings.batch_size, self.image_size, (-angle, angle),
                                                               seed=100, dp=self.dp)
        self.validation_dataset_loader = SyntheticLoader.images(settings.batch_size, self.image_size, (-angle, angle),
                                                                seed=101, dp=self.dp, pool=1)

    def model_setup(self):
        self.G = Generator(image_size=self.image_size)
        self.D = Discriminator(image_size=self.image_size)
        self.DNN = Discriminator(image_size=self.image_size)

    def validation_summaries(self, step):
        """MAE / NMAE / MSE of DNN and D on the train and validation batches (reference driving/srgan.py:48-67,87-104)."""
        self.regression_validation_summaries(normalized=True)
