"""Age application (surface of reference age/srgan.py:17-51).  The IMDB-WIKI loaders need downloads and are out
of scope; ``dataset_setup`` yields synthetic faces of the same batch contract (image f32[3,S,S] in [-1,1],
age in [10, 95])."""
from ..srgan import Experiment
from ..synthetic import SyntheticLoader
from .models import Generator, Discriminator
from .vgg import vgg16

model_architecture = 'dcgan'  # dcgan or vgg (reference age/srgan.py:14)


class AgeExperiment(Experiment):
    """The age estimation application."""
    image_size = None      # None = the architecture's native size (128 dcgan / 224 vgg)

    def _size(self):
        if self.image_size is not None:
            return self.image_size
        return 224 if model_architecture == 'vgg' else 128

    def dataset_setup(self):
        settings = self.settings
        self.train_dataset_loader = SyntheticLoader.images(settings.batch_size, self._size(), (10.0, 95.0),
                                                           seed=settings.labeled_dataset_seed, dp=self.dp)
        self.unlabeled_dataset_loader = SyntheticLoader.images(settings.batch_size, self._size(), (10.0, 95.0),
                                                               seed=100, dp=self.dp)
        self.validation_dataset_loader = SyntheticLoader.images(settings.batch_size, self._size(), (10.0, 95.0),
                                                                seed=101, dp=self.dp, pool=1)

    def model_setup(self):
        """reference age/srgan.py:42-51 (``pretrained=True`` VGG weights need a download: load a checkpoint)."""
        size = self._size()
        if model_architecture == 'vgg':
            self.G = Generator(image_size=size)
            self.D = vgg16(num_classes=1, image_size=size)
            self.DNN = vgg16(num_classes=1, image_size=size)
        else:
            self.G = Generator(image_size=size)
            self.D = Discriminator(image_size=size)
            self.DNN = Discriminator(image_size=size)

    def validation_summaries(self, step):
        """MAE / MSE of DNN and D on the train and validation batches (reference age/srgan.py:52-71,92-107)."""
        self.regression_validation_summaries()
