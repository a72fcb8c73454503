"""Driving networks: the reference's driving/models.py is a verbatim copy of age/models.py."""
from ..age.models import Generator, Discriminator, convolution, transpose_convolution  # noqa: F401
