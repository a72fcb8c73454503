"""Run settings: the attribute bag the experiments read, with the reference's names, defaults, attribute order and
list-expansion semantics (reference settings.py:12-127).  The defaults live in one table; ``Settings()`` instantiates it
in order, because ``convert_to_settings_list`` expands list-valued attributes in attribute order."""
import platform
import random
from copy import deepcopy
from enum import Enum

from .utility import abs_plus_one_sqrt_mean_neg, abs_mean

# (name, default) in the reference's declaration order; grouped as it groups them.
DEFAULTS = (
    # the run
    ('trial_name', 'base'), ('steps_to_run', 200000), ('temporary_directory', 'temporary'), ('logs_directory', 'logs'),
    ('batch_size', 1000), ('summary_step_period', 2000), ('labeled_dataset_size', 50), ('unlabeled_dataset_size', 50000),
    ('validation_dataset_size', 1000), ('learning_rate', 1e-4), ('weight_decay', 0),
    # the losses
    ('labeled_loss_multiplier', 1e0), ('matching_loss_multiplier', 1e0), ('contrasting_loss_multiplier', 1e0),
    ('srgan_loss_multiplier', 1e0), ('dggan_loss_multiplier', 1e1), ('gradient_penalty_on', True),
    ('gradient_penalty_multiplier', 1e1), ('mean_offset', 0), ('labeled_loss_order', 2),
    ('generator_training_step_period', 1), ('labeled_dataset_seed', 0), ('normalize_fake_loss', False),
    ('normalize_feature_norm', False), ('contrasting_distance_function', abs_plus_one_sqrt_mean_neg),
    ('matching_distance_function', abs_mean),
    # checkpoints, data loading, resuming
    ('load_model_path', None), ('should_save_models', True), ('skip_completed_experiment', True),
    ('number_of_data_workers', 4), ('pin_memory', True), ('continue_from_previous_trial', False),
    ('continue_existing_experiments', False), ('save_step_period', None),
    # coefficient application
    ('hidden_size', 10),
    # crowd application
    ('crowd_dataset', 'World Expo'), ('number_of_cameras', 5), ('number_of_images_per_camera', 5),
    ('test_summary_size', None), ('test_sliding_window_size', 128), ('image_patch_size', 224), ('label_patch_size', 224),
    ('map_multiplier', 1e-6), ('map_directory_name', 'i1nn_maps'),
    # SGAN models
    ('number_of_bins', 10),
)

# what ``local_setup`` shrinks on the reference author's laptop (settings.py:69-79)
LAPTOP_OVERRIDES = {'labeled_dataset_seed': 0, 'summary_step_period': 10, 'labeled_dataset_size': 10,
                    'unlabeled_dataset_size': 10, 'validation_dataset_size': 10, 'skip_completed_experiment': False,
                    'number_of_data_workers': 0}


class Settings:
    """One run's settings; any attribute may hold a list / tuple of alternatives (see ``convert_to_settings_list``)."""

    def __init__(self):
        for name, default in DEFAULTS:
            setattr(self, name, default)

    def local_setup(self):
        if 'Carbon' not in platform.node():
            return
        self.batch_size = min(10, self.batch_size)
        for name, value in LAPTOP_OVERRIDES.items():
            setattr(self, name, value)


def _first_alternative(settings):
    """(attribute name, alternatives) of the first list- or tuple-valued attribute, or None."""
    for name, value in vars(settings).items():
        if isinstance(value, (list, tuple)):
            return name, value
    return None


def convert_to_settings_list(settings, shuffle=True):
    """Cartesian expansion of every list / tuple valued attribute into separate deep copies, in attribute order,
    optionally shuffled (reference settings.py:82-111)."""
    expanded, pending = [], [settings]
    while pending:
        current = pending.pop(0)
        alternative = _first_alternative(current)
        if alternative is None:
            expanded.append(current)
            continue
        name, options = alternative
        for option in options:
            variant = deepcopy(current)
            setattr(variant, name, option)
            pending.append(variant)
    if shuffle:
        random.seed()
        random.shuffle(expanded)
    return expanded


ApplicationName = Enum('ApplicationName', {name: name for name in ('coefficient', 'age', 'crowd', 'driving')})
MethodName = Enum('MethodName', {name: name for name in ('srgan', 'dnn', 'dggan', 'sgan')})
