"""Runs a batch of experiments: the reference's ``run.py`` driver (application / method selection, per-app
hyper-parameters run.py:28-70, trial naming :85-105, the settings grid loop :108-112) on the MI355X step.

    python -m torch.distributed.run --nproc-per-node 8 -m srgan_amd.run      (or plain ``python -m srgan_amd.run``)

Application / method default to the reference's (age, srgan); override with SRGAN_APPLICATION / SRGAN_METHOD /
SRGAN_STEPS for smoke runs (additive: the reference has no argv/env handling)."""
import os

from .age.sgan import AgeSganExperiment
from .age.srgan import AgeExperiment
from .coefficient.dggan import CoefficientDgganExperiment
from .crowd.dggan import CrowdDgganExperiment
from .coefficient.sgan import CoefficientSganExperiment
from .coefficient.srgan import CoefficientExperiment
from .crowd.dnn import CrowdDnnExperiment
from .crowd.srgan import CrowdExperiment
from .driving.srgan import DrivingExperiment
from .settings import Settings, convert_to_settings_list, ApplicationName, MethodName
from .utility import seed_all, clean_scientific_notation, abs_plus_one_sqrt_mean_neg, abs_mean


def build_settings(application_name, method_name):
    settings_ = Settings()
    if application_name == ApplicationName.age:
        experiment_class = {MethodName.srgan: AgeExperiment, MethodName.sgan: AgeSganExperiment}[method_name]
        settings_.matching_loss_multiplier = [1e2]
        settings_.contrasting_loss_multiplier = [1e1]
        settings_.batch_size = 600
        settings_.unlabeled_dataset_size = 50000
        settings_.labeled_dataset_size = [5000]
        settings_.gradient_penalty_multiplier = 1e2
    elif application_name == ApplicationName.driving:
        experiment_class = DrivingExperiment
        settings_.matching_loss_multiplier = [1e2]
        settings_.contrasting_loss_multiplier = [1e1]
        settings_.batch_size = 600
        settings_.unlabeled_dataset_size = None
        settings_.labeled_dataset_size = [100]
        settings_.validation_dataset_size = 9000
        settings_.gradient_penalty_multiplier = 1e2
    elif application_name == ApplicationName.coefficient:
        experiment_class = {MethodName.srgan: CoefficientExperiment,
                            MethodName.sgan: CoefficientSganExperiment,
                            MethodName.dggan: CoefficientDgganExperiment}[method_name]
        settings_.matching_loss_multiplier = [1e-1, 1e0, 1e1]
        settings_.contrasting_loss_multiplier = [1e-1, 1e0, 1e1]
        settings_.batch_size = 5000
        settings_.unlabeled_dataset_size = 50000
        settings_.labeled_dataset_size = [500]
        settings_.gradient_penalty_multiplier = 1e1
    elif application_name == ApplicationName.crowd:
        experiment_class = {MethodName.srgan: CrowdExperiment, MethodName.dnn: CrowdDnnExperiment,
                            MethodName.dggan: CrowdDgganExperiment}[method_name]
        settings_.matching_loss_multiplier = [1e3]
        settings_.contrasting_loss_multiplier = [1e2]
        settings_.batch_size = 15
        settings_.labeled_loss_order = 2
        settings_.unlabeled_dataset_size = None
        settings_.labeled_dataset_size = 50
        settings_.gradient_penalty_multiplier = 1e2
        settings_.map_directory_name = ['density3e-1']
        settings_.map_multiplier = 1e-3
    else:
        raise ValueError(f'{application_name} is not an available application.')
    settings_.summary_step_period = 5000
    settings_.labeled_dataset_seed = 0
    settings_.steps_to_run = int(os.environ.get('SRGAN_STEPS', 100000))
    settings_.learning_rate = [1e-4]
    settings_.contrasting_distance_function = abs_plus_one_sqrt_mean_neg
    settings_.matching_distance_function = abs_mean
    settings_.continue_existing_experiments = False
    settings_.save_step_period = 20000
    settings_.local_setup()
    return experiment_class, settings_


def trial_name_for(settings_, application_name, method_name):
    """The reference's trial-name string (run.py:85-105)."""
    name = 'base'
    name += f' {settings_.matching_distance_function.__name__} {settings_.contrasting_distance_function.__name__}'
    name += f' {method_name.value}' if method_name != MethodName.srgan else ''
    name += f' {application_name.value}'
    if application_name == ApplicationName.crowd:
        name += f' {settings_.map_directory_name} {getattr(settings_.crowd_dataset, "value", settings_.crowd_dataset)}'
    if method_name != MethodName.dnn:
        name += f' le{settings_.labeled_dataset_size} ue{settings_.unlabeled_dataset_size}'
    name += f' ul{settings_.matching_loss_multiplier:e} fl{settings_.contrasting_loss_multiplier:e}'
    name += f' gp{settings_.gradient_penalty_multiplier:e} lr{settings_.learning_rate:e}'
    name += f' mm{settings_.map_multiplier:e}' if application_name == ApplicationName.crowd else ''
    name += f' ls{settings_.labeled_dataset_seed} bs{settings_.batch_size}'
    name += ' l' if settings_.load_model_path and not settings_.continue_existing_experiments else ''
    return clean_scientific_notation(name)


def main():
    application_name = ApplicationName(os.environ.get('SRGAN_APPLICATION', 'age'))
    method_name = MethodName(os.environ.get('SRGAN_METHOD', 'srgan'))
    experiment_class, settings_ = build_settings(application_name, method_name)
    dp = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        from .parallel import DataParallel
        dp = DataParallel.from_environment()
    settings_list = convert_to_settings_list(settings_, shuffle=dp is None)   # ranks must agree on the order
    seed_all(0)
    previous_trial_directory = None
    for settings_ in settings_list:
        settings_.trial_name = trial_name_for(settings_, application_name, method_name)
        if previous_trial_directory and settings_.continue_from_previous_trial:
            settings_.load_model_path = previous_trial_directory
        experiment = experiment_class(settings_)
        experiment.dp = dp
        experiment.train()
        previous_trial_directory = experiment.trial_directory
        if experiment.signal_quit:
            break


if __name__ == '__main__':
    main()
