"""Runs a batch of experiments: the reference's ``run.py`` driver (application / method selection, per-app
hyper-parameters run.py:28-70, trial naming :85-105, the settings grid loop :108-112) on the MI355X step.

    python -m torch.distributed.run --nproc-per-node 8 -m srgan_amd.run      (or plain ``python -m srgan_amd.run``)

Application / method default to the reference's (age, srgan); override with SRGAN_APPLICATION / SRGAN_METHOD /
SRGAN_STEPS / SRGAN_BATCH_SIZE for smoke runs (additive: the reference has no argv/env handling).  Under
torch.distributed.run ``batch_size`` is the GLOBAL batch and must be divisible by the number of ranks (the crowd
default of 15 is not divisible by 8: set SRGAN_BATCH_SIZE=16); ``Experiment.train`` checks it."""
import os

from .age.sgan import AgeSganExperiment
from .age.srgan import AgeExperiment
from .coefficient.dggan import CoefficientDgganExperiment
from .crowd.dggan import CrowdDgganExperiment
from .coefficient.sgan import CoefficientSganExperiment
from .coefficient.srgan import CoefficientExperiment
from .crowd.dnn import CrowdDnnExperiment
from .crowd.sgan import CrowdSganExperiment
from .crowd.srgan import CrowdExperiment
from .driving.srgan import DrivingExperiment
from .settings import Settings, convert_to_settings_list, ApplicationName, MethodName
from .utility import seed_all, clean_scientific_notation, abs_plus_one_sqrt_mean_neg, abs_mean


# Which experiment class runs an (application, method) pair, and the hyper-parameters each application overrides
# (reference run.py:28-70).  List values are grid axes (see settings.convert_to_settings_list).
EXPERIMENTS = {
    ApplicationName.age: {MethodName.srgan: AgeExperiment, MethodName.sgan: AgeSganExperiment},
    ApplicationName.driving: {method: DrivingExperiment for method in MethodName},
    ApplicationName.coefficient: {MethodName.srgan: CoefficientExperiment, MethodName.sgan: CoefficientSganExperiment,
                                  MethodName.dggan: CoefficientDgganExperiment},
    ApplicationName.crowd: {MethodName.srgan: CrowdExperiment, MethodName.sgan: CrowdSganExperiment,
                            MethodName.dnn: CrowdDnnExperiment, MethodName.dggan: CrowdDgganExperiment},
}
APPLICATION_SETTINGS = {
    ApplicationName.age: dict(matching_loss_multiplier=[1e2], contrasting_loss_multiplier=[1e1], batch_size=600,
                              unlabeled_dataset_size=50000, labeled_dataset_size=[5000],
                              gradient_penalty_multiplier=1e2),
    ApplicationName.driving: dict(matching_loss_multiplier=[1e2], contrasting_loss_multiplier=[1e1], batch_size=600,
                                  unlabeled_dataset_size=None, labeled_dataset_size=[100], validation_dataset_size=9000,
                                  gradient_penalty_multiplier=1e2),
    ApplicationName.coefficient: dict(matching_loss_multiplier=[1e-1, 1e0, 1e1],
                                      contrasting_loss_multiplier=[1e-1, 1e0, 1e1], batch_size=5000,
                                      unlabeled_dataset_size=50000, labeled_dataset_size=[500],
                                      gradient_penalty_multiplier=1e1),
    ApplicationName.crowd: dict(matching_loss_multiplier=[1e3], contrasting_loss_multiplier=[1e2], batch_size=15,
                                labeled_loss_order=2, unlabeled_dataset_size=None, labeled_dataset_size=50,
                                gradient_penalty_multiplier=1e2, map_directory_name=['density3e-1'],
                                map_multiplier=1e-3),
}
COMMON_SETTINGS = dict(summary_step_period=5000, labeled_dataset_seed=0, learning_rate=[1e-4],
                       contrasting_distance_function=abs_plus_one_sqrt_mean_neg, matching_distance_function=abs_mean,
                       continue_existing_experiments=False, save_step_period=20000)


def build_settings(application_name, method_name):
    """(experiment class, Settings) of one application / method pair; application values first, then the common ones,
    as in the reference (a later assignment wins)."""
    if application_name not in APPLICATION_SETTINGS:
        raise ValueError(f'{application_name} is not an available application.')
    experiment_class = EXPERIMENTS[application_name][method_name]
    settings_ = Settings()
    overrides = dict(APPLICATION_SETTINGS[application_name], **COMMON_SETTINGS)
    overrides['steps_to_run'] = int(os.environ.get('SRGAN_STEPS', 100000))
    if os.environ.get('SRGAN_BATCH_SIZE'):
        overrides['batch_size'] = int(os.environ['SRGAN_BATCH_SIZE'])
    for name, value in overrides.items():
        setattr(settings_, name, value)
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
    force_dp = bool(int(os.environ.get('SRGAN_FORCE_DP', '0')))       # the collective path on a world of one
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or force_dp:
        if force_dp:
            for key, value in (('RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533')):
                os.environ.setdefault(key, value)
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
