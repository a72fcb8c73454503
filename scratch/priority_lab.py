"""Laboratory: stream priorities for the chains of the training iteration (main chain + penalty chain are on the critical
path, the DNN step is free-running filler).   python scratch/priority_lab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(name, main_priority, penalty_priority, dnn_priority, aux_priority, steps=10, warmup=3):
    sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline']
    args = bench.parse()
    experiment = bench.build_experiment(args, None)
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
    make = lambda priority: torch.cuda.Stream(priority=priority)
    main_stream = make(main_priority) if main_priority is not None else torch.cuda.current_stream()
    if penalty_priority is not None: experiment._gp_stream = make(penalty_priority)
    if dnn_priority is not None: experiment._dnn_stream = make(dnn_priority)
    if aux_priority is not None: experiment._aux_stream = make(aux_priority)
    with torch.cuda.stream(main_stream):
        for step in range(warmup):
            bench.one_step(experiment, labeled, unlabeled, step)
        experiment.join_dnn_stream(); torch.cuda.synchronize()
        start = time.perf_counter()
        for step in range(steps):
            bench.one_step(experiment, labeled, unlabeled, warmup + step)
        experiment.join_dnn_stream(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - start
    print(f'{name:44s}: {16 * steps / elapsed:7.2f} images/s  {1e3 * elapsed / steps:7.2f} ms', flush=True)


if __name__ == '__main__':
    print('priority range', torch.cuda.Stream.priority_range())
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    cases = {'default': (None, None, None, None), 'main high': (-1, 0, 0, 0), 'main + penalty high': (-1, -1, 0, 0),
             'main + penalty + aux high (DNN filler)': (-1, -1, 0, -1), 'created main, all normal': (0, 0, 0, 0)}
    for name, priorities in cases.items():
        if which in ('all', name):
            run(name, *priorities)
