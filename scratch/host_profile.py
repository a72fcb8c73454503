"""Where does the HOST spend its time while it enqueues one training step?  (cProfile over three steps.)

    python scratch/host_profile.py --image-size 224
"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse()
args.no_cpu_baseline = True
exp = bench.build_experiment(args, None)
labeled = exp.infinite_iter(exp.train_dataset_loader); unlabeled = exp.infinite_iter(exp.unlabeled_dataset_loader)
for i in range(2): bench.one_step(exp, labeled, unlabeled, i)
torch.cuda.synchronize()
profile = cProfile.Profile()
profile.enable()
for i in range(3):
    bench.one_step(exp, labeled, unlabeled, 10 + i)
profile.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    out = io.StringIO()
    pstats.Stats(profile, stream=out).sort_stats(key).print_stats(45)
    print(out.getvalue())
