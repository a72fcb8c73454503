"""How long does the host take to ENQUEUE one training step, against the GPU time of the step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse() if hasattr(bench, 'parse') else None
args.no_cpu_baseline = True
exp = bench.build_experiment(args, None)
labeled = exp.infinite_iter(exp.train_dataset_loader); unlabeled = exp.infinite_iter(exp.unlabeled_dataset_loader)
for i in range(2): bench.one_step(exp, labeled, unlabeled, i)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    x, heads, knn = next(labeled); u = next(unlabeled)[0]
    exp.dnn_training_step(x, (heads, knn), 10 + i)
    t1 = time.perf_counter()
    exp.gan_training_step(x, (heads, knn), u, 10 + i)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f'host: dnn step {1e3*(t1-t0):.1f} ms, gan step {1e3*(t2-t1):.1f} ms; wait for GPU {1e3*(t3-t2):.1f} ms; total {1e3*(t3-t0):.1f} ms')
