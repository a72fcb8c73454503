"""Laboratory: the training iteration's chains on CU-masked streams (hipExtStreamCreateWithCUMask) -- does giving every chain
its own part of the GPU make them overlap more than four unmasked streams do?   python scratch/cu_mask_lab.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    """bits: iterable of 256 booleans (CU i enabled)."""
    words = [0] * 8
    for i, on in enumerate(bits):
        if on:
            words[i // 32] |= 1 << (i % 32)
    array = (ctypes.c_uint32 * 8)(*words)
    handle = ctypes.c_void_p()
    status = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, array)
    assert status == 0, status
    return torch.cuda.ExternalStream(handle.value)


def pattern(name):
    """(main, penalty, dnn, aux) masks of 256 CUs."""
    every = lambda f: [bool(f(i)) for i in range(256)]
    if name == 'interleaved 2:1:1':          # of every four consecutive CUs: two main, one penalty, one DNN (+ aux shares DNN's)
        return every(lambda i: i % 4 < 2), every(lambda i: i % 4 == 2), every(lambda i: i % 4 == 3), every(lambda i: i % 4 == 3)
    if name == 'interleaved 1:1 + all':      # main and penalty split the GPU, DNN and aux unrestricted
        return every(lambda i: i % 2 == 0), every(lambda i: i % 2 == 1), None, None
    if name == 'halves 1:1 + all':
        return every(lambda i: i < 128), every(lambda i: i >= 128), None, None
    if name == 'main all, sides quarter':
        return None, every(lambda i: i % 4 == 0), every(lambda i: i % 4 == 1), every(lambda i: i % 4 == 2)
    if name == 'interleaved 5:3 + all':
        return every(lambda i: i % 8 < 5), every(lambda i: i % 8 >= 5), None, None
    return None, None, None, None


def run(name, steps=10, warmup=3):
    sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline']
    args = bench.parse()
    experiment = bench.build_experiment(args, None)
    labeled = experiment.infinite_iter(experiment.train_dataset_loader)
    unlabeled = experiment.infinite_iter(experiment.unlabeled_dataset_loader)
    main, penalty, dnn, aux = pattern(name)
    main_stream = masked_stream(main) if main else torch.cuda.current_stream()
    if penalty: experiment._gp_stream = masked_stream(penalty)
    if dnn: experiment._dnn_stream = masked_stream(dnn)
    if aux: experiment._aux_stream = masked_stream(aux)
    with torch.cuda.stream(main_stream):
        for step in range(warmup):
            bench.one_step(experiment, labeled, unlabeled, step)
        experiment.join_dnn_stream(); torch.cuda.synchronize()
        start = time.perf_counter()
        for step in range(steps):
            bench.one_step(experiment, labeled, unlabeled, warmup + step)
        experiment.join_dnn_stream(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - start
    print(f'{name:28s}: {16 * steps / elapsed:7.2f} images/s  {1e3 * elapsed / steps:7.2f} ms', flush=True)


if __name__ == '__main__':
    for name in sys.argv[1:] or ['unmasked', 'interleaved 1:1 + all', 'halves 1:1 + all', 'interleaved 2:1:1', 'main all, sides quarter',
                                 'interleaved 5:3 + all', 'unmasked']:
        run(name)
