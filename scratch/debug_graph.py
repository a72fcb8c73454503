import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import test_step_graph_gpu as t
import srgan_amd
from srgan_amd import graph as G, functional as F
used = {}
capturing = [False]
real_call = F._call
def pointers_of(value, out):
    if isinstance(value, int) and value > (1 << 40):
        out.append(value)
    elif isinstance(value, ctypes.Structure):
        for name, _ in value._fields_:
            pointers_of(getattr(value, name), out)
def logged_call(name, *args):
    if capturing[0]:
        import traceback
        found = []
        for a in args[:-1]:
            pointers_of(a, found)
        for p in found:
            used.setdefault(p, (name, ''.join(traceback.format_stack(limit=6)[:-1])))
    return real_call(name, *args)
F._call = logged_call
original_capture = G.CapturedIteration.capture
def capture(self, *a):
    capturing[0] = True
    try:
        return original_capture(self, *a)
    finally:
        capturing[0] = False
G.CapturedIteration.capture = capture
original = G.CapturedIteration.run
def traced(self, x, labels, u, step):
    e = self.experiment
    canaries = []
    if step == 3:
        # occupy every cached free block of the default pool (and some fresh memory) with a pattern
        for size in [1 << k for k in range(26, 8, -1)]:
            for _ in range(64):
                before = torch.cuda.memory_reserved()
                c = torch.full((size // 4,), 12345.0, device='cuda')
                if torch.cuda.memory_reserved() > before and size < (1 << 22):
                    del c
                    break
                canaries.append(c)
        torch.cuda.synchronize()
        print('canaries', len(canaries), sum(c.numel() * 4 for c in canaries) / 1e6, 'MB')
        ranges = sorted((c.data_ptr(), c.data_ptr() + c.numel() * 4) for c in canaries)
        import bisect
        starts = [r[0] for r in ranges]
        shown = 0
        for p, (name, stack) in used.items():
            i = bisect.bisect_right(starts, p) - 1
            if i >= 0 and p < ranges[i][1] and shown < 6:
                shown += 1
                print('STALE captured pointer', hex(p), name); print(stack)
    original(self, x, labels, u, step)
    torch.cuda.synchronize()
    for c in canaries:
        bad = (c != 12345.0).nonzero()
        if bad.numel():
            lo = c.data_ptr() + 4 * int(bad.min()); hi = c.data_ptr() + 4 * int(bad.max())
            print('CORRUPTED canary', hex(c.data_ptr()), c.numel() * 4, 'bytes; touched', hex(lo), '..', hex(hi), int(bad.numel()), 'elements')
            for p, (name, stack) in used.items():
                if c.data_ptr() <= p < c.data_ptr() + c.numel() * 4:
                    print('   captured pointer', hex(p), name); print(stack)
    print('after step', step, {n: float(getattr(e, n)._srgan_arena.data.abs().max()) for n in ('D', 'G', 'DNN')})
G.CapturedIteration.run = traced
e, losses = t._run(True, 4)
