"""SRGAN_POISON_EMPTY=1 python scratch/debug_poison.py : name the first operation whose result contains a NaN."""
import sys, os, traceback
os.environ['SRGAN_POISON_EMPTY'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import srgan_amd
from srgan_amd import functional as F, tape
seen = [0]
real_out = F._out
def checked_out(data, inputs, backward, name):
    if seen[0] < 3 and bool(torch.isnan(data).any()):
        seen[0] += 1
        print('NaN in result of', name, tuple(data.shape), 'inputs', [tuple(v.shape) if v is not None else None for v in inputs],
              'input NaN', [bool(torch.isnan(v.data).any()) if v is not None else None for v in inputs])
        print(''.join(traceback.format_stack(limit=8)[:-1]))
    return real_out(data, inputs, backward, name)
F._out = checked_out
import test_steps_gpu as T
T.test_tiny_dcgan_with_active_gradient_penalty(srgan_amd, False)
