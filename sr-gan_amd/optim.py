"""Adam on a flat parameter arena: one HIP streaming kernel per network per step.

Same defaults, update rule and ``state_dict`` layout as ``torch.optim.Adam`` as the reference constructs it
(srgan.py:131-138: betas (0.9, 0.999), eps 1e-8, coupled L2 ``weight_decay``, no amsgrad), so checkpoints
written by either side load in the other (reference srgan.py:88-97,221-257)."""
import torch

from . import _lib
from . import functional as F


class Adam:
    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        self.arena = arena
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                  params=list(range(len(arena.parameters))))]
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self.step_count = 0
        self.device_state = None      # {int32 step, 2 floats}: set by count_on_device() for HIP-graph replay

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def count_on_device(self):
        """Keep the update count on the device from now on (``srgan_adam_step_counted``): a captured ``step()`` then
        replays as the NEXT update every time.  ``step_count`` stays the host's mirror for checkpoints."""
        if self.device_state is None:
            self.device_state = torch.zeros(4, dtype=torch.int32, device=self.arena.data.device)
        self.device_state[0] = self.step_count
        return self

    def step(self):
        group = self.param_groups[0]
        self.step_count += 1
        self.arena.version = getattr(self.arena, 'version', 0) + 1      # (raw-pointer update: torch's version counters do not see it)
        if self.device_state is not None:
            _lib.check(_lib.library().srgan_adam_step_counted(
                self.arena.data.data_ptr(), self.arena.grad.data_ptr(), self.exp_avg.data_ptr(),
                self.exp_avg_sq.data_ptr(), self.arena.numel, group['lr'], group['betas'][0], group['betas'][1],
                group['eps'], group['weight_decay'], self.device_state.data_ptr(), F._stream()),
                'srgan_adam_step_counted')
        else:
            _lib.check(_lib.library().srgan_adam_step(
                self.arena.data.data_ptr(), self.arena.grad.data_ptr(), self.exp_avg.data_ptr(),
                self.exp_avg_sq.data_ptr(), self.arena.numel, group['lr'], group['betas'][0], group['betas'][1],
                group['eps'], group['weight_decay'], self.step_count, F._stream()), 'srgan_adam_step')
        if getattr(self.arena, 'shadows', None):          # the 16-bit operand forms of these weights, on the update's own stream
            from . import blocked16
            blocked16.refresh(self.arena)

    # ---- torch.optim.Adam compatible checkpoint format ------------------------------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for index, (offset, size, p) in enumerate(zip(self.arena.offsets, self.arena.sizes,
                                                          self.arena.parameters)):
                state[index] = {'step': torch.tensor(float(self.step_count)),
                                'exp_avg': self.exp_avg[offset:offset + size].view(p.shape).clone(),
                                'exp_avg_sq': self.exp_avg_sq[offset:offset + size].view(p.shape).clone()}
        groups = [{k: v for k, v in self.param_groups[0].items()}]
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, state_dict):
        group = state_dict['param_groups'][0]
        for key in ('lr', 'betas', 'eps', 'weight_decay'):
            if key in group:
                self.param_groups[0][key] = tuple(group[key]) if key == 'betas' else group[key]
        steps = set()
        for index, entry in state_dict['state'].items():
            offset, size = self.arena.offsets[int(index)], self.arena.sizes[int(index)]
            self.exp_avg[offset:offset + size].copy_(entry['exp_avg'].reshape(-1))
            self.exp_avg_sq[offset:offset + size].copy_(entry['exp_avg_sq'].reshape(-1))
            steps.add(int(float(entry['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ; the flat Adam kernel needs one step count')
        self.step_count = steps.pop() if steps else 0
        if self.device_state is not None:
            self.device_state[0] = self.step_count
