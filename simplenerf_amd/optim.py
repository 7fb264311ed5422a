"""Optimiser of the training step on the HIP library: ``Adam(params, lr, betas)`` with the interface the reference's
trainer uses from ``torch.optim.Adam`` (src/Trainer01.py:80, :102, :293-295, :360, :377, :516-517): ``param_groups``
whose ``'lr'`` the trainer overwrites every iteration, ``zero_grad(set_to_none=True)``, ``step()``,
``state_dict()`` / ``load_state_dict()``.

It subclasses ``torch.optim.Optimizer`` for that bookkeeping only -- state layout (``step``, ``exp_avg``,
``exp_avg_sq`` per parameter) and ``param_groups`` keys are those of ``torch.optim.Adam``, so the
``optimizer_state_dict`` of a reference checkpoint loads here and vice versa.  The update itself is one HIP launch
per 64 tensors (snerf_adam_step), bit-identical to PyTorch's CPU single-tensor Adam.  No torch fallback.
"""
from __future__ import annotations

import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('the HIP Adam builds what the reference uses: weight_decay=0, amsgrad=False')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)

    # ``step`` counts live as Python ints while training (90 host tensors incremented one by one would cost more than
    # the update itself); the ``step`` tensors torch.optim.Adam keeps in its state are refreshed whenever the state
    # is read or replaced, so checkpoints carry them as usual.
    def _count(self, p) -> int:
        counts = self.__dict__.setdefault('_counts', {})
        if p not in counts:
            state = self.state[p]
            counts[p] = int(state['step']) if 'step' in state else 0
        return counts[p]

    def _sync_step_tensors(self):
        for p, count in self.__dict__.get('_counts', {}).items():
            if 'step' in self.state[p]:
                self.state[p]['step'].fill_(float(count))

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__['_counts'] = {}

    def graph_factors(self, step: int, lr: float):
        """(-(lr / (1 - beta1^step)), sqrt(1 - beta2^step)) as float32, evaluated in double like snerf_adam_step does: what a
        captured ``step_at`` reads from the device-resident iteration record."""
        import numpy
        self.require_single_schedule()
        beta1, beta2 = self.param_groups[0]['betas']
        bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
        return float(numpy.float32(-(lr / bc1))), float(numpy.float32(bc2 ** 0.5))

    @torch.no_grad()
    def step_at(self, record):
        """The update of ``step()`` with the step-dependent factors read from the device-resident iteration record (inside a
        captured graph: nothing here depends on the iteration).  Every parameter must hold a gradient and state; the step
        COUNTS are kept by the caller (``count_step``)."""
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            for p in ps:
                if len(self.state[p]) == 0:
                    self.state[p]['step'] = torch.tensor(0.0, dtype=torch.float32)
                    self.state[p]['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    self.state[p]['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            beta1, beta2 = group['betas']
            ops.adam_step([p.data for p in ps], [p.grad for p in ps], [self.state[p]['exp_avg'] for p in ps],
                          [self.state[p]['exp_avg_sq'] for p in ps], 0, 0.0, beta1, beta2, group['eps'], at=record)

    def count_step(self) -> int:
        """One optimiser step happened outside ``step()`` (a graph replay): advance every parameter's count; -> the new count."""
        counts = self.__dict__.setdefault('_counts', {})
        new = None
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                counts[p] = self._count(p) + 1
                new = counts[p] if new is None else new
                if counts[p] != new:
                    raise RuntimeError('graphed optimiser steps need every parameter at the same step count')
            torch.autograd.graph.increment_version([p for p in group['params'] if p.grad is not None])
        return int(new or 0)

    def next_count(self) -> int:
        """The step count the next graphed update runs at: the count of the parameters ``count_step`` advances (those holding a
        gradient) + 1.  They must agree -- ONE device-resident record serves every parameter of a replay."""
        found = None
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue                      # frozen / unused: count_step leaves it alone, so must this
                count = self._count(p)
                if found is None:
                    found = count
                elif count != found:
                    raise RuntimeError(f'graphed optimiser steps need every parameter at the same step count ({found} vs {count})')
        return (found or 0) + 1

    def require_single_schedule(self) -> None:
        """A graphed update reads ONE (step size, bias correction) record: every group must share betas and learning rate."""
        first = self.param_groups[0]
        for group in self.param_groups[1:]:
            if tuple(group['betas']) != tuple(first['betas']) or group['lr'] != first['lr']:
                raise RuntimeError('graphed optimiser steps need the same betas and learning rate in every parameter group')

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        counts = self.__dict__.setdefault('_counts', {})
        for group in self.param_groups:
            if group.get('weight_decay', 0) != 0 or group.get('amsgrad', False) or group.get('maximize', False):
                raise NotImplementedError('weight_decay / amsgrad / maximize are not built')
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0, dtype=torch.float32)
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                counts[p] = self._count(p) + 1
                by_step.setdefault(counts[p], []).append(p)
            beta1, beta2 = group['betas']
            for step, ps in by_step.items():
                ops.adam_step([p.data for p in ps], [p.grad for p in ps], [self.state[p]['exp_avg'] for p in ps],
                              [self.state[p]['exp_avg_sq'] for p in ps], step, group['lr'], beta1, beta2, group['eps'])
                # the kernel wrote the parameters behind autograd's back: bump their version counters, as an in-place
                # torch op would, so that anything keyed on them (the model's packed-weight cache) sees the change
                torch.autograd.graph.increment_version(ps)
        return loss
