"""Checkpoint files in the reference's format (src/Trainer01.py:352-381, src/Tester01.py:45-49): a ``torch.save``d
dictionary ``{'iteration_num', 'model_state_dict', 'optimizer_state_dict'}``.  The reference always wraps its model
in ``torch.nn.DataParallel`` (Trainer01.py:514, Tester01.py:42), so its parameter names carry a ``module.`` prefix;
this build runs one process per GPU without that wrapper, so the prefix is added on save and accepted (with or
without) on load.  Parameter names and shapes below the prefix are identical (models/SimpleNeRFHip01.py)."""
from __future__ import annotations

import os
from typing import Optional

import torch

PREFIX = 'module.'


def to_reference_names(state_dict: dict) -> dict:
    return {k if k.startswith(PREFIX) else PREFIX + k: v for k, v in state_dict.items()}


def from_reference_names(state_dict: dict) -> dict:
    return {k[len(PREFIX):] if k.startswith(PREFIX) else k: v for k, v in state_dict.items()}


def save_checkpoint(path, iter_num: int, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None) -> None:
    state = {'iteration_num': iter_num, 'model_state_dict': to_reference_names(model.state_dict())}
    if optimizer is not None:
        state['optimizer_state_dict'] = optimizer.state_dict()
    tmp = f'{path}.tmp'
    torch.save(state, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None,
                    map_location=None) -> int:
    """Loads model (and optimiser, when given and present) state; returns ``iteration_num``."""
    state = torch.load(path, map_location=map_location)   # tensors, dicts and ints only: loads under weights_only=True
    model.load_state_dict(from_reference_names(state['model_state_dict']))
    if optimizer is not None and 'optimizer_state_dict' in state:
        optimizer.load_state_dict(state['optimizer_state_dict'])
    return state['iteration_num']
