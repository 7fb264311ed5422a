"""Loader of the TORCH_LIBRARY binding (csrc_torch/snerf_torch.cpp -> csrc_torch/build/snerf_torch_ext.so): after ``load()``
``torch.ops.snerf.render`` is the one-call render op with its C++ autograd node ("Ownership / errors").  Built by
``python -m simplenerf_amd.build`` through torch.utils.cpp_extension; there is no fallback here -- a model configured for this
binding fails loudly when the extension is missing (``hip_host_binding='ctypes'`` selects the torch-free binding instead)."""
from __future__ import annotations

import os
import sys

import torch

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc_torch', 'build', 'snerf_torch_ext.so')
_loaded = False


def available() -> bool:
    return os.path.exists(EXT_PATH)


def load():
    """-> torch.ops.snerf (loads the extension on first use; the HIP library is loaded first so that both bind to one copy)."""
    global _loaded
    if not _loaded:
        _lib.load()
        if not os.path.exists(EXT_PATH):
            # host glue only (no kernels): built on the spot with torch.utils.cpp_extension when a checkout carries the HIP
            # library but not this file (~30 s, needs the host compiler); a failure is raised, never papered over
            try:
                from . import build
                print(f'[simplenerf_amd] {EXT_PATH} is missing: building the TORCH_LIBRARY binding (torch.utils.cpp_extension)',
                      file=sys.stderr, flush=True)
                build.build_torch_extension()
            except Exception as error:
                raise RuntimeError(f'{EXT_PATH} is missing and could not be built ({error}); run `python -m simplenerf_amd.build`, or '
                                   f"set configs['model']['hip_host_binding'] = 'ctypes'") from error
        if not hasattr(torch.ops, 'snerf') or not hasattr(torch.ops.snerf, 'render'):
            torch.ops.load_library(EXT_PATH)
        version = int(torch.ops.snerf.abi_version())
        if version != _lib.ABI_VERSION:
            raise RuntimeError(f'{EXT_PATH}: built against ABI {version}, the Python binding expects {_lib.ABI_VERSION}; rebuild')
        _loaded = True
    return torch.ops.snerf
