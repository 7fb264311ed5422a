"""Build the C-ABI shared library (hipcc, gfx950) in-tree: simplenerf_amd/libsimplenerf_hip.so.

    python -m simplenerf_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libsimplenerf_hip.so')
OBJ_DIR = os.path.join(CSRC, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: the elementwise stages must round exactly like the reference's separate torch/numpy ops;
# fused multiply-adds are written explicitly (fmaf) where they are wanted.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wall', '-Wno-unused-function', '-Wno-inline-asm',
         f'-I{INCLUDE}']


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith('.h')]
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, src[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([HIPCC, *FLAGS, '-c', os.path.join(CSRC, src), '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB])
    return LIB


if __name__ == '__main__':
    path = build_library(force='--force' in sys.argv, verbose=True)
    print(path)
