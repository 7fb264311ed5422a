"""Build a DIAGNOSTIC variant of the library: same sources and flags plus the given -D switches, into
gpurun_abl_<name>.so at the repo root (git-ignored, travels to the GPU box with gpurun).   usage:
    python tools/probes/build_variant.py clock -DSNERF_CLOCK_STAMP
Every variant is compiled with -DSNERF_PROBE_BUILD (csrc/probe_guard.h refuses the switches without it, build.py refuses them
for the shipped library's name).  Variants are loaded by the probes through _lib.LIB_PATH; nothing in the package or the tests ever loads one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simplenerf_amd import build
name, rest = sys.argv[1], sys.argv[2:]
only = ()
if '--only' in rest:          # --only stem[,stem...]: compile just these sources with the flags, link the shipped objects for the rest
    at = rest.index('--only')
    only = tuple(rest[at + 1].split(','))
    rest = rest[:at] + rest[at + 2:]
flags = ['-DSNERF_PROBE_BUILD'] + rest
out = os.path.join(ROOT, f'gpurun_abl_{name}.so')
print(build.build_library(extra_flags=flags, output=out, obj_dir=os.path.join('/tmp', f'snerf_variant_{name}'), only=only))
