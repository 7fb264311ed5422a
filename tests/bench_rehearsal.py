#!/usr/bin/env python3
"""Rehearsal wrapper around bench.py's protocol (test infrastructure -- never a measurement).

    python tests/bench_rehearsal.py [--standin] [--backend gloo] [--share-devices] -- <bench.py arguments>

bench.py itself has no test switches: it always runs the HIP renderer, one rank per GPU, over RCCL.  This file calls
``bench.main`` with what a GPU-less (or one-GPU) box needs to run the SAME launcher, rendezvous, settle / warm-up / timed
protocol, per-rank blocks, gathers, max-over-ranks reduction and JSON lines:

  --standin         the renderer replaced by a CPU stand-in whose outputs are a function of the global ray index, so every
                    gathered frame can be checked exactly (lines say ``"data": "stand-in"``)
  --backend NAME    torch.distributed backend instead of "nccl" (gloo on CPU, or for several ranks sharing one GPU)
  --share-devices   LOCAL_RANK modulo the device count: several ranks on the one GPU of a test box (RCCL refuses that, so
                    only together with --backend gloo)

The self-launcher (``--gpus N`` with no launcher around it) starts its ranks from THIS file, so the ranks get the same
switches.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402


class StandInRenderer:
    """bench.HipRenderer's interface without a GPU: outputs are a function of the global ray index."""
    gpu = False
    FRAME = (37, 41)      # rays = 1517: ragged over 2, 3 and 8 ranks

    def __init__(self, precision, device, rank, world, kind='headline', collective=None):
        from simplenerf_amd import harness
        self.harness, self.rank, self.world = harness, rank, world
        self.first = rank * bench.RAYS_PER_GPU
        self.collective = world > 1 if collective is None else collective
        self.gather_marks = None

    @staticmethod
    def _outputs(first, count):
        idx = torch.arange(first, first + count, dtype=torch.float32)
        return {'rgb_fine': torch.stack([idx, 2 * idx, 3 * idx], 1), 'depth_fine': idx + 0.5}

    def local(self):
        return self._outputs(self.first, bench.RAYS_PER_GPU)

    def step(self):
        local = self.local()
        if self.collective:
            before, after = bench._Mark(False), bench._Mark(False)
            before.record()
            full = self.harness.gather_rays(local, self.world * bench.RAYS_PER_GPU, self.rank, self.world)
            after.record()
            if self.gather_marks is not None:
                self.gather_marks.append((before, after))
        else:
            full = local
        if self.rank == 0:
            ref = self._outputs(0, self.world * bench.RAYS_PER_GPU)
            assert all(torch.equal(full[k], ref[k]) for k in ref)
        return full

    def frame_camera(self, name):
        return {'resolution': self.FRAME}

    def frame(self, name):
        n = self.FRAME[0] * self.FRAME[1]
        first, count = self.harness.shard_range(n, self.rank, self.world)
        local = self._outputs(first, count)
        full = self.harness.gather_rays(local, n, self.rank, self.world) if self.collective else local
        if self.rank != 0:
            return None
        ref = self._outputs(0, n)
        assert all(torch.equal(full[k], ref[k]) for k in ref)
        return {'image': full['rgb_fine'].numpy(), 'depth': full['depth_fine'].numpy()}

    def frame_block(self, name, rays=65536):
        self._outputs(0, 64)

    def profile(self, capacity):
        pass

    def profile_reset(self):
        pass

    def profile_collect(self):
        return [], [], 0


def main():
    argv = sys.argv[1:]
    if '--' not in argv:
        raise SystemExit(__doc__)
    cut = argv.index('--')
    own, rest = argv[:cut], argv[cut + 1:]
    backend = 'nccl'
    if '--backend' in own:
        backend = own[own.index('--backend') + 1]
    # the self-launcher re-runs `sys.argv[1:]` on `script`: keep our own switches in front of the bench arguments
    bench.main(rest, renderer_cls=StandInRenderer if '--standin' in own else None, backend=backend,
               share_devices='--share-devices' in own, script=os.path.abspath(__file__))


if __name__ == '__main__':
    main()
