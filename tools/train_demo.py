#!/usr/bin/env python3
"""End-to-end training sanity run on one MI355X: the synthetic 3-view plane scene (synth.training_scene at 96x128),
every stage on the device (BatchAssembler -> SimpleNeRFHip -> LossComputer -> optim.Adam with the NeRF decay), a few
hundred iterations.  Prints the loss curve and the PSNR of a training view rendered before and after.
    python tools/train_demo.py [iterations]            (SNERF_PREC=f16x3 for the split-precision kernels, f16 for the 16-bit mode;
                                                        SNERF_GRAPH=1 replays the pass from one HIP graph; SNERF_SEED=n: initial
                                                        weights, epoch order and training draws of seed n)
tools/train_seeds.py runs it over seeds x precisions and writes the spread."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplenerf_amd import harness, optim, synth  # noqa: E402
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler  # noqa: E402
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer  # noqa: E402
from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402

DEV = torch.device('cuda', 0)


def psnr_of_view(model, scene, view):
    cam = {'resolution': scene['resolution'], 'intrinsic': scene['intrinsics'][view], 'pose': scene['poses'][view],
           'near': scene['near'], 'far': scene['far'], 'near_ndc': 0.0, 'far_ndc': 1.0}
    model.eval()
    rgb = harness.render_frame(model, cam, True, DEV, keys=('rgb_fine',))['rgb_fine']
    model.train()
    target = torch.as_tensor(scene['images'][view]).reshape(-1, 3).to(DEV)
    mse = float(torch.mean((rgb - target) ** 2))
    return -10 * math.log10(max(mse, 1e-12))


def run(iters=300, precision='fp32', graph=False, seed=0):
    """-> the record ``main`` prints.  ``seed`` selects the initial weights (torch.manual_seed), the epoch order and the training
    draws (configs['seed']); the scene is the same for every seed."""
    cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256, seed=seed)
    cfg['sub_batch_size'] = 1280
    cfg['losses'] = synth.loss_configs(iter_weighted=False)      # consistency losses on from the first iteration
    scene = synth.training_scene(0, 3, 96, 128, sparse_fraction=0.02)
    torch.manual_seed(seed)
    model = get_model(cfg, None).to(DEV).train()
    batcher = BatchAssembler(cfg, scene, DEV)
    losses = LossComputer(cfg)
    opt = optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'], betas=(0.9, 0.999))
    decayer = get_lr_decayer(cfg)
    before = psnr_of_view(model, scene, 0)
    curve = []
    graphed = None
    if graph:
        graphed = harness.GraphedTrainStep(model, losses, batcher.get_next_batch(0), sub_batch_size=cfg['sub_batch_size'])
        batcher = BatchAssembler(cfg, scene, DEV)       # restart the index stream after the sample batch
    worst = 0.0
    t0 = time.perf_counter()
    for it in range(iters):
        for group in opt.param_groups:
            group['lr'] = decayer.get_updated_learning_rate(it)
        if graphed is None:
            totals = harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'])
        else:
            totals = graphed(batcher.get_next_batch(it))
            opt.step()
        if it % 50 == 0:
            value = float(totals['TotalLoss'])
            if not math.isfinite(value):
                raise SystemExit(f'non-finite loss at iteration {it}')
            worst = max(worst, value)
        if it % max(1, iters // 10) == 0 or it == iters - 1:
            curve.append({'iter': it, 'TotalLoss': float(totals['TotalLoss']), 'MSE01': float(totals['MSE01'])})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = psnr_of_view(model, scene, 0)
    views = [psnr_of_view(model, scene, v) for v in range(len(scene['poses']))]
    return {'precision': precision, 'graph': graphed is not None, 'seed': seed, 'iterations': iters, 'seconds': dt,
            'psnr_view0_before': before, 'psnr_view0_after': after, 'psnr_all_views_after': views, 'curve': curve}


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    print(json.dumps(run(iters, os.environ.get('SNERF_PREC', 'fp32'), os.environ.get('SNERF_GRAPH') == '1',
                         int(os.environ.get('SNERF_SEED', '0'))), indent=1))


if __name__ == '__main__':
    main()
