"""Training quality across seeds (VERDICT r3 "next" #6): the 16-bit modes train as well as fp32 as a statement about
DISTRIBUTIONS over seeds, not about one trajectory.  The long record -- 8 seeds x {fp32, f16x3, f16, bf16} x 5000 iterations
-- is profiles/r04_train_seeds.json (tools/train_seeds.py): final PSNR 40.3 +- 1.4 dB (fp32), 39.7 +- 1.4 (f16x3), 39.4 +- 1.2
(f16), 39.2 +- 1.4 (bf16); every mean inside fp32's own min..max, every difference under two standard errors; f16s8 (fp8 saved
activations; profiles/r04_train_seeds_f16s8.json, same seeds): 39.8 +- 1.9.  The same seeds on the round's final code
(r04_train_seeds_final_code.json): 40.3 / 40.7 / 40.3 / 39.6 / 40.1 -- the ordering of the first record was trajectory.  Here: the same runs, shorter (3 seeds x 1200
iterations), as a gate on the means with the pooled seed spread as the yardstick."""
import os
import statistics
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


def test_sixteen_bit_training_lands_inside_the_fp32_seed_spread():
    import train_demo
    from tests import util
    seeds, iterations = (0, 1, 2), 1200
    psnr = {p: [train_demo.run(iterations, p, False, s)['psnr_view0_after'] for s in seeds] for p in ('fp32', 'f16', 'bf16', 'f16s8', 'bf16s8')}
    mean = {p: statistics.fmean(v) for p, v in psnr.items()}
    # The yardstick is the seed-to-seed spread -- but not fp32's own from three runs: that estimate came out as 0.06, 0.34 and
    # 0.84 dB in three runs of this test (the trajectories change with every change of a summation order), while single runs of
    # any precision range over 24.4 .. 29.4 dB.  Pooled over the precisions (fifteen runs, ten degrees of freedom) it is stable;
    # the gate is three standard errors of a difference of two three-seed means.
    pooled = statistics.fmean(statistics.variance(v) for v in psnr.values()) ** 0.5
    allowed = 3.0 * pooled * (2.0 / len(seeds)) ** 0.5
    util.observe('train_quality', ', '.join(f"{p} {mean[p]:.2f} dB ({' '.join(f'{x:.1f}' for x in psnr[p])})" for p in psnr)
                 + f'; pooled seed stdev {pooled:.2f} dB [16-bit means within 3 standard errors = {allowed:.2f} dB of the fp32 mean]')
    assert mean['fp32'] > 20.0                      # the runs converge at all
    assert pooled < 2.0                             # ... and a seed does not decide by more than this
    for p in ('f16', 'bf16', 'f16s8', 'bf16s8'):
        assert abs(mean[p] - mean['fp32']) <= allowed, (p, mean, pooled)


def test_sixteen_bit_gradients_and_short_run_track_fp32_deterministically():
    """the seed-spread gate above allows a 16-bit mean ~2.4 dB off fp32's, so a real precision regression -- broken
    fp8 weight gradients, say -- could pass it.  This gate is deterministic: same initial weights, same batches, same draws
    (they are functions of seed, iteration and global row only) for every precision.  (i) The parameter gradients of the first
    iteration against fp32's, relative L2 over ALL 2.27 M parameters; (ii) the loss of iteration 0 (before any update) and the
    mean loss of the last five of 40 iterations against fp32's.  Tolerances are ~5x the figures observed on the round's code
    (printed in the summary): a wrong scale or a dropped operand moves a gradient by tens of per cent."""
    import torch
    from simplenerf_amd import harness, optim, synth
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    from tests import util
    dev = torch.device('cuda', 0)
    scene = synth.training_scene(0, 3, 96, 128, sparse_fraction=0.02)

    def run(precision, iterations=40):
        cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256, seed=3)
        cfg['sub_batch_size'] = 1280
        cfg['losses'] = synth.loss_configs(iter_weighted=False)
        torch.manual_seed(3)
        model = get_model(cfg, None).to(dev).train()
        batcher, losses = BatchAssembler(cfg, scene, dev), LossComputer(cfg)
        opt = optim.Adam(list(model.parameters()), lr=5e-4, betas=(0.9, 0.999))
        first_grads, curve = None, []
        for it in range(iterations):
            totals = harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'])
            curve.append(float(totals['TotalLoss']))
            if it == 0:      # (the gradients of iteration 0 survive the step: Adam reads them, zero_grad comes with the next iteration)
                first_grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double().cpu()
        return first_grads, curve

    ref_grads, ref_curve = run('fp32')
    assert all(v == v and v < 1e6 for v in ref_curve) and float(ref_grads.norm()) > 0
    # relative L2 of the first iteration's gradients / relative error of its loss / of the mean loss of iterations 35..39
    # (observed on the round's code: gradients 1.3e-5 / 8.8e-4 / 1.1e-3 / 2.3e-3 / 2.4e-3, first loss 0 / 2e-5 / 2e-5 / 4e-5 / 4e-5, tail 4e-4 .. 8e-4)
    gates = {'f16x3': (1e-4, 1e-5, 0.01), 'f16': (5e-3, 2e-4, 0.01), 'f16s8': (6e-3, 2e-4, 0.01), 'bf16': (1.2e-2, 4e-4, 0.01), 'bf16s8': (1.2e-2, 4e-4, 0.01)}
    text = []
    for precision, (grad_tol, loss_tol, end_tol) in gates.items():
        grads, curve = run(precision)
        rel = float((grads - ref_grads).norm() / ref_grads.norm())
        first = abs(curve[0] - ref_curve[0]) / abs(ref_curve[0])
        tail, ref_tail = sum(curve[-5:]) / 5, sum(ref_curve[-5:]) / 5
        end = abs(tail - ref_tail) / abs(ref_tail)
        text.append(f'{precision}: gradients {rel:.2e} [{grad_tol}], first loss {first:.1e} [{loss_tol}], mean loss of the last five {end:.1e} [{end_tol}]')
        assert rel <= grad_tol and first <= loss_tol and end <= end_tol, (precision, rel, first, end)
    assert ref_curve[-1] < ref_curve[0]              # the run trains
    util.observe('train_deterministic', '; '.join(text) + f'; fp32 loss {ref_curve[0]:.4f} -> {ref_curve[-1]:.4f}')
