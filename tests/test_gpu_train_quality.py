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
