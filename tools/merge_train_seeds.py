#!/usr/bin/env python3
"""Fold several tools/train_seeds.py outputs (one GPU call each) into ONE record: per precision mean / min / max / sample
standard deviation of the final PSNR over the seeds, and against fp32 the difference of the means with the standard error of
that difference (sqrt(s_a^2 / n_a + s_b^2 / n_b)).
    python tools/merge_train_seeds.py out.json part1.json part2.json ..."""
import json
import statistics
import sys


def main():
    out, parts = sys.argv[1], sys.argv[2:]
    runs, iterations = [], None
    for path in parts:
        with open(path) as f:
            part = json.load(f)
        iterations = part['iterations'] if iterations is None else iterations
        assert part['iterations'] == iterations, 'parts of different lengths'
        runs.extend(part['runs'])
    by = {}
    for r in runs:
        by.setdefault(r['precision'], []).append(r)
    summary = {}
    for precision, rs in by.items():
        v0 = [r['psnr_view0_after'] for r in rs]
        views = [statistics.fmean(r['psnr_all_views_after']) for r in rs]
        summary[precision] = {
            'n': len(rs), 'seeds': [r['seed'] for r in rs],
            'psnr_view0': {'mean': statistics.fmean(v0), 'stdev': statistics.stdev(v0), 'min': min(v0), 'max': max(v0)},
            'psnr_mean_of_views': {'mean': statistics.fmean(views), 'stdev': statistics.stdev(views)},
            'seconds_per_run': statistics.fmean(r['seconds'] for r in rs)}
    base = summary.get('fp32')
    if base:
        for row in summary.values():
            diff = row['psnr_view0']['mean'] - base['psnr_view0']['mean']
            se = (row['psnr_view0']['stdev'] ** 2 / row['n'] + base['psnr_view0']['stdev'] ** 2 / base['n']) ** 0.5
            row['mean_minus_fp32_db'] = diff
            row['standard_error_of_that_difference_db'] = se
            row['difference_in_standard_errors'] = diff / se if se else 0.0
            row['mean_within_fp32_min_max'] = base['psnr_view0']['min'] - 1e-9 <= row['psnr_view0']['mean'] <= base['psnr_view0']['max'] + 1e-9
    record = {'what': f'tools/train_seeds.py runs folded by tools/merge_train_seeds.py: tools/train_demo.py (synthetic 3-view scene 96x128, '
                      f'1280-row batches, {iterations} iterations, every stage on the device) per seed and precision on one MI355X; final PSNR of '
                      'training view 0 (and of all three views) per run, and per precision mean / min / max / sample standard deviation over the seeds',
              'iterations': iterations, 'parts': parts, 'runs': runs, 'summary': summary}
    with open(out, 'w') as f:
        json.dump(record, f, indent=1)
    for precision, row in summary.items():
        p = row['psnr_view0']
        print(f"{precision}: {p['mean']:.2f} +- {p['stdev']:.2f} dB [{p['min']:.2f}, {p['max']:.2f}] over {row['n']} seeds"
              + (f"; mean - fp32 = {row['mean_minus_fp32_db']:+.2f} +- {row['standard_error_of_that_difference_db']:.2f}" if base else ''))


if __name__ == '__main__':
    main()
