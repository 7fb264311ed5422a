#!/usr/bin/env python3
"""Training quality across seeds (VERDICT r3 "next" #6): tools/train_demo.py's run -- the synthetic 3-view scene at 96x128,
1280-row batches, every stage on the device -- for SEEDS x {fp32, f16x3, f16[, bf16]}, final PSNR of every training view.
Writes one JSON with each run's PSNR and, per precision, mean / min / max / sample standard deviation over the seeds, so
that "the 16-bit mode trains as well as fp32" is a statement about distributions and not about one trajectory.

    python tools/train_seeds.py [iterations] [out.json] [precisions, comma-separated] [seeds, comma-separated]
"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_demo  # noqa: E402


def summarise(values):
    return {'mean': statistics.fmean(values), 'min': min(values), 'max': max(values),
            'stdev': statistics.stdev(values) if len(values) > 1 else 0.0, 'n': len(values)}


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    out = sys.argv[2] if len(sys.argv) > 2 else None
    precisions = sys.argv[3].split(',') if len(sys.argv) > 3 else ['fp32', 'f16x3', 'f16']
    seeds = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0, 1, 2]
    report = {'iterations': iters, 'seeds': seeds, 'runs': [], 'summary': {}}
    for precision in precisions:
        finals = []
        for seed in seeds:
            r = train_demo.run(iters, precision, False, seed)
            r.pop('curve')
            report['runs'].append(r)
            finals.append(r['psnr_view0_after'])
            print(f"{precision} seed {seed}: view 0 {r['psnr_view0_after']:.2f} dB, all views "
                  + ' '.join(f'{v:.2f}' for v in r['psnr_all_views_after']) + f" ({r['seconds']:.0f} s)", flush=True)
            if out:
                with open(out, 'w') as f:
                    json.dump(report, f, indent=1)
        report['summary'][precision] = {'psnr_view0_after': summarise(finals)}
    base = report['summary'].get('fp32')
    if base:
        spread = base['psnr_view0_after']
        for precision, row in report['summary'].items():
            row['mean_minus_fp32_mean_db'] = row['psnr_view0_after']['mean'] - spread['mean']
            row['within_fp32_range'] = spread['min'] - 1e-9 <= row['psnr_view0_after']['mean'] <= spread['max'] + 1e-9
    print(json.dumps(report['summary'], indent=1))
    if out:
        with open(out, 'w') as f:
            json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
