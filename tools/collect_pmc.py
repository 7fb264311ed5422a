#!/usr/bin/env python3
"""Fold the rocprofv3 counter CSVs of three separate --pmc passes of bench.py into profiles/pmc_traffic.json.

On the GPU box (one pass per counter set; --pmc is never combined with the trace domains gpurun refuses):
    cd /tmp && export TMPDIR=/tmp
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${set%% *} -o p -- \
            python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 2
    done
then here:  python tools/collect_pmc.py gpurun_out profiles/pmc_traffic.json

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE count KiB, and on gfx950 FETCH_SIZE reports
half of a wide (16 B per lane) streaming read (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact.
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).
"""
import csv
import json
import os
import sys
from collections import defaultdict

KERNEL = 'mlp_forward_kernel'


def load(path):
    rows = defaultdict(list)      # (counter, grid size) -> values, per dispatch of the fused MLP kernel
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL in r['Kernel_Name'] and 'f16x3' not in r['Kernel_Name']:
                rows[(r['Counter_Name'], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return rows


def main():
    src, dst = sys.argv[1], sys.argv[2]
    data = {}
    for sub in ('pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_SQ_VALU_MFMA_BUSY_CYCLES'):
        data.update(load(os.path.join(src, sub, 'p_counter_collection.csv')))
    grids = sorted({g for _, g in data})
    mean = lambda c, g: sum(data[(c, g)]) / len(data[(c, g)])
    out = {
        'source': 'rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 2 '
                  '(three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES); '
                  'folded by tools/collect_pmc.py',
        'correction': 'HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a wide (16 B/lane) '
                      'streaming read (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact',
        'mfma_busy_definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)',
        'per_launch': [],
    }
    total = 0.0
    for g in grids:
        hbm = (2 * mean('FETCH_SIZE', g) + mean('WRITE_SIZE', g)) * 1024
        busy = mean('SQ_VALU_MFMA_BUSY_CYCLES', g) / (mean('GRBM_GUI_ACTIVE', g) / 8 * 1024)
        total += hbm
        out['per_launch'].append({'kernel': KERNEL, 'grid_threads': g, 'samples': g // 2, 'dispatches': len(data[('FETCH_SIZE', g)]),
                                  'FETCH_SIZE_KiB': mean('FETCH_SIZE', g), 'WRITE_SIZE_KiB': mean('WRITE_SIZE', g),
                                  'hbm_bytes': hbm, 'mfma_busy_fraction': busy,
                                  'GRBM_GUI_ACTIVE': mean('GRBM_GUI_ACTIVE', g), 'SQ_BUSY_CYCLES': mean('SQ_BUSY_CYCLES', g)})
    out['mlp_forward_hbm_bytes_per_launch'] = total / len(grids)
    import subprocess
    head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    out['commit'] = (sys.argv[3] if len(sys.argv) > 3 else head.stdout.strip()) or 'unrecorded'      # the tree the passes ran on
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'source'}, indent=1))


if __name__ == '__main__':
    main()
