#!/usr/bin/env python3
"""Fold separate rocprofv3 --pmc passes of one command into a per-kernel table (duration, HBM bytes and rate, matrix-pipe
and wave-state fractions).

On the GPU box, one pass per counter set (--pmc never together with the trace domains gpurun refuses):
    cd /tmp && export TMPDIR=/tmp
    for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/<tag>_${set%% *} -o p -- python3 <command>
    done
here:  python tools/collect_pmc_kernels.py gpurun_out/<tag> profiles/<name>.json "<command, for the record>" [kernel-name filter ...]
       [--iterations N|auto]    the command ran N iterations of the step (auto: half the adam_kernel dispatches): adds `hbm_gb_per_iteration` = sum over ALL kernels of
                           (HBM bytes per dispatch x dispatches) / N -- what `bench.py --train` reports as its `traffic`

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of a wide streaming read, WRITE_SIZE is
exact -- MI355X_MICROARCH.md, HBM).  Fractions: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs);
wave states relative to SQ_WAVE_CYCLES."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', name)


def load(path):
    values, times = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    with open(path) as f:
        seen = set()
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            values[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                times[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return values, times


def main():
    tag, dst, command = sys.argv[1], sys.argv[2], sys.argv[3]
    filters = sys.argv[4:]
    iterations = None
    if '--iterations' in filters:
        at = filters.index('--iterations')
        iterations = filters[at + 1]          # a number, or `auto`: one iteration = two launches of adam_kernel (counted below)
        filters = filters[:at] + filters[at + 2:]
    sq, sq_t = load(f'{tag}_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv')
    fetch, fetch_t = load(f'{tag}_FETCH_SIZE/p_counter_collection.csv')
    write, write_t = load(f'{tag}_WRITE_SIZE/p_counter_collection.csv')
    mean = lambda xs: sum(xs) / len(xs)
    rows = []
    every_kernel_bytes = 0.0
    fetch_bytes = sum(2 * sum(fetch[k]['FETCH_SIZE']) * 1024 for k in fetch)
    write_bytes = sum(sum(write[k]['WRITE_SIZE']) * 1024 for k in write)
    for k in fetch:
        if k in write:
            every_kernel_bytes += (2 * sum(fetch[k]['FETCH_SIZE']) + sum(write[k]['WRITE_SIZE'])) * 1024
    for k in sq:
        if filters and not any(f in k for f in filters):
            continue
        c = {n: mean(v) for n, v in sq[k].items()}
        us = mean(sq_t[k]) / 1e3
        hbm = (2 * mean(fetch[k]['FETCH_SIZE']) + mean(write[k]['WRITE_SIZE'])) * 1024 if k in fetch and k in write else None
        row = {'kernel': k, 'dispatches': len(sq_t[k]), 'duration_us': round(us, 1),
               'mfma_busy_fraction': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024), 3)}
        if c.get('SQ_WAVE_CYCLES'):
            w = c['SQ_WAVE_CYCLES']
            row.update({'parked_at_waitcnt_or_barrier': round(c['SQ_WAIT_ANY'] / w, 3),
                        'issue_stalled': round(c['SQ_WAIT_INST_ANY'] / w, 3), 'issuing': round(c['SQ_ACTIVE_INST_ANY'] / w, 3)})
        if hbm is not None:
            row.update({'hbm_bytes': hbm, 'hbm_tb_per_s': round(hbm / (us * 1e-6) / 1e12, 2)})
        rows.append(row)
    rows.sort(key=lambda r: -r['duration_us'] * r['dispatches'])
    out = {'source': f'rocprofv3 --pmc <set> --kernel-trace -- python3 {command}  (three passes: SQ set | FETCH_SIZE | WRITE_SIZE; '
                     'durations from the SQ pass, averages over the dispatches of each kernel); folded by tools/collect_pmc_kernels.py',
           'correction': 'HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (see pmc_traffic.json)', 'kernels': rows}
    import os
    import subprocess
    head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    out['commit'] = head.stdout.strip() or 'unrecorded'
    if iterations:
        if iterations == 'auto':
            # the passes are separate runs, and a run's settle phase is timed, not counted: each pass has its own iteration count
            def count(times):
                adam = [len(times[k]) for k in times if k.endswith('adam_kernel')]
                if not adam:
                    raise SystemExit('--iterations auto: no adam_kernel dispatches in the trace')
                return adam[0] // 2
            out['iterations'] = {'sq_pass': count(sq_t), 'fetch_pass': count(fetch_t), 'write_pass': count(write_t)}
            out['hbm_gb_per_iteration'] = (fetch_bytes / count(fetch_t) + write_bytes / count(write_t)) / 1e9
        else:
            iterations = int(iterations)
            out['iterations'] = iterations
            out['hbm_gb_per_iteration'] = every_kernel_bytes / iterations / 1e9
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    for r in rows:
        print(r)


if __name__ == '__main__':
    main()
