#!/usr/bin/env python3
"""Fold ONE rocprofv3 --pmc pass (any counter set) into a per-kernel table of mean counter values and durations.
    python tools/collect_pmc_generic.py <dir with p_counter_collection.csv> <out.json> "<command, for the record>" [kernel filter ...]
Derived where the counters are there: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);
lds_conflict_fraction = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles over all LDS-array cycles);
lds_active_fraction = SQ_LDS_IDX_ACTIVE / (4 x SQ_BUSY_CYCLES ...) is NOT derived: the counters' units differ by block (see
MI355X_MICROARCH.md, rocprofv3 PMC slots) -- raw means are kept instead."""
import csv
import json
import os
import re
import subprocess
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', name)


def main():
    src, dst, command = sys.argv[1], sys.argv[2], sys.argv[3]
    filters = sys.argv[4:]
    values, times, seen = defaultdict(lambda: defaultdict(list)), defaultdict(list), set()
    with open(os.path.join(src, 'p_counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            if filters and not any(x in k for x in filters):
                continue
            values[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                times[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    rows = []
    for k, counters in values.items():
        c = {n: sum(v) / len(v) for n, v in counters.items()}
        row = {'kernel': k, 'dispatches': len(times[k]), 'duration_us': round(sum(times[k]) / len(times[k]) / 1e3, 1), 'counters': c}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c.get('GRBM_GUI_ACTIVE'):
            row['mfma_busy_fraction'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024), 3)
        if c.get('SQ_LDS_IDX_ACTIVE'):
            row['lds_conflict_fraction'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'], 4)
        if c.get('SQ_WAVE_CYCLES'):
            for name, key in (('parked_at_waitcnt_or_barrier', 'SQ_WAIT_ANY'), ('issue_stalled', 'SQ_WAIT_INST_ANY'),
                              ('lds_issue_stalled', 'SQ_WAIT_INST_LDS'), ('issuing', 'SQ_ACTIVE_INST_ANY')):
                if key in c:
                    row[name] = round(c[key] / c['SQ_WAVE_CYCLES'], 3)
        rows.append(row)
    rows.sort(key=lambda r: -r['duration_us'] * r['dispatches'])
    head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__)))
    out = {'source': f'rocprofv3 --pmc <one set> --kernel-trace -- python3 {command}; folded by tools/collect_pmc_generic.py',
           'commit': head.stdout.strip() or 'unrecorded', 'kernels': rows}
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    for r in rows[:12]:
        print({k: v for k, v in r.items() if k != 'counters'}, {n: round(v) for n, v in r['counters'].items()})


if __name__ == '__main__':
    main()
