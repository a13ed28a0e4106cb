#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter CSVs (one or several passes): rows = kernel (+ grid size, which tells the
whole-panel launch from a share), columns = counters; the launch duration of the same dispatches beside them.

    python tools/pmc_kernel_table.py pass1.csv pass2.csv ...
"""
import csv
import sys
from collections import defaultdict

KEEP = ('symv_tiles', 'probe_read', 'gram_mfma', 'symv_reduce')


def main():
    table = defaultdict(lambda: defaultdict(list))
    for path in sys.argv[1:]:
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                k = row['Kernel_Name'].split('(')[0].replace('void ', '')
                if not any(t in k for t in KEEP):
                    continue
                key = (k[:44], int(row['Grid_Size']))
                table[key][row['Counter_Name']].append(float(row['Counter_Value']))
                if 'End_Timestamp' in row and row.get('Start_Timestamp'):
                    table[key]['_ms'].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6)
    counters = sorted({c for v in table.values() for c in v})
    for key in sorted(table):
        print(f'{key[0]}  grid={key[1]}')
        for c in counters:
            vals = table[key].get(c)
            if vals:
                print(f'    {c:48s} mean {sum(vals) / len(vals):16.4f}   min {min(vals):16.4f}   max {max(vals):16.4f}   n={len(vals)}')


if __name__ == '__main__':
    main()
