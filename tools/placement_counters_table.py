#!/usr/bin/env python3
"""Per-PANEL means of the rocprofv3 --pmc rows of symv_tiles_kernel written by tools/placement_counters.py runs:
    python tools/placement_counters_table.py K LAUNCHES_PER_PANEL_AND_SWEEP pass1.csv pass2.csv ...
Dispatch i of symv_tiles_kernel (in dispatch order) belongs to panel (i // L) % K."""
import csv
import sys
from collections import defaultdict


def main():
    K, L = int(sys.argv[1]), int(sys.argv[2])
    table = defaultdict(lambda: defaultdict(list))
    for path in sys.argv[3:]:
        rows = [r for r in csv.DictReader(open(path, newline='')) if 'symv_tiles' in r['Kernel_Name']]
        by_counter = defaultdict(list)
        for r in rows:
            by_counter[r['Counter_Name']].append(r)
        for cname, rs in by_counter.items():
            rs.sort(key=lambda r: int(r['Dispatch_Id']))
            for i, r in enumerate(rs):
                panel = (i // L) % K
                table[panel][cname].append(float(r['Counter_Value']))
                if cname == sorted(by_counter)[0] and r.get('Start_Timestamp'):
                    table[panel]['_ms[' + path.split('/')[-1] + ']'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    counters = sorted({c for v in table.values() for c in v})
    print('%-52s' % 'counter (mean per launch)' + ''.join('%16s' % f'panel {k}' for k in range(K)))
    for c in counters:
        print('%-52s' % c[:52] + ''.join('%16.4f' % (sum(table[k][c]) / max(len(table[k][c]), 1)) for k in range(K)))


if __name__ == '__main__':
    main()
