#!/usr/bin/env python3
"""Per-kernel means of every counter in rocprofv3 --pmc counter_collection CSVs (several passes may be given).
    python tools/pmc_table.py pass1.csv [pass2.csv ...]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path, newline='') as fh:
        for row in csv.DictReader(fh):
            acc[row['Kernel_Name'].replace('void ', '')[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
names = sorted({c for k in acc.values() for c in k})
for k, cs in acc.items():
    print(k)
    for c in names:
        if c in cs:
            print(f'    {c:34s} {sum(cs[c]) / len(cs[c]):16.1f}  (x{len(cs[c])})')
