#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected in separate runs) into per-kernel HBM bytes.

    python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json [key=value ...]

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled; WRITE_SIZE is
exact for 16-B streaming stores.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline='') as fh:
        for row in csv.DictReader(fh):
            if row['Counter_Name'] == counter:
                acc[row['Kernel_Name']].append(float(row['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    meta = dict(kv.split('=', 1) for kv in sys.argv[4:])
    out = {'meta': meta, 'units': 'bytes per launch (mean over launches)',
           'correction': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read under-count); WRITE_SIZE KiB x 1024', 'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        short = k.split('(')[0].replace('void ', '')
        out['kernels'][short] = {'fetch_size_kib_raw': f, 'write_size_kib_raw': w, 'launches': max(nf, nw),
                                 'hbm_read_bytes': 2.0 * f * 1024.0, 'hbm_write_bytes': w * 1024.0,
                                 'hbm_bytes': 2.0 * f * 1024.0 + w * 1024.0}
    with open(sys.argv[3], 'w') as fh:
        json.dump(out, fh, indent=1)
    for k, v in out['kernels'].items():
        print(f"{k[:70]:70s} {v['hbm_bytes'] / 1e9:10.3f} GB/launch  x{v['launches']}")


if __name__ == '__main__':
    main()
