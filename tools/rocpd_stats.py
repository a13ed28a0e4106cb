#!/usr/bin/env python3
"""Per-kernel summary (the columns of `rocprofv3 --stats`' kernel_stats.csv) from a rocprofv3 rocpd database.

    rocprofv3 --kernel-trace --stats -d OUT -- python3 bench.py ...     # ROCm 7.2 writes OUT/<host>/<pid>_results.db
    python tools/rocpd_stats.py OUT/<host>/<pid>_results.db > profiles/rNN/<name>_kernel_stats.csv
"""
import collections
import csv
import sqlite3
import statistics
import sys


def main(db):
    cur = sqlite3.connect(db).cursor()
    durs = collections.defaultdict(list)
    for name, dur in cur.execute('select name, duration from kernels'):
        durs[name].append(dur)
    total = sum(sum(v) for v in durs.values())
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    # CallsFull / AverageFullNs: the launches that did their work — a kernel enqueued behind a solver's `done` flag returns at once
    # (microseconds), and the mean over ALL launches of such a kernel says nothing about the kernel: launches shorter than a fifth of
    # the kernel's median are left out of these two columns
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev', 'CallsFull', 'AverageFullNs'])
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        med = statistics.median(v)
        full = [x for x in v if x >= 0.2 * med] or v
        w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 6), round(100 * sum(v) / total, 2), min(v), max(v),
                    round(statistics.pstdev(v), 6) if len(v) > 1 else 0.0, len(full), round(sum(full) / len(full), 6)])


if __name__ == '__main__':
    main(sys.argv[1])
