#!/usr/bin/env python3
"""Steady-state kernel time per solver step from TWO kernel-trace summaries of the same command with different --steps:

    python tools/trace_diff.py stats_A.csv stepsA stats_B.csv stepsB [product-kernel-prefix] > per_step.csv

(total_B - total_A) / (stepsB - stepsA) per kernel: everything that happens once per run (Gram build, features, first full
preconditioner Gram, warm-up) cancels.  With a product-kernel prefix the table ends with the split product / everything else.
"""
import csv
import sys


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row['Name']] = (int(float(row['Calls'])), float(row['TotalDurationNs']))
    return out


def main(a_path, a_steps, b_path, b_steps, prefix=None):
    a, b = load(a_path), load(b_path)
    ds = float(b_steps) - float(a_steps)
    rows = []
    for name in sorted(set(a) | set(b)):
        ca, ta = a.get(name, (0, 0.0))
        cb, tb = b.get(name, (0, 0.0))
        if cb == ca and abs(tb - ta) < 1e3 * ds:
            continue
        rows.append((name, (cb - ca) / ds, (tb - ta) / ds / 1e6))
    rows.sort(key=lambda r: -r[2])
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(['Name', 'LaunchesPerStep', 'MsPerStep'])
    for r in rows:
        w.writerow([r[0], round(r[1], 3), round(r[2], 5)])
    total = sum(r[2] for r in rows)
    w.writerow(['TOTAL', round(sum(r[1] for r in rows), 3), round(total, 5)])
    if prefix:
        prod = sum(r[2] for r in rows if r[0].startswith(prefix))
        w.writerow([f'{prefix}* (the panel product)', round(sum(r[1] for r in rows if r[0].startswith(prefix)), 3), round(prod, 5)])
        w.writerow(['everything else', round(sum(r[1] for r in rows if not r[0].startswith(prefix)), 3), round(total - prod, 5)])


if __name__ == '__main__':
    main(*sys.argv[1:])
