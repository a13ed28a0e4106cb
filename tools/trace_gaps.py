#!/usr/bin/env python3
"""Where does a stream sit idle?  From a rocprofv3 kernel-trace database: the time between the end of one kernel and the start of
the next (same queue), summed per (previous kernel -> next kernel) transition.

    rocprofv3 --kernel-trace -d OUT -- python3 ...            # OUT/<host>/<pid>_results.db
    python tools/trace_gaps.py OUT/<host>/<pid>_results.db [skip_first_ms] > gaps.txt

Prints busy time, idle time, and the transitions that carry the idle time (count, total, mean).  Kernel names are cut at '('.
"""
import collections
import sqlite3
import sys


def main(db, skip_ms='0'):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    start, end = ('start', 'end') if 'start' in cols else ('start_timestamp', 'end_timestamp')
    rows = list(cur.execute(f'select name, {start}, {end} from kernels order by {start}'))
    if not rows:
        print('no kernels')
        return
    t0 = rows[0][1] + float(skip_ms) * 1e6
    rows = [r for r in rows if r[1] >= t0]
    busy = sum(e - s for _, s, e in rows)
    gaps = collections.defaultdict(lambda: [0, 0])
    idle = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g <= 0:
            continue
        idle += g
        key = (n0.split('(')[0][-48:], n1.split('(')[0][-48:])
        gaps[key][0] += 1
        gaps[key][1] += g
    span = rows[-1][2] - rows[0][1]
    print(f'kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {idle / 1e6:.3f} ms ({100 * idle / span:.1f} %)')
    print('%-50s %-50s %8s %12s %10s' % ('after', 'before', 'count', 'idle ms', 'mean us'))
    for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%-50s %-50s %8d %12.3f %10.2f' % (a, b, c, g / 1e6, g / c / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:])
