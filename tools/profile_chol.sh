#!/bin/bash
# tools/profile_chol.sh NAME N: one blocked Cholesky of order N (tools/bench_extra.py chol) under a rocprofv3 kernel trace: per-kernel summary
# and the time line of the factorisation's launches (start offsets and durations), to see what the latency chain of a mid-size factorisation is
set -o pipefail
out=gpurun_out/$1; n=${2:-16384}; mkdir -p "$out"; export TMPDIR=/tmp
python3 tools/bench_extra.py chol --n "$n" > "$out/plain.json" 2>/dev/null
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 tools/bench_extra.py chol --n "$n" > "$out/traced.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"
python3 - "$db" > "$out/timeline.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
st, en = ('start', 'end') if 'start' in cols else ('start_timestamp', 'end_timestamp')
rows = list(cur.execute(f'select name, {st}, {en} from kernels order by {st}'))
# the big factorisation = the last run of potrf/trsm/syrk kernels
names = ('potrf_diag128', 'trsm_gemm', 'syrk_head', 'syrk_col', 'syrk_kernel', 'pivot_thr')
sel = [r for r in rows if any(k in r[0] for k in names)]
# split at gaps > 5 ms: keep the last group (the order-N factorisation; the first is the 256-order warm-up)
groups, cur_g = [], [sel[0]]
for a, b in zip(sel, sel[1:]):
    if b[1] - a[2] > 5e6:
        groups.append(cur_g); cur_g = []
    cur_g.append(b)
groups.append(cur_g)
g = max(groups, key=len)
t0 = g[0][1]
print('launches %d  span %.3f ms' % (len(g), (max(r[2] for r in g) - t0) / 1e6))
busy = {}
for nme, s, e in g:
    k = nme.split('(')[0]
    busy[k] = busy.get(k, 0) + (e - s)
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print('%-28s %9.3f ms' % (k, v / 1e6))
print('first 60 launches: name, start offset us, duration us')
for nme, s, e in g[:60]:
    print('%-24s %10.1f %9.1f' % (nme.split('(')[0], (s - t0) / 1e3, (e - s) / 1e3))
print('last 40 launches')
for nme, s, e in g[-40:]:
    print('%-24s %10.1f %9.1f' % (nme.split('(')[0], (s - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf "$out/trace"
cat "$out/plain.json"; head -40 "$out/timeline.txt"
