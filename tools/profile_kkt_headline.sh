#!/bin/bash
# Time-to-KKT at the HEADLINE size (n = 100 000, d = 128) by the two factorising solvers of the north star — too long for a bench
# line, so measured here and committed as profiles/rNN/time_to_kkt_headline.json (bench.py quotes the file in its complete record):
#   tools/profile_kkt_headline.sh OUTDIR ip|as [WALL_LIMIT_S]
# InteriorPoint (interior_point.py:95-281) runs to its stop test (~6 min).  Dense ActiveSet (active_set.py:82-237) needs ~n outer
# iterations: it runs to 'optimal' or to the wall limit, whichever comes first, and a run that was cut says how far it got.
set -o pipefail
out=gpurun_out/$1; which=$2; limit=${3:-1000}
mkdir -p "$out"
python3 bench.py --solver "$which" --samples 100000 --features 128 --no-cpu --wall-limit "$limit" > "$out/kkt_${which}_n100000.json" 2> "$out/kkt_${which}_n100000.err"
rc=$?
tail -n 3 "$out/kkt_${which}_n100000.err"
python3 - "$out" <<'PY'
import json, os, sys
d = sys.argv[1]
rec = {}
path = os.path.join(d, 'time_to_kkt_headline.json')
if os.path.exists(path):
    rec = json.load(open(path))
for which, key in (('ip', 'ip_headline'), ('as', 'as_headline')):
    p = os.path.join(d, f'kkt_{which}_n100000.json')
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.strip().startswith('{')]
        if lines:
            r = json.loads(lines[-1])
            keep = ('value', 'unit', 'iterations', 'status', 'f', 'n_sv', 's_per_iteration', 'stop_test', 'includes', 'route', 'roofline', 'config',
                    'wall_limit_s', 'cut_at_wall_limit', 'bounds_at_the_end', 'counters')
            rec[key] = {k: r[k] for k in keep if k in r}
json.dump(rec, open(path, 'w'), indent=1)
print(json.dumps({k: {q: v.get(q) for q in ('value', 'iterations', 'status')} for k, v in rec.items()}))
PY
exit $rc
