#!/bin/bash
# tools/as_kkt_sweep.sh NAME VAR v1 v2 ...: dense ActiveSet at BASELINE config 2's shape (n = 20 000, d = 64) to 'optimal' once per value of the
# environment variable VAR (e.g. BQ_TEST_HOOKS as_schur_min=64 as_schur_min=256), with BQ_AS_TIMING's host-side breakdown; one line per run in NAME/sweep.txt
out=gpurun_out/$1; var=$2; shift; shift; mkdir -p "$out"
for v in "$@"; do
    env "$var=$v" BQ_AS_TIMING=1 python3 bench.py --solver as --samples 20000 --features 64 --no-cpu --records none > "$out/kkt_$v.json" 2> "$out/kkt_$v.err"
    echo "== $var=$v" >> "$out/sweep.txt"
    grep BQ_AS_TIMING "$out/kkt_$v.err" >> "$out/sweep.txt"
    python3 -c "
import json,sys
d=json.loads(open('$out/kkt_$v.json').read().strip().splitlines()[-1])
print('   time to optimal %.3f s, %d iterations, status %s, f %.12g, factorisations %d' % (d['value'], d['iterations'], d['status'], d['f'], d['roofline']['factorisations']))" >> "$out/sweep.txt"
done
cat "$out/sweep.txt"
