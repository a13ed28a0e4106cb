#!/bin/bash
# tools/share_variant_sweep.sh NAME G v1 v2 ...: the headline workload's 1/G shares (bench.py --emulate-shares G) once per BQ_SYMV_VARIANT
# (<tiles per strip><rows per step>: 84 88 44 48 24 28); prints the symv_tiles ms of every share.  (Another strip length is another
# sum association: an experiment on the launch's tail, not a candidate default.)
out=gpurun_out/$1; G=$2; shift; shift; mkdir -p "$out"
for v in "$@"; do
    BQ_SYMV_VARIANT=$v python3 bench.py --emulate-shares "$G" --steps 30 --warmup 3 > "$out/shares_$v.json" 2> "$out/shares_$v.err"
    python3 - "$out/shares_$v.json" "$v" <<'PY' | tee -a "$out/sweep.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d['partitions'][0]
t = [s['symv_tiles_ms'] for s in p['shares']]
print('variant %s  G=%d  symv_tiles ms: %s  mean %.4f  slowest step %.4f ms' % (sys.argv[2], p['G'], ' '.join('%.3f' % x for x in t), sum(t) / len(t), p['slowest_share_ms_per_step']))
PY
done
