#!/bin/bash
# usage: as_limit_sweep.sh OUT N D limits...
out=gpurun_out/$1; n=$2; d=$3; shift 3; mkdir -p "$out"
for L in "$@"; do
  if [ "$L" = default ]; then unset BQ_TEST_HOOKS; else export BQ_TEST_HOOKS=as_schur_limit=$L; fi
  python3 bench.py --solver as --samples $n --features $d --no-cpu > "$out/as_n${n}_L$L.json" 2> "$out/as_n${n}_L$L.err" || { tail -3 "$out/as_n${n}_L$L.err"; exit 1; }
  python3 -c "
import json; r=json.loads(open('$out/as_n${n}_L$L.json').read().strip().splitlines()[-1])
print('limit $L: %.2f s, %d iterations, %s, %d base factorisations x %.1f ms (%.1f %% of the wall)' % (r['value'], r['iterations'], r['status'], r['counters']['base_factorisations'], r['roofline']['avg_factor_ms'], 100*r['roofline']['factor_share_of_wall']))"
done
