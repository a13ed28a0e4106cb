#!/bin/bash
# tools/as_sweep_block.sh OUT N D blocks...: dense ActiveSet (SVC hinge RBF) to 'optimal' with the big block of the kept factor's sweeps
# forced (BQ_TEST_HOOKS=sweep_block=B: a sweep is two launches per block and direction; larger blocks = fewer launches, more
# inverse-block traffic)
out=gpurun_out/$1; n=$2; d=$3; shift 3; mkdir -p "$out"
for B in "$@"; do
  export BQ_TEST_HOOKS=sweep_block=$B
  python3 bench.py --solver as --samples $n --features $d --no-cpu > "$out/as_n${n}_B$B.json" 2> "$out/as_n${n}_B$B.err" || { tail -3 "$out/as_n${n}_B$B.err"; exit 1; }
  python3 -c "
import json; r=json.loads(open('$out/as_n${n}_B$B.json').read().strip().splitlines()[-1])
print('n = $n, sweep block $B: %.2f s, %d iterations, %s, f = %.10g, %d base factorisations x %.1f ms' % (r['value'], r['iterations'], r['status'], r['f'], r['counters']['base_factorisations'], r['roofline']['avg_factor_ms']))"
done
