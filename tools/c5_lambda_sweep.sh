#!/bin/bash
# tools/c5_lambda_sweep.sh OUT scale...: BASELINE config 5 with the Chebyshev interval of the order-2 remainder [1, 1 + scale * lambda_power]
# (build-time macro BQ_PC2_LAMBDA_SCALE of bq_as_pc.hip; the library is rebuilt on the box for every value and restored at the end)
out=$1; shift
mkdir -p "$out"
for sc in "$@"; do
    touch optiml_amd/csrc/bq_as_pc.hip
    BQ_EXTRA_CXXFLAGS=-DBQ_PC2_LAMBDA_SCALE=$sc python3 -m optiml_amd.build > "$out/build_$sc.log" 2>&1 || { tail -3 "$out/build_$sc.log"; exit 1; }
    python3 bench.py --config c5 --steps 20 --warmup 2 --no-cpu --kkt none --line full > "$out/c5_scale_$sc.json" 2> "$out/c5_scale_$sc.err" || exit 1
    python3 -c "
import json,sys
r=json.loads(open('$out/c5_scale_$sc.json').read().strip().splitlines()[-1])
print('lambda scale $sc: %.3f outer it/s, %.1f ms per outer iteration, %.2f products per outer iteration, product %.2f ms' % (r['value'], r['ms_per_step'], r['inner_products_per_step'], r['roofline']['avg_launch_ms']))"
done
touch optiml_amd/csrc/bq_as_pc.hip; python3 -m optiml_amd.build > /dev/null 2>&1
