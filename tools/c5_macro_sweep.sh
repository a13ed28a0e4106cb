#!/bin/bash
# tools/c5_macro_sweep.sh OUT MACRO value...: BASELINE config 5 with bq_as_pc.hip rebuilt on the box with -DMACRO=value for every value
# (BQ_PC2_LAMBDA_SCALE: widening of the order-2 remainder's spectrum bound);
# the default library is restored at the end
out=$1; macro=$2; shift 2
mkdir -p "$out"
for v in "$@"; do
    touch optiml_amd/csrc/bq_as_pc.hip
    BQ_EXTRA_CXXFLAGS=-D$macro=$v python3 -m optiml_amd.build > "$out/build_$v.log" 2>&1 || { tail -3 "$out/build_$v.log"; exit 1; }
    python3 bench.py --config c5 --steps 20 --warmup 2 --no-cpu --kkt none --line full > "$out/c5_${macro}_$v.json" 2> "$out/c5_${macro}_$v.err" || exit 1
    python3 -c "
import json
r=json.loads(open('$out/c5_${macro}_$v.json').read().strip().splitlines()[-1])
print('$macro=$v: %.3f outer it/s, %.1f ms per outer iteration, %.2f products per outer iteration, product %.2f ms' % (r['value'], r['ms_per_step'], r['inner_products_per_step'], r['roofline']['avg_launch_ms']))"
done
touch optiml_amd/csrc/bq_as_pc.hip; python3 -m optiml_amd.build > /dev/null 2>&1
