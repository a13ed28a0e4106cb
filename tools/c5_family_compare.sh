#!/bin/bash
# tools/c5_family_compare.sh OUTDIR [STEPS]: BASELINE config 5 (ActiveSetCG, n = 250 000, d = 256, fp32 panel) with the preconditioner's
# second feature family switched: 1 = class-mean cross term (rounds 3-4), 2 = projected order-2 directions (round 5 default).
out=${1:-gpurun_out/c5fam}; steps=${2:-20}
mkdir -p "$out"
for fam in ${FAMILIES:-1 2 3}; do
    BQ_TEST_HOOKS=as_cg_pc_class=$fam python3 bench.py --config c5 --steps "$steps" --warmup 2 --no-cpu --kkt none --line full > "$out/c5_fam$fam.json" 2> "$out/c5_fam$fam.err" || exit 1
    python3 - "$out/c5_fam$fam.json" "$fam" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"family {sys.argv[2]}: {r['value']:.3f} outer it/s, {r['ms_per_step']:.1f} ms per outer iteration, {r['inner_products_per_step']:.2f} products per outer iteration, "
      f"product {r['roofline']['avg_launch_ms']:.2f} ms ({r['roofline']['frac']:.3f} of 8 TB/s), f_last {r['f_last']:.6f}")
PY
done
