#!/bin/bash
# tools/profile_variants.sh NAME "VAR=a VAR=b ..." KERNEL_REGEX [bench.py args...]: one rocprofv3 kernel trace of the bench command
# per environment setting; prints calls / average / max of the kernels that match, per setting -> gpurun_out/NAME/variants.txt
set -o pipefail
name=$1; settings=$2; regex=$3; shift 3
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
: > "$out/variants.txt"
i=0
for setting in $settings; do
  i=$((i+1))
  env_args=(${setting//,/ })
  ( export "${env_args[@]}"; rocprofv3 --kernel-trace --stats -d "$out/trace$i" -- python3 bench.py "$@" --no-cpu --kkt none --records none > "$out/bench_$i.json" 2> "$out/trace$i.err" ) || { tail -5 "$out/trace$i.err"; exit 1; }
  db=$(find "$out/trace$i" -name '*_results.db' | head -1)
  python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats_$i.csv"
  rm -rf "$out/trace$i"
  echo "== $setting  $(python3 -c "import json;r=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]);print('ms_per_step %.3f inner %.2f'%(r['ms_per_step'], r.get('inner_products_per_step') or 0))")" >> "$out/variants.txt"
  python3 - "$out/kernel_stats_$i.csv" "$regex" >> "$out/variants.txt" <<'PY'
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], row['Name']):
        print('   %-40s calls %5d avg %9.1f us  max %9.1f us  total %9.3f ms' % (row['Name'][:40], int(float(row['Calls'])), float(row['AverageNs']) / 1e3, float(row['MaxNs']) / 1e3, float(row['TotalDurationNs']) / 1e6))
PY
done
cat "$out/variants.txt"
