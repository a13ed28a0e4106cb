#!/bin/bash
# Gram build with the plain 2-D grid (BQ_GRAM_ORDER=0) against the XCD-local rectangle order (default): launch time from the
# kernel trace and L2 fill traffic (FETCH_SIZE) from a PMC pass, n = 100 000 d = 128 fp64.   tools/profile_gram_order.sh NAME
set -o pipefail
out=gpurun_out/$1
mkdir -p "$out"
export TMPDIR=/tmp
for order in 0 1; do
  export BQ_GRAM_ORDER=$order
  rocprofv3 --kernel-trace --stats -d "$out/trace$order" -- python3 bench.py --steps 3 --warmup 1 --no-cpu --kkt none > "$out/bench_order$order.json" 2> "$out/trace$order.err" || { tail -n 5 "$out/trace$order.err"; exit 1; }
  db=$(find "$out/trace$order" -name '*_results.db' | head -1)
  python3 tools/rocpd_stats.py "$db" | grep -i "gram_mfma\|Name" > "$out/gram_order${order}_kernel_stats.csv"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc$order" -- python3 bench.py --steps 3 --warmup 1 --no-cpu --kkt none > /dev/null 2> "$out/pmc$order.err" || { tail -n 5 "$out/pmc$order.err"; exit 1; }
  f=$(find "$out/pmc$order" -name '*counter_collection.csv' | head -1)
  python3 - "$f" $order >> "$out/summary.txt" <<'PY'
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'gram_mfma' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print(f'BQ_GRAM_ORDER={sys.argv[2]}: gram_mfma_kernel FETCH_SIZE {sum(v) / len(v) * 1024 * 2 / 1e9:.2f} GB per launch (x2 gfx950 correction), {len(v)} launch(es)')
PY
  cat "$out/gram_order${order}_kernel_stats.csv" >> "$out/summary.txt"
  rm -rf "$out/trace$order" "$out/pmc$order"
done
cat "$out/summary.txt"
