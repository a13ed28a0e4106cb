#!/bin/bash
# tools/profile_gram_variants.sh NAME: the streamed product (gram_stream_sym_kernel, n = 100 000, d = 128) with the product library
# and with the measured epilogue variants of tools/build_variant.py (build/variants/*/libbcqp_hip.so: fold1 = argument of the
# exponential as one fma + one add + one min; cheapexp = a 2-instruction stand-in for exp (wrong values: the ceiling of ANY cheaper
# map); fold1_cheapexp = both) — kernel trace per library, and the SQ counters of the product library and of fold1.
# -> gpurun_out/NAME/gram_epilogue_variants.txt   (measure, then close)
set -o pipefail
out=gpurun_out/$1; mkdir -p "$out"; export TMPDIR=/tmp
args=(--storage stream --steps 6 --warmup 1 --no-cpu --kkt none --records none)
sum=$out/gram_epilogue_variants.txt
: > "$sum"
for v in product fold1 cheapexp fold1_cheapexp; do
    if [ $v = product ]; then prog=(python3 bench.py); else prog=(python3 tools/bench_with_lib.py build/variants/$v/libbcqp_hip.so); fi
    rocprofv3 --kernel-trace --stats -d "$out/t_$v" -- "${prog[@]}" "${args[@]}" > "$out/bench_$v.json" 2> "$out/t_$v.err" || { tail -5 "$out/t_$v.err"; exit 1; }
    db=$(find "$out/t_$v" -name '*_results.db' | head -1)
    python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats_$v.csv"; rm -rf "$out/t_$v"
    python3 - "$v" "$out/kernel_stats_$v.csv" "$out/bench_$v.json" >> "$sum" <<'PY'
import csv, json, sys
v, path, bench = sys.argv[1:]
r = json.loads(open(bench).read().strip().splitlines()[-1])
for row in csv.DictReader(open(path)):
    if 'gram_stream_sym_kernel' in row['Name']:
        print('%-16s gram_stream_sym_kernel: %3d launches, avg %8.3f ms, min %8.3f ms   (bench: %.3f ms per step, frac %.3f of the fp64 MFMA peak)'
              % (v, int(float(row['Calls'])), float(row['AverageNs']) / 1e6, float(row['MinNs']) / 1e6, r['ms_per_step'], r['roofline']['frac']))
PY
done
for v in product fold1; do
    if [ $v = product ]; then prog=(python3 bench.py); else prog=(python3 tools/bench_with_lib.py build/variants/$v/libbcqp_hip.so); fi
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$out/p_$v" -- "${prog[@]}" "${args[@]}" > /dev/null 2> "$out/p_$v.err" || { tail -5 "$out/p_$v.err"; exit 1; }
    echo "== counters, $v (mean per launch of gram_stream_sym_kernel)" >> "$sum"
    python3 tools/pmc_table.py $(find "$out/p_$v" -name '*counter_collection.csv') | awk '/^[^ ]/{keep = ($0 ~ /gram_stream/)} keep' >> "$sum"
    rm -rf "$out/p_$v"
done
cat "$sum"
