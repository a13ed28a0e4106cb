#!/bin/bash
# tools/profile_smo_gaps.sh NAME [n]: SVC.fit(optimizer='smo') at n samples under a rocprofv3 kernel trace: kernel summary + stream idle per transition
set -o pipefail
out=gpurun_out/$1; n=${2:-100000}; mkdir -p "$out"; export TMPDIR=/tmp
python3 tools/bench_extra.py smo --n "$n" --d 128 > "$out/plain.json" 2>/dev/null
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 tools/bench_extra.py smo --n "$n" --d 128 > "$out/traced.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"
python3 tools/trace_gaps.py "$db" 0 > "$out/gaps.txt"
rm -rf "$out/trace"
cat "$out/plain.json"; head -8 "$out/kernel_stats.csv" | cut -c1-180; head -12 "$out/gaps.txt"
