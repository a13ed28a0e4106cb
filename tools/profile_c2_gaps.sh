#!/bin/bash
# tools/profile_c2_gaps.sh NAME: BASELINE config 2 (n = 20 000 PG) under a rocprofv3 kernel trace: per-kernel summary and the stream's idle
# time per kernel -> kernel transition (tools/trace_gaps.py) — what the ~25 us beside the product of a short iteration are made of.
set -o pipefail
out=gpurun_out/$1; mkdir -p "$out"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 bench.py --config c2 --steps 400 --warmup 20 --no-cpu --kkt none --line full > "$out/bench.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"
python3 tools/trace_gaps.py "$db" 0 > "$out/gaps.txt"
rm -rf "$out/trace"
head -12 "$out/kernel_stats.csv"; head -24 "$out/gaps.txt"
