#!/bin/bash
# tools/profile_as_gaps.sh NAME [max_iter]: dense ActiveSet at BASELINE config 2's shape (n = 20 000, d = 64) under a rocprofv3 kernel
# trace; per-kernel summary and the idle time of the stream per kernel -> kernel transition (tools/trace_gaps.py)
set -o pipefail
out=gpurun_out/$1; iters=${2:-3000}; mkdir -p "$out"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 tools/bench_extra.py fit --n 20000 --d 64 --solver as --max-iter "$iters" > "$out/fit.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"
python3 tools/trace_gaps.py "$db" 300 > "$out/gaps.txt"
rm -rf "$out/trace"
cat "$out/fit.json"; head -12 "$out/kernel_stats.csv"; head -30 "$out/gaps.txt"
