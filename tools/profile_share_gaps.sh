#!/bin/bash
# tools/profile_share_gaps.sh NAME [G] [extra bench args]: the headline workload's 1/G shares (bench.py --emulate-shares G) under a
# rocprofv3 kernel trace: per-kernel summary and the stream's idle time per kernel -> kernel transition (tools/trace_gaps.py).
set -o pipefail
out=gpurun_out/$1; G=${2:-8}; shift; shift; mkdir -p "$out"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 bench.py --emulate-shares "$G" --steps 30 --warmup 3 "$@" > "$out/shares.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"
python3 tools/trace_gaps.py "$db" 0 > "$out/gaps.txt"
rm -rf "$out/trace"
head -14 "$out/kernel_stats.csv"; head -24 "$out/gaps.txt"
