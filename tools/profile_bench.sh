#!/bin/bash
# Evidence behind bench.py's roofline record, collected on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh NAME [bench.py args...]
# 1. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/NAME/trace/  (+ NAME_kernel_stats.csv, the bench line)
# 2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, kernel-trace only (MI355X_MICROARCH.md, HBM section)
#    -> gpurun_out/NAME/pmc_traffic.json via tools/pmc_summary.py (FETCH_SIZE doubled per the gfx950 note)
# The program after `--` is python3 itself (no env/bash hop: the profiler has initialised the GPU by then).
set -o pipefail
name=$1; shift
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@" --no-cpu --kkt none --records none)
rocprofv3 --kernel-trace --stats -d "$out/trace" -- python3 bench.py "${args[@]}" > "$out/bench_under_rocprof.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
db=$(find "$out/trace" -name '*_results.db' | head -1)
if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats.csv"; else cp $(find "$out/trace" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats.csv"; fi
head -6 "$out/kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py "${args[@]}" > "$out/bench_pmc_fetch.json" 2> "$out/pmc_fetch.err" || { tail -5 "$out/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py "${args[@]}" > "$out/bench_pmc_write.json" 2> "$out/pmc_write.err" || { tail -5 "$out/pmc_write.err"; exit 1; }
f=$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1)
w=$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)
wl=$(python3 -c "import json,sys; print(json.loads(open('$out/bench_under_rocprof.json').read().strip().splitlines()[-1])['config']['workload'])")
python3 tools/pmc_summary.py "$f" "$w" "$out/pmc_traffic.json" workload=$wl n_gpus=1 command="bench.py ${args[*]}" | head -12
# the raw per-dispatch CSVs are large: keep the summaries only
rm -rf "$out/pmc_fetch" "$out/pmc_write" "$out/trace"
