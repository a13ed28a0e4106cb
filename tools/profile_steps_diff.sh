#!/bin/bash
# tools/profile_steps_diff.sh NAME STEPS_A STEPS_B PRODUCT_KERNEL_PREFIX [bench.py args...]: two rocprofv3 kernel traces of the same
# bench command at two step counts -> gpurun_out/NAME/{kernel_stats_A.csv,kernel_stats_B.csv,per_step.csv} (tools/trace_diff.py)
set -o pipefail
name=$1; sa=$2; sb=$3; prefix=$4; shift 4
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
for tag in A B; do
  steps=$sa; [ $tag = B ] && steps=$sb
  rocprofv3 --kernel-trace --stats -d "$out/trace$tag" -- python3 bench.py "$@" --steps "$steps" --no-cpu --kkt none --records none > "$out/bench_$tag.json" 2> "$out/trace$tag.err" || { tail -5 "$out/trace$tag.err"; exit 1; }
  db=$(find "$out/trace$tag" -name '*_results.db' | head -1)
  if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" > "$out/kernel_stats_$tag.csv"; else cp $(find "$out/trace$tag" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats_$tag.csv"; fi
  rm -rf "$out/trace$tag"
done
python3 tools/trace_diff.py "$out/kernel_stats_A.csv" "$sa" "$out/kernel_stats_B.csv" "$sb" "$prefix" > "$out/per_step.csv"
cat "$out/per_step.csv"
