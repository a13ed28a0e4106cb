#!/bin/bash
# Matrix-pipe counters of the streamed product for two builds of the library (before / after), on the GPU box:
#   tools/profile_stream.sh NAME LIB_BEFORE.so
# -> gpurun_out/NAME/{before,after}_pmc.txt (per-kernel counter means, tools/pmc_table.py) and the bench lines.
# Counter passes carry --kernel-trace only; the program after `--` is python3 itself.
set -o pipefail
name=$1; before=$2
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
args=(--storage stream --steps 6 --warmup 1 --no-cpu --kkt none)
pass1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
pass2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for which in before after; do
    if [ $which = before ]; then prog=(python3 tools/bench_with_lib.py "$before"); else prog=(python3 bench.py); fi
    "${prog[@]}" "${args[@]}" > "$out/${which}_bench.json" 2> "$out/${which}_bench.err" || { tail -5 "$out/${which}_bench.err"; exit 1; }
    rocprofv3 --pmc $pass1 --kernel-trace --output-format csv -d "$out/${which}_p1" -- "${prog[@]}" "${args[@]}" > /dev/null 2> "$out/${which}_p1.err" || { tail -5 "$out/${which}_p1.err"; exit 1; }
    rocprofv3 --pmc $pass2 --kernel-trace --output-format csv -d "$out/${which}_p2" -- "${prog[@]}" "${args[@]}" > /dev/null 2> "$out/${which}_p2.err" || { tail -5 "$out/${which}_p2.err"; exit 1; }
    python3 tools/pmc_table.py $(find "$out/${which}_p1" "$out/${which}_p2" -name '*counter_collection.csv') > "$out/${which}_pmc_all.txt"
        awk '/^[^ ]/{keep = ($0 ~ /gram_stream/)} keep' "$out/${which}_pmc_all.txt" > "$out/${which}_pmc.txt"
    cat "$out/${which}_pmc.txt"
    rm -rf "$out/${which}_p1" "$out/${which}_p2"
done
