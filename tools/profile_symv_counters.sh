#!/bin/bash
# Counters behind the launch-time spread of symv_tiles_kernel, collected on the GPU box:
#   tools/profile_symv_counters.sh NAME
# One process per pass (PMC passes carry --kernel-trace only): the panel product over the whole n = 100 000 panel (G = 1) and
# over a half share (G = 2), plus the bare streaming-read probe on 4 and 40 GiB, so that the kernel and the probe are compared
# under the same conditions.  Raw per-dispatch CSVs are kept (small: few launches) under gpurun_out/NAME/.
set -o pipefail
name=$1
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
args=(--emulate-shares 1,2 --steps 12 --warmup 2 --probe-gib 4,40)
pass() {   # pass TAG counters...
    tag=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$tag" -- python3 bench.py "${args[@]}" > "$out/$tag.json" 2> "$out/$tag.err"
    rc=$?
    f=$(find "$out/$tag" -name '*counter_collection.csv' | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "[pmc] $tag failed rc=$rc"; tail -n 4 "$out/$tag.err"; return 0; fi
    cp "$f" "$out/$tag.csv"; rm -rf "$out/$tag"
    echo "[pmc] $tag ok"
}
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
pass tlb2 TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
pass chan TCC_EA0_RDREQ
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 tools/pmc_kernel_table.py "$out"/*.csv > "$out/table.txt"
cat "$out/table.txt"
