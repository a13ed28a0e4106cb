#!/bin/bash
# tools/profile_placement.sh NAME: tools/placement_counters.py plain (launch time per panel) and under six PMC passes; the
# per-panel table -> gpurun_out/NAME/placement_table.txt.  PMC passes carry --kernel-trace only.
set -o pipefail
out=gpurun_out/$1; mkdir -p "$out"; export TMPDIR=/tmp
K=5; R=4
python3 tools/placement_counters.py $K $R > "$out/plain.json" 2> "$out/plain.err" || { tail -5 "$out/plain.err"; exit 1; }
cat "$out/plain.json"
pass() {
    tag=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$tag" -- python3 tools/placement_counters.py $K $R > "$out/$tag.json" 2> "$out/$tag.err"
    rc=$?
    f=$(find "$out/$tag" -name '*counter_collection.csv' | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "[pmc] $tag failed rc=$rc"; tail -n 4 "$out/$tag.err"; return 0; fi
    cp "$f" "$out/$tag.csv"; rm -rf "$out/$tag"; echo "[pmc] $tag ok: $(cat $out/$tag.json | head -c 300)"
}
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
pass tlb2 TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 tools/placement_counters_table.py $K $((R+1)) "$out"/tlb.csv "$out"/tlb2.csv "$out"/ea.csv "$out"/l2.csv "$out"/lat.csv > "$out/placement_table.txt"
cat "$out/placement_table.txt"
