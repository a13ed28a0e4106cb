#!/bin/bash
# tools/profile_gram_counters.sh NAME: SQ counter passes of the Gram build (gram_mfma_kernel) and of the streamed product
# (gram_stream_sym_kernel) at n = 100 000, d = 128 -> gpurun_out/NAME/gram_counters.txt (per kernel: mean per launch).
# What still blocks the matrix pipe?  PMC passes carry --kernel-trace only.
set -o pipefail
out=gpurun_out/$1; mkdir -p "$out"; export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_available.txt" 2>&1 || true
grep -o "SQ_[A-Z0-9_]*" "$out/counters_available.txt" | sort -u > "$out/sq_counter_names.txt"
pass() {
    tag=$1; mode=$2; shift 2
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$tag" -- python3 bench.py --samples 100000 --features 128 --storage $mode --steps 3 --warmup 1 --no-cpu --kkt none --no-placement --records none > "$out/$tag.json" 2> "$out/$tag.err"
    rc=$?
    f=$(find "$out/$tag" -name '*counter_collection.csv' | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "[pmc] $tag failed rc=$rc"; tail -n 3 "$out/$tag.err"; return 0; fi
    cp "$f" "$out/$tag.csv"; rm -rf "$out/$tag"; echo "[pmc] $tag ok"
}
for mode in f64 stream; do
  pass a_$mode $mode SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass b_$mode $mode SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64
  pass c_$mode $mode SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE
done
python3 - "$out" > "$out/gram_counters.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
tab = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(sys.argv[1] + '/[abc]_*.csv')):
    for r in csv.DictReader(open(path, newline='')):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'gram_' not in k:
            continue
        tab[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        tab[k[:40]]['_ms'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
for k in sorted(tab):
    print(k)
    for c in sorted(tab[k]):
        v = tab[k][c]
        print('    %-36s mean %18.3f  n=%d' % (c, sum(v) / len(v), len(v)))
PY
cat "$out/gram_counters.txt"
