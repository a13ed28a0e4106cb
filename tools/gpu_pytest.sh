#!/bin/bash
# usage: tools/gpu_pytest.sh OUTDIR [pytest args...] — runs the GPU suite with the log under gpurun_out/, exits non-zero on failure
out=gpurun_out/$1; shift
mkdir -p "$out"
python -m pytest "$@" > "$out/pytest.log" 2>&1
rc=$?
tail -25 "$out/pytest.log"
if grep -q "Memory access fault" "$out/pytest.log"; then rc=99; fi
exit $rc
