#!/bin/bash
# One named step of a GPU session: output to gpurun_out/<dir>/<name>.log, a one-line verdict on stdout, the step's exit code
# returned (join steps with && so that nothing runs after a failure or a timeout).
#   [SOFT=1] tools/gpu_step.sh DIR NAME TIMEOUT_S command...
dir=gpurun_out/$1; name=$2; limit=$3; shift 3
mkdir -p "$dir"
start=$(date +%s)
timeout -k 10 "$limit" "$@" > "$dir/$name.log" 2> "$dir/$name.err"
rc=$?
echo "[step] $name rc=$rc $(( $(date +%s) - start ))s"
if [ $rc -ne 0 ]; then tail -n 25 "$dir/$name.log"; tail -n 25 "$dir/$name.err"; fi
# SOFT=1: an ordinary failure (a failing assertion) does not stop the chain; a timeout or a kill (124 / 137 / signals) always does
if [ -n "$SOFT" ] && [ $rc -ne 124 ] && [ $rc -lt 128 ]; then exit 0; fi
exit $rc
