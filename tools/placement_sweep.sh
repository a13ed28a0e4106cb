#!/bin/bash
# tools/placement_sweep.sh NAME: tools/placement_probe.py (5 panels of the headline problem held at once, timed, released,
# rebuilt) in fresh processes, with and without BQ_PANEL_CONTIGUOUS — and once after a warm-up process has churned the device's
# memory — is the spread of the launch time a property of the physical contiguity the driver hands out?
out=gpurun_out/$1; mkdir -p "$out"; : > "$out/placement_sweep.txt"
for rep in 1 2 3; do
  for c in 0 1; do
    echo "== process $rep BQ_PANEL_CONTIGUOUS=$c" >> "$out/placement_sweep.txt"
    BQ_PANEL_CONTIGUOUS=$c timeout -k 10 120 python tools/placement_probe.py 5 10 >> "$out/placement_sweep.txt" 2>> "$out/placement_sweep.err" || echo "   (failed)" >> "$out/placement_sweep.txt"
  done
done
cat "$out/placement_sweep.txt"
