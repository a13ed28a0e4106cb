#!/bin/bash
out=gpurun_out/r6rr; mkdir -p $out
for i in 1 2 3 4 5 6; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --kkt none --records none > $out/run$i.json 2> $out/run$i.err || exit 1
  python3 -c "
import json; r=json.loads(open('$out/run$i.json').read().strip().splitlines()[-1])
print('run $i: %.1f iter/s, %.3f ms per step, symv_tiles %.3f ms = %.3f of 8 TB/s; placement candidates %s ms, frac_first_placement %.3f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['config']['panel_placement_ms'], r['roofline']['frac_first_placement']))"
done
