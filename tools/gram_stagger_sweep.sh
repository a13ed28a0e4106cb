#!/bin/bash
# tools/gram_stagger_sweep.sh NAME: the Gram build at n = 100 000, d = 128 (bench.py --steps 2: gram_build_s from HIP events) for
# the start-up phase shifts of the first round of workgroups (BQ_GRAM_STAGGER mode / BQ_GRAM_STAGGER_UNIT x 2 us)
out=gpurun_out/$1; mkdir -p "$out"; : > "$out/gram_stagger.txt"
for cfg in "0 0" "1 5" "1 10" "1 15" "2 5" "2 10" "2 15" "3 10" "4 3" "4 5" "4 8"; do
  set -- $cfg
  for rep in 1 2; do
    BQ_GRAM_STAGGER=$1 BQ_GRAM_STAGGER_UNIT=$2 python bench.py --samples 100000 --features 128 --steps 2 --warmup 1 --no-cpu --kkt none --no-placement 2>/dev/null \
      | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $1 unit $2 rep $rep: gram_build %.3f ms' % (1e3*r['gram_build_s']))" >> "$out/gram_stagger.txt"
  done
done
cat "$out/gram_stagger.txt"
