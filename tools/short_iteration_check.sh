#!/bin/bash
# tools/short_iteration_check.sh OUT: the fixed cost of a short iteration — BASELINE config 2 three times (ms per step, the product's own
# time, f after 420 iterations: the bits), the headline's f and residual after 25 iterations (the bits again), the eight 1/8 shares
out=${1:-gpurun_out/short}; mkdir -p "$out"
for i in 1 2 3; do python3 bench.py --config c2 --steps 400 --warmup 20 --no-cpu --kkt none --line full 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 %.1f iter/s  %.4f ms per step  product %.4f ms  beside it %.1f us  f_last %r' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], 1e3 * (r['ms_per_step'] - r['roofline']['avg_launch_ms']), r['f_last']))"; done | tee "$out/summary.txt"
python3 bench.py --steps 20 --warmup 5 --no-cpu --kkt none --records none --line full 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.2f iter/s  %.4f ms per step  product %.4f ms  f_last %r  resid %r' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['f_last'], r['kkt_resid_last']))" | tee -a "$out/summary.txt"
python3 bench.py --emulate-shares 8 --steps 30 --warmup 3 > "$out/shares8.json" 2>/dev/null
python3 -c "
import json
r=json.loads(open('$out/shares8.json').read().strip().splitlines()[-1])
p=r['partitions'][0]
print('1/8 shares: ms per step', [round(v['ms_per_step'],4) for v in p['shares']], ' product ms', [round(v['symv_tiles_ms'],4) for v in p['shares']], ' beside it (us)', [round(v['fixed_cost_ms']*1e3,1) for v in p['shares']])" | tee -a "$out/summary.txt"
