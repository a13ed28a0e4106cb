export TMPDIR=/tmp
for sg in 4 8 0; do
  for n in 20000 30000 40000 50000 60000; do
    BQ_TEST_HOOKS=rows_per_step=$sg python bench.py --samples $n --features 64 --steps 200 --warmup 20 --no-cpu --kkt none 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows_per_step=$sg n=$n', '%.1f it/s'%b['value'], 'symv %.4f ms frac %.3f'%(b['roofline']['avg_launch_ms'], b['roofline']['frac']))"
  done
done
