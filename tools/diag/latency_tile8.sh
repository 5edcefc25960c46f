for m in 0 200 700; do
 for g in 0 1; do
  python bench.py --batch 1 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-exact --no-timer --graph $g --mode tile8=$m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile8 $m graph $g: batch1 1 stream %.2f ms/step = %.1f scenes/s' % (d['ms_per_step'], d['value']))"
 done
done
