for m in 0 256; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact --mode tile8=$m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile8 $m metric: %.1f scenes/s, latency one scene %.2f ms' % (d['value'], d['latency_ms_one_scene']))"
  python bench.py --workload cfg2 --steps 30 --warmup 5 --no-cpu-baseline --no-exact --mode tile8=$m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile8 $m cfg2: %.1f scenes/s, latency one scene %.3f ms' % (d['value'], d['latency_ms_one_scene']))"
done
