#!/bin/bash
# scenes per step x streams sweep of the default bench (no CPU baseline, no exact pass)
for cfg in "4 3" "4 4" "6 3" "8 3" "8 2" "2 6" "3 4" "4 2"; do
  set -- $cfg
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exact --no-timer --batch $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 streams $2: %.1f scenes/s' % d['value'])"
done
