#!/bin/bash
# shipped / v2xreal shapes are launch-latency-bound (64-256 workgroups per launch): how many independent scene batches should be in flight?
for wl in shipped v2xreal; do
  for cfg in "4 3" "4 4" "4 6" "4 8" "8 4" "8 6" "2 8"; do
    set -- $cfg
    python bench.py --workload $wl --steps 60 --warmup 6 --no-cpu-baseline --no-exact --no-timer --batch $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl batch $1 streams $2: %.1f scenes/s' % d['value'])"
  done
done
