#!/bin/bash
O=gpurun_out; mkdir -p $O
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|Power|fclk|socclk|Temperature \(Sensor (edge|junction|memory)" | head -12
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-exact --no-timer > $O/r4_clk_bench.json 2> $O/r4_clk_bench.err &
BP=$!
sleep 14
for i in 1 2 3 4; do echo "-- sample $i"; rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use|fclk" | head -8; sleep 1.5; done
wait $BP
python -c "
import json;d=json.load(open('gpurun_out/r4_clk_bench.json'));print('bench (400 steps): %.1f scenes/s'%d['value'])"
