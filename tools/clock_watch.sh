#!/bin/bash
# samples the GPU's clocks and power while bench.py runs (is the overlapped frame rate limited by the power / clock governor?)
python bench.py --no-c3 --no-latency --no-cpu-baseline --no-stage-events --steps 4000 --windows 5 "$@" > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' ' ; echo
  sleep 0.4
done
wait $BP
python3 -c "import json; d=json.load(open('gpurun_out/clock_bench.json')); print('ms/frame', d['ms_per_step'], d['windows_ms_per_step'])"
