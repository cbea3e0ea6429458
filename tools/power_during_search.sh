#!/bin/bash
# Run on the GPU box: samples rocm-smi (power, clocks) once a second while the benchmark's search runs for ~20 s.
timeout -k 10 120 python3 bench.py --steps 600 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
sleep 10
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "\(W\)|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
cut -c1-160 gpurun_out/power_bench.json
rocm-smi --showmaxpower 2>/dev/null | grep "(W)"
