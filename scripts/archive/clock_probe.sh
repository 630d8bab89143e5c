#!/bin/bash
# kernel times of consecutive compiled C3 launches with the shader clock sampled beside them
cd "$(dirname "$0")/.."
( for k in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | tr '\n' ' '; echo; sleep 0.4; done ) > gpurun_out/clock_probe.txt &
P=$!
python scripts/launch_times.py 1e8
python scripts/launch_times.py 1e8
wait $P
cat gpurun_out/clock_probe.txt | cut -c1-200
rocm-smi --showperflevel --showmaxpower 2>/dev/null | grep -i "level\|max" | head -4
