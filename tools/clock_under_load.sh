#!/bin/bash
# Samples the GPU's shader clock and power (rocm-smi) while the headline verification runs in a loop.
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python -u bench.py --lean --value-basis resident --steps 600 --warmup 5 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 40
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|power|fclk|mclk" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('ms_per_step', d['ms_per_step'])"
