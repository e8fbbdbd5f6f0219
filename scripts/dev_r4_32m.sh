#!/bin/bash
# Round-4 developer scratch: how much the 32M + 32M step varies within one box (same process restarted), and with the hipGraph off.
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
  timeout -k 10 200 python3 bench.py --points 32000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/m32_$k.json 2> gpurun_out/m32_$k.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/m32_$k.json').read().strip().splitlines()[-1]); print('graph run $k', d['ms_per_step'], d['kernel_us_per_step'], d['roofline']['frac'])"
done
timeout -k 10 200 python3 bench.py --points 32000000 --steps 5 --warmup 2 --no-graph --no-extras --no-cpu-baseline > gpurun_out/m32_e.json 2> gpurun_out/m32_e.err
python3 -c "
import json; d=json.loads(open('gpurun_out/m32_e.json').read().strip().splitlines()[-1]); print('eager run', d['ms_per_step'], d['kernel_us_per_step'], d['roofline']['frac'])"
timeout -k 10 200 python3 bench.py --points 16000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/m16.json 2> gpurun_out/m16.err
python3 -c "
import json; d=json.loads(open('gpurun_out/m16.json').read().strip().splitlines()[-1]); print('16M', d['ms_per_step'], d['kernel_us_per_step'], d['roofline']['frac'])"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
