#!/bin/bash
# A/B of two environment settings, interleaved (box-to-box and run-to-run noise is ~3 %)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in "$A" "$B"; do
  env $v python bench.py --steps 300 --no-extras --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_us_per_step'])"
done; done
