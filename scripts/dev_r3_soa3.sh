#!/bin/bash
# round-3 dev: occupancy experiments with the 47-register SoA brick kernel (5 workgroups per CU / a ninth wave for leftovers)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3soa; mkdir -p $O
export TMPDIR=/tmp
for e in "PCCM_X=0" "PCCM_BRICK_SLOTS=40" "PCCM_BRICK_SLOTS=40 PCCM_BRICK_PL=1728" "PCCM_BRICK_PL=1728"; do
  env $e timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$e', 'ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
