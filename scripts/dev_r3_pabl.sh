#!/bin/bash
# round-3 dev (DIAG build): timing-only ablations of the persistent brick kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
for a in 0 1 8 9 32 2 34 16 50 4; do
  PCCM_BRICK_NT=512 PCCM_P_ABLATE=$a timeout -k 10 200 python bench.py --steps 60 --no-graph --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('abl=$a ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
