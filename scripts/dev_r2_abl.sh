#!/bin/bash
# round-2 dev: what LDS bank conflicts cost the brick kernel (timing-only ablations: 4 = no epilogue, 68 = 4 + broadcast reads)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
for a in 0 4 68 4 68; do
  PCCM_BRICK_ABLATE=$a timeout -k 10 300 python bench.py --steps 100 --no-graph --no-extras --no-cpu-baseline > $O/b_$a.json 2> $O/b_$a.err; python -c "
import json; d=json.load(open('$O/b_$a.json')); print('ablate=$a ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
