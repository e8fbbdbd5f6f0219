#!/bin/bash
# round-3 dev (DIAG build): in-kernel phase stamps + ablations of the brick kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
PCCM_BRICK_STAMP=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-graph --no-extras --no-cpu-baseline > $O/stamp.json 2> $O/stamp.err
tail -2 $O/stamp.err
for a in 0 4 7 16 32; do
  PCCM_BRICK_ABLATE=$a timeout -k 10 300 python bench.py --steps 100 --no-graph --no-extras --no-cpu-baseline > $O/b_$a.json 2> $O/b_$a.err; python -c "
import json; d=json.load(open('$O/b_$a.json')); print('ablate=$a ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
