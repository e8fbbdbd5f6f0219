#!/bin/bash
# round-3 dev: in-kernel phase stamps + ablations of the brick kernel on the present build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
PCCM_BRICK_STAMP=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-graph --no-extras --no-cpu-baseline > $O/stamp.json 2> $O/stamp.err
tail -4 $O/stamp.err
for a in 0 1 2 4 7 8 16 32; do
  PCCM_BRICK_ABLATE=$a timeout -k 10 300 python bench.py --steps 100 --no-graph --no-extras --no-cpu-baseline > $O/b_$a.json 2> $O/b_$a.err; python -c "
import json; d=json.load(open('$O/b_$a.json')); print('ablate=$a ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
