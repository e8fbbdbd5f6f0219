#!/bin/bash
# round-3 dev: points-per-cell x brick length sweep with the SoA brick kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3soa; mkdir -p $O
export TMPDIR=/tmp
for ppc in 0.7 0.9 1.1 1.4; do for bx in 48 64; do
  PCCM_GRID_PPC=$ppc PCCM_BRICK_BX=$bx timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('ppc $ppc bx $bx', 'ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done; done
