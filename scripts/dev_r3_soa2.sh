#!/bin/bash
# round-3 dev: points-per-cell sweep with the trimmed SoA brick kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3soa; mkdir -p $O
export TMPDIR=/tmp
for ppc in 1.2 1.3 1.4 1.5 1.6 1.8; do
  PCCM_GRID_PPC=$ppc timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('ppc $ppc', 'ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d.get('grid_cells'))"
done
