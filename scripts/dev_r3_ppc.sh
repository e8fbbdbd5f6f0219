#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
for v in "PCCM_GRID_PPC=1.4" "PCCM_GRID_PPC=1.33" "PCCM_GRID_PPC=1.28" "PCCM_GRID_PPC=1.22" "PCCM_GRID_PPC=1.5"; do
  env $v timeout -k 10 200 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$v ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d['config'].get('grid_cells'))"
done
