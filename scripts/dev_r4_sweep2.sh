#!/bin/bash
# round-4 dev: how the number of bricks falls into rounds of resident workgroups (brick length, cells per point), 1M and 8M
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
run() { env "$@" TAG="$*" timeout -k 10 200 python scripts/dev_r4_brick.py 2>&1 | grep RESULT >> $O/sweep.txt; }
for n in 1000000 8000000; do
  export N=$n; export STEPS=$([ $n = 1000000 ] && echo 40 || echo 12)
  run X=0
  for bx in 16 23 30 45 64; do run PCCM_BRICK_BX=$bx; run PCCM_BRICK_BX=$bx PCCM_BRICK_VAR=1; done
  for ppc in 1.3 1.5 1.6; do run PCCM_GRID_PPC=$ppc; run PCCM_GRID_PPC=$ppc PCCM_BRICK_VAR=1; run PCCM_GRID_PPC=$ppc PCCM_BRICK_BX=30 PCCM_BRICK_VAR=1; done
done
cat $O/sweep.txt
