#!/bin/bash
# round-4 dev: stagger of the launch's first workgroups x scan-loop variants of k_brick_query (DIAG build), 1M and 8M
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
run() { env "$@" TAG="$*" timeout -k 10 200 python scripts/dev_r4_brick.py 2>&1 | grep RESULT >> $O/sweep.txt; }
for n in 1000000 8000000; do
  export N=$n; export STEPS=$([ $n = 1000000 ] && echo 40 || echo 12)
  run X=0
  for st in 1 2 3 4 6 8; do run PCCM_BRICK_STAGGER=$st; done
  run PCCM_BRICK_VAR=1
  run PCCM_BRICK_VAR=2
  for st in 2 4 6; do run PCCM_BRICK_VAR=1 PCCM_BRICK_STAGGER=$st; run PCCM_BRICK_VAR=2 PCCM_BRICK_STAGGER=$st; done
  run X=0
done
cat $O/sweep.txt
