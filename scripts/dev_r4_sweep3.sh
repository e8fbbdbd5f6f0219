#!/bin/bash
# round-4 dev: resident-workgroup brick kernel (pccm_bstream.hip) against the workgroup-per-brick kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p $O; rm -f $O/sweep.txt
export TMPDIR=/tmp
run() { env "$@" TAG="$*" timeout -k 10 200 python scripts/dev_r4_brick.py 2>&1 | grep -a "RESULT\|rror\|fault" >> $O/sweep.txt; }
for n in 1000000 8000000; do
  export N=$n; export STEPS=$([ $n = 1000000 ] && echo 40 || echo 12)
  run PCCM_BRICK_STREAM=0
  run PCCM_BRICK_STREAM=0 PCCM_BRICK_VAR=1
  run PCCM_BRICK_STREAM=1
  run PCCM_BRICK_STREAM=2
  run PCCM_BRICK_STREAM=1 PCCM_BRICK_STREAM_WGS=2
  run PCCM_BRICK_STREAM=1 PCCM_BRICK_STREAM_WGS=4
  [ $n = 1000000 ] && for ppc in 1.3 1.5; do run PCCM_BRICK_STREAM=1 PCCM_GRID_PPC=$ppc; done
done
cat $O/sweep.txt
