#!/bin/bash
# round-2 dev: brick shape sweep with the final kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 100 --no-extras --no-cpu-baseline > $O/b_$name.json 2> $O/b_$name.err || { echo "$name FAILED"; tail -3 $O/b_$name.err; return; }
  python -c "
import json; d=json.load(open('$O/b_$name.json')); print('$name ms/step', d['ms_per_step'], d['kernel_us_per_step']['grid_query'], d['kernel_us_per_step']['grid_finish'])"
}
run default X=1
run bx32 PCCM_BRICK_BX=32
run bx40 PCCM_BRICK_BX=40
run bx56 PCCM_BRICK_BX=56
run bx64 PCCM_BRICK_BX=64
run nt448 PCCM_BRICK=4,2,448
run nt576 PCCM_BRICK=4,2,576
run nt640 PCCM_BRICK=4,2,640
run cap1800 PCCM_BRICK_CAP=1800
run cap1800_nt384 PCCM_BRICK_CAP=1800 PCCM_BRICK=4,2,384
run cap2400 PCCM_BRICK_CAP=2400
run s22 PCCM_BRICK=2,2
run s44 PCCM_BRICK=4,4
run bx32_s44 PCCM_BRICK_BX=32 PCCM_BRICK=4,4
run default2 X=1
