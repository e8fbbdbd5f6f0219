#!/bin/bash
# round-3 dev: the persistent brick kernel against round 2's, parity first
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
for v in "PCCM_BRICK_PERSIST=1" "PCCM_BRICK_PERSIST=1 PCCM_BRICK_NT=512" "PCCM_BRICK_PERSIST=0" $EXTRA; do
  env $v timeout -k 10 200 python bench.py --steps 100 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$v ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
