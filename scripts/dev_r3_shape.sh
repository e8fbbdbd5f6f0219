#!/bin/bash
# round-3 dev: brick shapes under the instruction-bound kernel (whole clouds, 1M and 8M)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3soa; mkdir -p $O
export TMPDIR=/tmp
for n in 1000000 8000000; do for sh in "4,2" "4,4" "2,2"; do
  PCCM_BRICK=$sh timeout -k 10 300 python bench.py --points $n --steps 60 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('n $n shape $sh', 'ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done; done
