#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
for lg in 0 9 8; do
  PCCM_BUILD_LG=$lg timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b_$lg.json 2> $O/b_$lg.err && python -c "
import json; d=json.load(open('$O/b_$lg.json')); print('1M lg=$lg ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
