#!/bin/bash
# round-2 dev: 8M points, bin size of the grid build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
for lg in 0 10 11 12; do
  PCCM_BUILD_LG=$lg timeout -k 10 300 python bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > $O/b8_$lg.json 2> $O/b8_$lg.err && python -c "
import json; d=json.load(open('$O/b8_$lg.json')); print('8M lg=$lg ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
for lg in 0 10 11; do
  PCCM_BUILD_LG=$lg timeout -k 10 300 python bench.py --points 4000000 --steps 20 --no-extras --no-cpu-baseline > $O/b4_$lg.json 2> $O/b4_$lg.err && python -c "
import json; d=json.load(open('$O/b4_$lg.json')); print('4M lg=$lg ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
