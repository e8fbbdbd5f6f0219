#!/bin/bash
# round-3 dev: brick kernel with SoA LDS planes -- parity subset, bench, kernel stats
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3soa; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_ab_paths.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3 &&
for e in "PCCM_X=0"; do
  env $e timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$e', 'ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
done
