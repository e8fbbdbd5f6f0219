#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q > $O/parity.log 2>&1; echo "parity rc=$?"; tail -2 $O/parity.log
for v in "A=1" $EXTRA; do
  env $v timeout -k 10 200 python bench.py --steps 100 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$v ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d['config'].get('fallback_queries'))"
done
env timeout -k 10 200 python bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > $O/b8.json 2> $O/b8.err; python -c "
import json; d=json.load(open('$O/b8.json')); print('8M ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
