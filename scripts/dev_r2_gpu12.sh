#!/bin/bash
# round-2 dev: parity subset + kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_round2.py tests/test_gpu_sharded.py -x -q > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc $(tail -1 $O/parity.log)"
[ $rc -eq 0 ] || { tail -40 $O/parity.log; exit 1; }
timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d['roofline']['frac'])"
bash scripts/dev_r2_trace.sh
