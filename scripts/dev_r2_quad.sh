#!/bin/bash
# round-2 dev: quad loads in the build's count/scatter kernels -- parity subset + bench + trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_round2.py tests/test_gpu_sharded.py tests/test_gpu_normals.py -x -q > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc $(tail -1 $O/parity.log)"
[ $rc -eq 0 ] || { tail -40 $O/parity.log; exit 1; }
for k in 1 2; do timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/bench$k.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench$k.json')); print('ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"; done
timeout -k 10 300 python bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > $O/bench8.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench8.json')); print('8M ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
bash scripts/dev_r2_trace.sh
