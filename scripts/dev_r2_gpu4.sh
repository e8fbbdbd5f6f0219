#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2d
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2d/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2d/smoke.log; exit 1; }
tail -1 gpurun_out/r2d/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_config3_8m.py > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2d/pytest.log)"
PCCM_REDUCE_MERGE=0 timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2d/bench_nomerge.json 2> gpurun_out/r2d/bench_nomerge.err
python -c "
import json; d=json.load(open('gpurun_out/r2d/bench_nomerge.json')); print('nomerge', d['ms_per_step'], d['kernel_us_per_step'])"
bash scripts/prof_r02.sh
