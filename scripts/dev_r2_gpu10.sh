#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2j
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2j/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2j/smoke.log; exit 1; }
tail -1 gpurun_out/r2j/smoke.log
timeout -k 10 1100 python -m pytest tests/test_gpu_round2.py tests/test_gpu_bench_multi.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2j/pytest.log)"
timeout -k 10 600 python scripts/rank_profile.py > gpurun_out/r2j/rank_profile.log 2>&1; grep -v "^{" gpurun_out/r2j/rank_profile.log | tail -16
timeout -k 10 300 python bench.py --steps 200 > gpurun_out/r2j/bench_line.json 2> gpurun_out/r2j/bench_line.err; python -c "
import json; d=json.load(open('gpurun_out/r2j/bench_line.json')); print('graph ms/step', d['ms_per_step'], d['kernel_us_per_step'], d['roofline']['frac'], d['full_report'], d['cold_pair'], d['end_to_end'], d['brute'], d['parity_vs_oracle'])"
