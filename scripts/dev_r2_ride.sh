#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2v; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_round2.py tests/test_gpu_config3_8m.py tests/test_gpu_bench_multi.py tests/test_gpu_parity.py -x -q > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc $(tail -1 $O/t.log)"
[ $rc -eq 0 ] || { tail -40 $O/t.log; exit 1; }
timeout -k 10 600 python scripts/rank_profile.py > $O/rank_profile.log 2>&1; grep -v "^{" $O/rank_profile.log | grep "direction" | tail -8
