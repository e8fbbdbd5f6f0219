#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2l
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2l/pytest.log)"
grep -n "^FAILED\|^E " gpurun_out/r2l/pytest.log | head
