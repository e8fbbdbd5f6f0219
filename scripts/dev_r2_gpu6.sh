#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2f/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2f/smoke.log; exit 1; }
tail -1 gpurun_out/r2f/smoke.log
for v in 4,2,512 4,2,576 2,2,256 4,2,256; do
  PCCM_BRICK=$v PCCM_BRICK_STAMP=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2f/stamp_$v.json 2> gpurun_out/r2f/stamp_$v.err
  echo "variant $v"; grep "brick stamps" gpurun_out/r2f/stamp_$v.err | tail -2
done
timeout -k 10 300 python scripts/dev_host_profile.py > gpurun_out/r2f/host.log 2>&1; head -30 gpurun_out/r2f/host.log
