#!/bin/bash
# round-2 dev: 16-byte fp32-exact normals in the brick kernel's gather (A/B), build tile size with bin cursors, host-side changes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc $(tail -1 $O/gpu_tests.log)"
[ $rc -eq 0 ] || { tail -60 $O/gpu_tests.log; exit 1; }
run() {  # name, mode flags, env...
  name=$1; shift; flags=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 100 $flags --no-extras --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name FAILED"; tail -5 $O/bench_$name.err; return; }
  python - <<PY
import json
d=json.load(open("$O/bench_$name.json"))
print("$name ms/step", d["ms_per_step"], d.get("kernel_us_per_step"), d["roofline"]["frac"])
PY
}
run graph_default "" X=1
run graph_nrm64 "" PCCM_NRM32=0
run graph_tile2048 "" PCCM_BUILD_TILE=2048
run graph_tile8192 "" PCCM_BUILD_TILE=8192
run graph_default2 "" X=1
run eager_default --no-graph X=1
run eager_nrm64 --no-graph PCCM_NRM32=0
timeout -k 10 300 python scripts/host_split.py --points 1000000 2>&1 | tail -1
