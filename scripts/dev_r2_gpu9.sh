#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2i
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2i/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2i/smoke.log; exit 1; }
tail -1 gpurun_out/r2i/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2i/pytest.log)"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2i/bench_$name.json 2> gpurun_out/r2i/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2i/bench_$name.json"))
    print("$name eager ms/step", d["ms_per_step"], d["kernel_us_per_step"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/r2i/bench_$name.err").read()[-400:])
PY
}
run default X=1
run lg11 PCCM_BUILD_LG=11
run lg12 PCCM_BUILD_LG=12
run lg13 PCCM_BUILD_LG=13
run lg12t4096 PCCM_BUILD_LG=12 PCCM_BUILD_TILE=4096
run t4096 PCCM_BUILD_TILE=4096
timeout -k 10 600 python scripts/rank_profile.py > gpurun_out/r2i/rank_profile.log 2>&1; grep -v "^{" gpurun_out/r2i/rank_profile.log | tail -20
