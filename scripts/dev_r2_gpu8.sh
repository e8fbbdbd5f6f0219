#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2h
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2h/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2h/smoke.log; exit 1; }
tail -1 gpurun_out/r2h/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2h/pytest.log)"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2h/bench_$name.json 2> gpurun_out/r2h/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2h/bench_$name.json"))
    print("$name eager ms/step", d["ms_per_step"], d["kernel_us_per_step"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/r2h/bench_$name.err").read()[-400:])
PY
}
run default X=1
run v64 PCCM_BRICK_V64=1
run nt576 PCCM_BRICK=4,2,576
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-extras > gpurun_out/r2h/bench_graph.json 2> gpurun_out/r2h/bench_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r2h/bench_graph.json')); print('graph ms/step', d['ms_per_step'], d['kernel_us_per_step'])"
