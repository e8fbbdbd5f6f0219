#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2e/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2e/smoke.log; exit 1; }
tail -1 gpurun_out/r2e/smoke.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r2e/parity.log 2>&1; echo "parity rc=$? $(tail -1 gpurun_out/r2e/parity.log)"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2e/bench_$name.json 2> gpurun_out/r2e/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2e/bench_$name.json"))
    print("$name eager ms/step", d["ms_per_step"], d["kernel_us_per_step"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/r2e/bench_$name.err").read()[-400:])
PY
}
run default X=1
run nt512 PCCM_BRICK=4,2,512
run nt640 PCCM_BRICK=4,2,640
run nt256 PCCM_BRICK=4,2,256
run b22_256 PCCM_BRICK=2,2,256
run b22_320 PCCM_BRICK=2,2,320
run bx30 PCCM_BRICK_BX=30
run bx30_384 PCCM_BRICK_BX=30 PCCM_BRICK=4,2,384
run bx22_256 PCCM_BRICK_BX=22 PCCM_BRICK=4,2,256
run red1 PCCM_REDUCE_VARIANT=1
run red2 PCCM_REDUCE_VARIANT=2
run tile4096 PCCM_BUILD_TILE=4096
N=800000 timeout -k 10 300 python scripts/dev_surface_check.py > gpurun_out/r2e/surface.log 2>&1; tail -8 gpurun_out/r2e/surface.log
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-extras > gpurun_out/r2e/bench_graph.json 2> gpurun_out/r2e/bench_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r2e/bench_graph.json')); print('graph ms/step', d['ms_per_step'], d['kernel_us_per_step'])"
