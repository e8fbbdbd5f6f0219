#!/bin/bash
# round-2 dev: b128 scan loop, points-per-cell sweep, build tile size, new config[3]/[4] tests
set -o pipefail
mkdir -p gpurun_out/r2c
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2c/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2c/smoke.log; exit 1; }
tail -1 gpurun_out/r2c/smoke.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q > gpurun_out/r2c/parity.log 2>&1; echo "parity rc=$? $(tail -1 gpurun_out/r2c/parity.log)"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline > gpurun_out/r2c/bench_$name.json 2> gpurun_out/r2c/bench_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2c/bench_$name.json"))
k=d["kernel_ms_total"]; n=40
print("$name eager ms/step", d["ms_per_step"], {a: round(b/n*1000,1) for a,b in k.items()})
PY
}
run default X=1
run b256_2_2 PCCM_BRICK=256,2,2
run ppc1.0 PCCM_GRID_PPC=1.0
run ppc1.2 PCCM_GRID_PPC=1.2
run ppc2.0 PCCM_GRID_PPC=2.0
run tile4096 PCCM_BUILD_TILE=4096
run tile1024 PCCM_BUILD_TILE=1024
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r2c/bench_graph.json 2> gpurun_out/r2c/bench_graph.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2c/bench_graph.json')); print('graph ms/step', d['ms_per_step'], d['kernel_ms_total'], d.get('end_to_end'))"
timeout -k 10 900 python -m pytest tests/test_gpu_config4_surrogate.py -x -q > gpurun_out/r2c/cfg4.log 2>&1; echo "cfg4 rc=$? $(tail -3 gpurun_out/r2c/cfg4.log)"
timeout -k 10 1100 python -m pytest tests/test_gpu_config3_8m.py -x -q > gpurun_out/r2c/cfg3.log 2>&1; echo "cfg3 rc=$? $(tail -3 gpurun_out/r2c/cfg3.log)"
