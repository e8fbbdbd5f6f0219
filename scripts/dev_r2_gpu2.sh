#!/bin/bash
# round-2 dev: brick kernel variants (A/B), parity per variant, full suite on the default
set -o pipefail
mkdir -p gpurun_out/r2b
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2b/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2b/smoke.log; exit 1; }
tail -1 gpurun_out/r2b/smoke.log
for v in 512,4,2 256,4,2 256,2,2 512,4,4 1024,4,4; do
  PCCM_BRICK=$v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2b/parity_$v.log 2>&1; echo "variant $v parity rc=$? $(tail -1 gpurun_out/r2b/parity_$v.log)"
  PCCM_BRICK=$v timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline > gpurun_out/r2b/bench_$v.json 2> gpurun_out/r2b/bench_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2b/bench_$v.json"))
k=d["kernel_ms_total"]; n=10
print("variant $v eager ms/step", d["ms_per_step"], {a: round(b/n*1000,1) for a,b in k.items()})
PY
done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2b/pytest.log
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r2b/bench_graph.json 2> gpurun_out/r2b/bench_graph.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2b/bench_graph.json')); print('graph ms/step', d['ms_per_step'], d['kernel_ms_total'], d.get('end_to_end'))"
