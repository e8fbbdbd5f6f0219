#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2g
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2g/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2g/smoke.log; exit 1; }
tail -1 gpurun_out/r2g/smoke.log
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r2g/parity.log 2>&1; echo "parity rc=$? $(tail -1 gpurun_out/r2g/parity.log)"
for v in 4,2,512 4,2,576 2,2,256; do
  PCCM_BRICK=$v PCCM_BRICK_STAMP=1 timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2g/stamp_$v.json 2> gpurun_out/r2g/stamp_$v.err
  echo "variant $v"; grep "brick stamps" gpurun_out/r2g/stamp_$v.err | tail -1
done
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2g/bench_$name.json 2> gpurun_out/r2g/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2g/bench_$name.json"))
    print("$name eager ms/step", d["ms_per_step"], d["kernel_us_per_step"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/r2g/bench_$name.err").read()[-400:])
PY
}
run default X=1
run nt512 PCCM_BRICK=4,2,512
run nt384 PCCM_BRICK=4,2,384
run nt256 PCCM_BRICK=4,2,256
run b22_256 PCCM_BRICK=2,2,256
run b44_1024 PCCM_BRICK=4,4,1024
run b44_512 PCCM_BRICK=4,4,512
run bx22_256 PCCM_BRICK_BX=22 PCCM_BRICK=4,2,256
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-extras > gpurun_out/r2g/bench_graph.json 2> gpurun_out/r2g/bench_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r2g/bench_graph.json')); print('graph ms/step', d['ms_per_step'], d['kernel_us_per_step'])"
