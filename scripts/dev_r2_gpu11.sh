#!/bin/bash
# round-2 dev: threads per tile of the grid build's count/scatter kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_round2.py tests/test_gpu_normals.py -x -q > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc $(tail -1 $O/parity.log)"
[ $rc -eq 0 ] || { tail -40 $O/parity.log; exit 1; }
run() {  # name, mode flags, env...
  name=$1; shift; flags=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 100 $flags --no-extras --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name FAILED"; tail -5 $O/bench_$name.err; return; }
  python - <<PY
import json
d=json.load(open("$O/bench_$name.json"))
print("$name ms/step", d["ms_per_step"], d.get("kernel_us_per_step"), d["roofline"]["frac"])
PY
}
run graph_t1024 "" X=1
run graph_t256 "" PCCM_BUILD_THREADS=256
run graph_t512 "" PCCM_BUILD_THREADS=512
run graph_t1024_tile16k "" PCCM_BUILD_TILE=16384
run graph_t1024_tile4k "" PCCM_BUILD_TILE=4096
run graph_t1024b "" X=1
PCCM_BUILD_THREADS=256 timeout -k 10 300 python bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > $O/bench_8m_256.json 2> $O/bench_8m.err && python -c "
import json; d=json.load(open('$O/bench_8m_256.json')); print('8M t256 ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
timeout -k 10 300 python bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > $O/bench_8m.json 2> $O/bench_8m.err && python -c "
import json; d=json.load(open('$O/bench_8m.json')); print('8M t1024 ms/step', d['ms_per_step'], d.get('kernel_us_per_step'))"
