#!/bin/bash
# round-2 dev: brick staging per run (wave w copies runs w, w+8, w+16) -- parity + bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_round2.py -x -q > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc $(tail -1 $O/parity.log)"
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
run graph_default "" X=1
run graph_default2 "" X=1
run eager_default --no-graph X=1
run graph_ppc1.2 "" PCCM_GRID_PPC=1.2
run graph_ppc1.3 "" PCCM_GRID_PPC=1.3
run graph_ppc1.7 "" PCCM_GRID_PPC=1.7
