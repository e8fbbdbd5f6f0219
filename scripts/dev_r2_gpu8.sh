#!/bin/bash
# round-2 dev: build tile size / bin size with bin cursors
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; mkdir -p $O
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
run graph_tile12288 "" PCCM_BUILD_TILE=12288
run graph_tile16384 "" PCCM_BUILD_TILE=16384
run graph_tile6144 "" PCCM_BUILD_TILE=6144
run graph_lg11 "" PCCM_BUILD_LG=11
run graph_lg11_t16k "" PCCM_BUILD_LG=11 PCCM_BUILD_TILE=16384
run graph_default2 "" X=1
for n in 8000000; do
  timeout -k 10 300 python bench.py --points $n --steps 20 --no-extras --no-cpu-baseline > $O/bench_8m.json 2> $O/bench_8m.err && python -c "
import json; d=json.load(open('$O/bench_8m.json')); print('8M ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d['roofline']['frac'])"
  PCCM_BUILD_TILE=4096 timeout -k 10 300 python bench.py --points $n --steps 20 --no-extras --no-cpu-baseline > $O/bench_8m_t4096.json 2> $O/bench_8m.err && python -c "
import json; d=json.load(open('$O/bench_8m_t4096.json')); print('8M tile4096 ms/step', d['ms_per_step'], d.get('kernel_us_per_step'), d['roofline']['frac'])"
done
