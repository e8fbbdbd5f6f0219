#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2k
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2k/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2k/smoke.log; exit 1; }
tail -1 gpurun_out/r2k/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2k/pytest.log)"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 40 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r2k/bench_$name.json 2> gpurun_out/r2k/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2k/bench_$name.json"))
    print("$name eager ms/step", d["ms_per_step"], d["kernel_us_per_step"])
except Exception as e:
    print("$name FAILED", e, open("gpurun_out/r2k/bench_$name.err").read()[-400:])
PY
}
run default X=1
run again X=1
timeout -k 10 300 python bench.py --steps 200 > gpurun_out/r2k/bench_line.json 2> gpurun_out/r2k/bench_line.err; python -c "
import json; d=json.load(open('gpurun_out/r2k/bench_line.json')); print('graph ms/step', d['ms_per_step'], d['kernel_us_per_step'], d['roofline']['frac'], d['full_report']['ms_per_step'], d['cold_pair']['ms_per_pair'], d['end_to_end']['ms_per_pair'], d['parity_vs_oracle'])"
