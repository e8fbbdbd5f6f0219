#!/bin/bash
# round-2 dev: smoke, the whole -m gpu suite, the default bench line (one gpurun call)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2full; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc $(tail -3 $O/gpu_tests.log)"
[ $rc -eq 0 ] || { tail -60 $O/gpu_tests.log; exit 1; }
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","ms_per_step","kernel_us_per_step","roofline","full_report","cold_pair","end_to_end") if k in d})
PY
