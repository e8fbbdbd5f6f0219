#!/bin/bash
# round-2 dev: smoke, GPU tests, bench, kernel stats (run through gpurun)
set -o pipefail
mkdir -p gpurun_out/r2
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 gpurun_out/r2/smoke.log; exit 1; }
tail -1 gpurun_out/r2/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2/pytest.log
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline > gpurun_out/r2/bench_graph.json 2> gpurun_out/r2/bench_graph.err; echo "bench rc=$?"; cat gpurun_out/r2/bench_graph.json | cut -c1-1500
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2/prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 50 --no-graph --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2/bench_prof.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"; f=$(find gpurun_out/r2/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | head -25
