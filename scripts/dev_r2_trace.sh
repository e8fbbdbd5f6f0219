#!/bin/bash
# per-kernel durations of the default (hipGraph) step under rocprofv3
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r2trace"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_graph" -o s -- python3 "$R/bench.py" --steps 200 --no-extras --no-cpu-baseline > "$O/stats_graph.log" 2>&1 || echo "stats_graph failed"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats_graph/s_kernel_stats.csv")))
for r in rows[:10]: print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
tail -1 "$O/stats_graph.log" | cut -c1-300
