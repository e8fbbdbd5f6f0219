#!/bin/bash
# round-3 dev: per-kernel durations of the content step with the voxel-brick search
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3vox"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o s -- python3 "$R/bench.py" --content-only --steps 50 --no-graph > "$O/stats.log" 2>&1 || echo "stats failed"
tail -1 "$O/stats.log" | cut -c1-400
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
