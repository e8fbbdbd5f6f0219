#!/bin/bash
# Round-3 profile of the content path (voxelised-surface pair): per-kernel durations + HBM traffic counters.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03content"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o s -- python3 "$B" --content-only --steps 50 --no-graph > "$O/stats.log" 2>&1 || echo "stats failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc$i" -o p -- python3 "$B" --content-only --steps 10 --no-graph > "$O/pmc$i.log" 2>&1
  echo "pass $i [$set] rc=$?"
done
tail -1 "$O/stats.log"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
for i in (1,2,3,4):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"$O/pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(t in k for t in ("k_grid_query", "k_bin_", "k_unit", "k2b")):
            print(i, k, {c: round(sorted(x)[len(x)//2], 1) for c, x in v.items()})
PY
