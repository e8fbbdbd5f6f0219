#!/bin/bash
# Round-3: SQ counters of the content step's kernels (voxel-brick search), merged into profiles/r03/content_0.8M_pmc_counters.json
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03content_sq"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/p"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$O/p" -o p -- python3 "$R/bench.py" --content-only --steps 10 --no-graph > "$O/p.log" 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "pccm" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sorted(v)[len(v)//2] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("$O/content_sq.json", "w"), indent=1)
for k, v in out.items():
    if "vox" in k: print(k, v)
PY
