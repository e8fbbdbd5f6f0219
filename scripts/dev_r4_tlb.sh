#!/bin/bash
# Round-4 developer scratch: address-translation counters of the brick kernel at 16M and 32M points per cloud (is the per-point
# slow-down at 32M the TLB?).  --pmc with --kernel-trace only; the program directly after `--`.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/tlb"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/avail.txt" 2>&1
grep -i -o "TCP_UTCL1[A-Z0-9_]*\|UTCL2[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCC_EA0_RDREQ[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Z0-9_]*" "$O/avail.txt" | sort -u | head -40
for n in 16000000 32000000; do
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $set | cut -c1-12)_$n
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/$tag" -o p -- python3 "$R/bench.py" --points $n --steps 2 --warmup 1 --no-graph --no-extras --no-cpu-baseline > "$O/$tag.log" 2>&1
    echo "$tag rc=$?"
    f=$(find "$O/$tag" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_brick_query" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v.sort(); print("   ", k, v[len(v) // 2], "launches", len(v))
PY
  done
done
