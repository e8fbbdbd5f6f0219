#!/bin/bash
# Developer scratch: SQ / TCP / TCC counters of the grid kernels (separate rocprofv3 --pmc passes over bench.py --no-graph).
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_coop; mkdir -p $O; cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           ; do   # the TCP_*_LATENCY / TCC_* counters do not fit one pass ("error code 38: Request exceeds the capabilities of the
                  # hardware to collect", SIGABRT in rocprofv3): scripts/prof_r02.sh collects them one small set per pass
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graph > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/pmc_coop/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "pccm" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if any(s in k for s in ("k_grid_query_coop", "k_grid_cells", "k_point_jobs", "k_grid_scatter")):
        print(k)
        for c, v in sorted(acc[k].items()):
            v = sorted(v)
            print("   %-34s %16.1f  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
