#!/bin/bash
# Round-2 profiles: per-kernel durations and PMC counters of the headline step (run through gpurun; raw output under
# gpurun_out/r02prof, summarised into profiles/r02/ by scripts/summarise_r02.py).
#   rocprofv3 rules on this pool: the program itself directly after `--`; --pmc passes with --kernel-trace only;
#   counters that do not fit one pass fail with "error code 38: Request exceeds the capabilities of the hardware to
#   collect" (that, not a hang, is what ended round 1's TCP/TCC pass), hence one small set per pass.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02prof"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
# 1. kernel durations of the default (hipGraph) run -- the command the driver runs, without the side records
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_graph" -o s -- python3 "$B" --steps 200 --no-extras --no-cpu-baseline > "$O/stats_graph.log" 2>&1 || echo "stats_graph failed"
# 2. the same eagerly (one launch per kernel: what the PMC passes see)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_eager" -o s -- python3 "$B" --steps 50 --no-graph --no-extras --no-cpu-baseline > "$O/stats_eager.log" 2>&1 || echo "stats_eager failed"
# 3. counters, one small set per pass
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc$i" -o p -- python3 "$B" --steps 10 --warmup 3 --no-graph --no-extras --no-cpu-baseline > "$O/pmc$i.log" 2>&1
  echo "pass $i [$set] rc=$? $(grep -c . "$O"/pmc$i/*counter_collection.csv 2>/dev/null | tail -1)"
done
# 4. the brute-force engine's kernels
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_brute" -o s -- python3 "$B" --engine brute --steps 3 --no-extras --no-cpu-baseline > "$O/stats_brute.log" 2>&1 || echo "stats_brute failed"
# 5. a plain run: the line the driver will see
cd "$R" && timeout -k 10 600 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"; echo "bench rc=$?"
python3 "$R/scripts/summarise_r02.py" "$O" "$O/summary" 2>&1 | tail -40
