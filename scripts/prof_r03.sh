#!/bin/bash
# Round-3 profiles: per-kernel durations and PMC counters of the headline step, the 8M and 32M lines, the content path and the
# per-rank profile (run through gpurun; raw output under gpurun_out/r03prof, summarised into profiles/r03/ by
# scripts/summarise_r03.py).  rocprofv3 rules on this pool: the program itself directly after `--`; --pmc passes with
# --kernel-trace only; one small counter set per pass.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03prof"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
# 1. kernel durations of the default (hipGraph) run -- the command the driver runs, without the side records
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_graph" -o s -- python3 "$B" --steps 200 --no-extras --no-cpu-baseline > "$O/stats_graph.log" 2>&1 || echo "stats_graph failed"
# 2. the same eagerly (one launch per kernel: what the PMC passes see)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_eager" -o s -- python3 "$B" --steps 50 --no-graph --no-extras --no-cpu-baseline > "$O/stats_eager.log" 2>&1 || echo "stats_eager failed"
# 3. counters at 1M, one small set per pass
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc$i" -o p -- python3 "$B" --steps 10 --warmup 3 --no-graph --no-extras --no-cpu-baseline > "$O/pmc$i.log" 2>&1
  echo "pass $i [$set] rc=$?"
done
# 4. 8M vs 8M (BASELINE configs[3] on one GPU): kernel durations + traffic
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_8M" -o s -- python3 "$B" --points 8000000 --steps 20 --no-extras --no-cpu-baseline > "$O/stats_8M.log" 2>&1 || echo "stats_8M failed"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc8M_$set" -o p -- python3 "$B" --points 8000000 --steps 5 --warmup 2 --no-graph --no-extras --no-cpu-baseline > "$O/pmc8M_$set.log" 2>&1
  echo "8M pass [$set] rc=$?"
done
# 5. the content path (voxelised surface pair): kernel durations + traffic
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_content" -o s -- python3 "$B" --content-only --steps 50 --no-graph > "$O/stats_content.log" 2>&1 || echo "stats_content failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmcC_$n" -o p -- python3 "$B" --content-only --steps 10 --no-graph > "$O/pmcC_$n.log" 2>&1
  echo "content pass [$set] rc=$?"
done
# 6. the brute-force engine's kernels
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_brute" -o s -- python3 "$B" --engine brute --steps 3 --no-extras --no-cpu-baseline > "$O/stats_brute.log" 2>&1 || echo "stats_brute failed"
# 7. plain runs: the line the driver will see, the 8M and 32M lines, the per-rank profile
cd "$R"
timeout -k 10 600 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > "$O/bench_8M.json" 2> "$O/bench_8M.err"; echo "bench 8M rc=$?"
timeout -k 10 300 python3 bench.py --points 32000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > "$O/bench_32M.json" 2> "$O/bench_32M.err"; echo "bench 32M rc=$?"
timeout -k 10 600 python3 scripts/rank_profile.py > "$O/rank_profile.log" 2>&1; echo "rank profile rc=$?"
cp -f gpurun_out/rank_profile.json "$O/rank_profile.json" 2>/dev/null
python3 "$R/scripts/summarise_r03.py" "$O" "$O/summary" 2>&1 | tail -60
