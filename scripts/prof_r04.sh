#!/bin/bash
# Round-4 profiles: per-kernel durations, HBM traffic and busy-cycle counters of the headline step; the 8M and 32M lines with
# traffic and cache counters; the content paths (distances only, and the full D1 + D2 + colour report).  Raw output under
# gpurun_out/r04prof, summarised into profiles/r04/ by scripts/summarise_r04.py.
# rocprofv3 rules on this pool: the program itself directly after `--`; --pmc passes with --kernel-trace only.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04prof"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
st() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$1" -o s -- python3 "$B" "${@:2}" > "$O/$1.log" 2>&1 || echo "$1 failed"; }
pm() { local name=$1 set=$2; shift 2; timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/$name" -o p -- python3 "$B" "$@" > "$O/$name.log" 2>&1; echo "$name [$set] rc=$?"; }
# 1. the headline step: durations under the hipGraph replay and eagerly
st stats_graph --steps 200 --no-extras --no-cpu-baseline
st stats_eager --steps 50 --no-graph --no-extras --no-cpu-baseline
# 2. traffic + busy counters at 1M
E="--steps 10 --warmup 3 --no-graph --no-extras --no-cpu-baseline"
pm pmc1_fetch "FETCH_SIZE" $E
pm pmc1_write "WRITE_SIZE" $E
pm pmc1_tcc "TCC_HIT_sum TCC_MISS_sum" $E
pm busy1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" $E
pm busy2 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" $E
pm busy3 "SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_CYCLES" $E
pm busy4 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $E
pm busy5 "GRBM_GUI_ACTIVE GRBM_COUNT" $E
# 3. 8M vs 8M
st stats_8M --points 8000000 --steps 20 --no-extras --no-cpu-baseline
E8="--points 8000000 --steps 5 --warmup 2 --no-graph --no-extras --no-cpu-baseline"
pm pmc8_fetch "FETCH_SIZE" $E8
pm pmc8_write "WRITE_SIZE" $E8
# 4. 32M vs 32M: durations, traffic, L2 hit rate, waits
E32="--points 32000000 --steps 3 --warmup 2 --no-graph --no-extras --no-cpu-baseline"
st stats_32M $E32
pm pmc32_fetch "FETCH_SIZE" $E32
pm pmc32_write "WRITE_SIZE" $E32
pm pmc32_tcc "TCC_HIT_sum TCC_MISS_sum" $E32
pm pmc32_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $E32
# 5. content: distances only, and the full report with rows
st stats_content --content-only --steps 50 --no-graph
pm pmcC_fetch "FETCH_SIZE" --content-only --steps 10 --no-graph
pm pmcC_write "WRITE_SIZE" --content-only --steps 10 --no-graph
st stats_content_full --content-full-only --steps 30 --no-graph
pm pmcF_fetch "FETCH_SIZE" --content-full-only --steps 10 --no-graph
pm pmcF_write "WRITE_SIZE" --content-full-only --steps 10 --no-graph
# 6. plain runs: the line the driver will see, the 8M and 32M lines
cd "$R"
timeout -k 10 600 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > "$O/bench_8M.json" 2> "$O/bench_8M.err"; echo "bench 8M rc=$?"
timeout -k 10 300 python3 bench.py --points 32000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > "$O/bench_32M.json" 2> "$O/bench_32M.err"; echo "bench 32M rc=$?"
python3 "$R/scripts/summarise_r04.py" "$O" "$O/summary" 2>&1 | tail -70
