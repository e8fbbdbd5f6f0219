#!/bin/bash
# Round 4: busy-cycle counters of the headline step's kernels ("prove what bounds k_brick_query").
# Usage (through gpurun): bash scripts/prof_r04_busy.sh <outdir-under-gpurun_out> [extra bench.py args]
# rocprofv3 rules on this pool: the program itself directly after `--`; --pmc passes with --kernel-trace only.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/${1:-r04busy}"; shift; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
EXTRA="$*"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/busy$i" -o p -- python3 "$B" --steps 10 --warmup 3 --no-graph --no-extras --no-cpu-baseline $EXTRA > "$O/busy$i.log" 2>&1
  echo "busy pass $i [$set] rc=$?"
done
python3 "$R/scripts/summarise_pmc.py" "$O" 'busy\d+' > "$O/busy_summary.txt" 2>&1
tail -80 "$O/busy_summary.txt"
