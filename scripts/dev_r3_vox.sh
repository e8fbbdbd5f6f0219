#!/bin/bash
# round-3 dev: voxel-brick search (pccm_vox.hip) -- parity tests, then the content step with and without it
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3vox; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_config4_surrogate.py tests/test_gpu_tie_exposure.py tests/test_gpu_ab_paths.py tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15 &&
for v in 1 0; do
  PCCM_VOX=$v timeout -k 10 300 python bench.py --content-only --steps 100 > $O/c$v.json 2> $O/c$v.err; python -c "
import json; d=json.load(open('$O/c$v.json'))['content']; print('vox $v', d['ms_per_step'], d['kernel_us_per_step'], d['grid_cells'], d['mse_left'])"
done
