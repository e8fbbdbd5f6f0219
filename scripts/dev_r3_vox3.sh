#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3vox; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  PCCM_VOX_DBG=$v timeout -k 10 300 python bench.py --content-only --steps 30 > $O/d$v.json 2> $O/d$v.err; python -c "
import json; d=json.load(open('$O/d$v.json'))['content']; print('dbg $v', d['ms_per_step'], d['kernel_us_per_step'])"
done
