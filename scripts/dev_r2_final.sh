#!/bin/bash
# smoke + whole GPU suite + default bench line, then the round-2 profile passes (one gpurun call)
cd "$GRAFT_REPO_ROOT"
bash scripts/dev_r2_full.sh || exit 1
bash scripts/prof_r02.sh
