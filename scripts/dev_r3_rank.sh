#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 400 python scripts/rank_profile.py 1000000 8000000 2>&1 | grep -E "world=8|world=1|world=4" 
