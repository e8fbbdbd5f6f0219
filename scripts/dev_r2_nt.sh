#!/bin/bash
# round-2 dev: non-temporal result stores / normal loads (compile-time A/B, rebuilt on the box)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
b() { timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline > $O/b_$1.json 2> $O/b_$1.err && python -c "
import json; d=json.load(open('$O/b_$1.json')); print('$1 ms/step', d['ms_per_step'], d['kernel_us_per_step'], d['parity_vs_oracle'] if 'parity_vs_oracle' in d else '')"; }
F="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
b base
for v in "-DPCCM_NT_RESULTS" "-DPCCM_NT_NORMALS" "-DPCCM_NT_RESULTS -DPCCM_NT_NORMALS"; do
  (cd open_pcc_metric_amd/csrc && touch pccm_brick.hip pccm_grid.h && make -j8 CXXFLAGS="$F $v" > /dev/null 2>&1) || { echo build failed; exit 1; }
  b "$(echo $v | tr -d ' -')"
done
(cd open_pcc_metric_amd/csrc && touch pccm_brick.hip pccm_grid.h && make -j8 > /dev/null 2>&1)
b base_again
