#!/bin/bash
# round-2 dev: points per cell x problem size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
for n in 500000 1000000 2000000 4000000 8000000; do
  for ppc in 1.2 1.3 1.4 1.5; do
    PCCM_GRID_PPC=$ppc timeout -k 10 300 python bench.py --points $n --steps 40 --no-extras --no-cpu-baseline > $O/b_${n}_$ppc.json 2> $O/b_${n}_$ppc.err || { echo "$n $ppc FAILED"; tail -3 $O/b_${n}_$ppc.err; continue; }
    python - <<PY
import json
d=json.load(open("$O/b_${n}_$ppc.json"))
print("n=$n ppc=$ppc ms/step", d["ms_per_step"], d.get("kernel_us_per_step"))
PY
  done
done
