#!/bin/bash
# round-3 dev (DIAG build): VALU instructions per wave of the brick kernel's phases (PMC over the timing-only ablations)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3valu; mkdir -p $O
export TMPDIR=/tmp
for a in 0 1 4 16; do
  rm -rf $O/p$a
  PCCM_BRICK_ABLATE=$a timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $O/p$a -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-extras --no-cpu-baseline > $O/b$a.json 2> $O/b$a.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/p$a/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        if 'k_brick_query' in r['Kernel_Name']:
            acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
out = {k: sum(v.values()) / max(len(v), 1) for k, v in acc.items()}
w = out.get('SQ_WAVES', 1)
print('ablate $a', {k: round(v / w, 1) for k, v in out.items()}, 'waves', w)
PY
done
