#!/bin/bash
# Developer scratch: wall time of the command line on two 800k-point files (GPU box).
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from open_pcc_metric_amd.io import write_point_cloud
from open_pcc_metric_amd.point_cloud import PointCloud
rng = np.random.default_rng(0)
v = rng.standard_normal((1200000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
a = np.unique(np.round(v * 400 + 512), axis=0)
b = np.unique(np.round((v * 400 + 512) + rng.normal(0, 0.7, v.shape)), axis=0)
n = v[: len(a)]
write_point_cloud("/tmp/a.ply", PointCloud(a, n), binary=True)
write_point_cloud("/tmp/b.ply", PointCloud(b, v[: len(b)]), binary=True)
print(len(a), len(b))
PY
for k in 1 2; do time python -m open_pcc_metric_amd --ocloud /tmp/a.ply --pcloud /tmp/b.ply --hausdorff --point-to-plane --normal-index neighbour --extent 800 800 800 | tail -3; done
echo "--- with the torch import"; PCCM_NO_TORCH=0; export PCCM_NO_TORCH; time python -m open_pcc_metric_amd --ocloud /tmp/a.ply --pcloud /tmp/b.ply --hausdorff --point-to-plane --normal-index neighbour --extent 800 800 800 | tail -1
