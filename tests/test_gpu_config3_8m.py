"""BASELINE.json configs[3]: 8M vs 8M uniform fp32 clouds with normals, D1 + D2 + Hausdorff.

* one context: every row of the full report equals the oracle's (exact kd-tree, C + OpenMP) bit for bit, and
  size-independent properties hold at full size (fused sums == NumPy's, the returned rows are valid witnesses);
* the pair split over 2 ``gloo`` ranks that share the test box's GPU (the N > 1 path of bench.py --gpus N on the
  real kernels): both ranks print the same rows as the single context.
Needs an MI355X (``-m gpu``) and about two minutes of host time for the oracle at this size."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import same_bits
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 8_000_000


def synth(n):
    """the benchmark's clouds at this size."""
    a = np.random.default_rng(1234).random((n, 3), dtype=np.float32)
    b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)

    def unit(seed):
        g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
        return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)

    return a, b, unit(4321), unit(8765)


def hexrows(res):
    return [[list(map(str, k)), [float(x).hex() for x in np.atleast_1d(v)]] for k, v in res.items()]


@pytest.fixture(scope="module")
def single():
    a, b, na, nb = synth(N)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0])
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
    yield {"pair": pair, "res": res, "clouds": (a, b, na, nb)}
    pair.close()


def test_config3_8m_single_context_bit_exact(single):
    a, b, na, nb = single["clouds"]
    pair, res = single["pair"], single["res"]
    o = orc.OraclePair(a, b, na, nb, method="kdtree")
    want = o.report(hausdorff=True, point_to_plane_=True, peak=1.0)
    assert list(res.keys()) == list(want.keys())
    for k in want:
        assert same_bits(res[k], want[k]), (k, res[k], want[k])
    # properties at full size
    col = pair.get_right_neighbour_distances()
    host = np.asarray(col)
    assert np.sum(col, axis=0) == np.sum(host, axis=0) and np.max(col) == host.max()
    idx = pair._neighbour_index(1)
    assert np.array_equal(idx, o.nn_idx[1])
    a64, b64 = o.points
    d = b64 - a64[idx]
    assert np.array_equal((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], host)
    stats = [pair._engine.nn_stats(k) for k in (0, 1)]
    assert all(s["pairs"] == 0 for s in stats)                      # the grid engine ran (not the brute-force scan)


WORKER = r'''
import json, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["PCCM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCCM_ROOT"], "tests"))
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from test_gpu_config3_8m import synth, hexrows, N
dist.init_process_group("gloo")
a, b, na, nb = synth(N)
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=0, group=dist.group.WORLD)
res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
with open(os.path.join(os.environ["PCCM_OUT"], f"rank{dist.get_rank()}.json"), "w") as fh:
    json.dump({"rows": hexrows(res), "shards": [list(pair._engine.shard_range(d)) for d in (0, 1)]}, fh)
dist.destroy_process_group()
'''


def test_config3_8m_two_ranks_match_the_single_context(single, tmp_path):
    script = tmp_path / "worker8m.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCCM_ROOT=ROOT, PCCM_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    ranks = [json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1)]
    want = hexrows(single["res"])
    assert ranks[0]["rows"] == ranks[1]["rows"] == want
    # split by direction: rank 0 searches the whole left direction (and builds B's grid only), rank 1 the right one
    assert ranks[0]["shards"] == [[0, N], [0, 0]] and ranks[1]["shards"] == [[0, 0], [0, N]]
