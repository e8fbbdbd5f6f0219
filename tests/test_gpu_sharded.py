"""The N > 1 path on the real kernels: 2 / 3 / 4 ranks (gloo, sharing the one GPU of the test box -- the box admits
at most six GPU processes at once, the test runner itself being one of them) split the pair by direction and shard the query rows inside each half (or shard rows only:
shard_mode="rows") and must print the single-process report bit for bit -- geometry, Hausdorff, D2 and the colour rows,
eager and hipGraph."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["PCCM_ROOT"])
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
dist.init_process_group("gloo")
n = int(os.environ["PCCM_N"])
rng = np.random.default_rng(7)
a = rng.random((n, 3), dtype=np.float32); b = rng.random((n + 1000, 3), dtype=np.float32)
na = rng.standard_normal((n, 3)); nb = rng.standard_normal((n + 1000, 3))
ca = rng.integers(0, 256, (n, 3)) / 255.0; cb = rng.integers(0, 256, (n + 1000, 3)) / 255.0
group = dist.group.WORLD if os.environ["PCCM_SHARD"] == "1" else None
pair = CloudPair(PointCloud(a, na, ca), PointCloud(b, nb, cb), extent=[1, 1, 1], normal_index="neighbour", device=0, group=group,
                 use_graph=os.environ["PCCM_GRAPH"] == "1", shard_mode=os.environ.get("PCCM_MODE", "direction"))
rows = None
for rep in range(3):
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions("ycc", True, True))).as_dict()
    now = [[list(map(str, k)), [float(x).hex() for x in np.atleast_1d(v)]] for k, v in res.items()]
    assert rows is None or rows == now
    rows = now
    pair.recompute()
col = np.asarray(pair.get_left_neighbour_distances())
with open(os.path.join(os.environ["PCCM_OUT"], f"rank{dist.get_rank()}.json"), "w") as fh:
    json.dump({"rows": rows, "col_sum": float(np.sum(col)).hex(), "shard": list(pair._engine.shard_range(0)),
               "shards": [list(pair._engine.shard_range(d)) for d in (0, 1, 2)]}, fh)
dist.destroy_process_group()
'''


def _run(tmp_path, nproc, shard, graph, n, mode="direction"):
    out = tmp_path / f"out_{nproc}_{shard}_{graph}_{mode}"
    out.mkdir()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCCM_ROOT=ROOT, PCCM_N=str(n), PCCM_OUT=str(out), PCCM_SHARD=shard, PCCM_GRAPH=graph,
               PCCM_MODE=mode, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    return [json.load(open(out / f"rank{r}.json")) for r in range(nproc)]


def _tiles(ranks, d, n_iter):
    owned = sorted(tuple(r["shards"][d]) for r in ranks if r["shards"][d][1] > r["shards"][d][0])
    return owned[0][0] == 0 and owned[-1][1] == n_iter and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))


@pytest.mark.parametrize("graph", ["0", "1"])
@pytest.mark.parametrize("mode", ["direction", "rows"])
def test_two_ranks_on_the_real_kernels_match_one_process(tmp_path, graph, mode):
    n = 150000
    single = _run(tmp_path, 1, "0", graph, n)[0]
    ranks = _run(tmp_path, 2, "1", graph, n, mode)
    assert ranks[0]["rows"] == ranks[1]["rows"] == single["rows"]
    assert ranks[0]["col_sum"] == ranks[1]["col_sum"] == single["col_sum"]
    assert _tiles(ranks, 0, n) and _tiles(ranks, 1, n + 1000) and _tiles(ranks, 2, n)
    if mode == "direction":
        assert ranks[0]["shards"][0] == [0, n] and ranks[0]["shards"][1] == [0, 0]          # rank 0: the whole left direction
        assert ranks[1]["shards"][0] == [0, 0] and ranks[1]["shards"][1] == [0, n + 1000]   # rank 1: the right one (+ self)


@pytest.mark.parametrize("world", [3, 4])
def test_more_ranks_split_by_direction_then_rows(tmp_path, world):
    n = 100003
    single = _run(tmp_path, 1, "0", "0", n)[0]
    ranks = _run(tmp_path, world, "1", "0", n)
    assert all(r["rows"] == single["rows"] for r in ranks)
    assert all(r["col_sum"] == single["col_sum"] for r in ranks)
    assert _tiles(ranks, 0, n) and _tiles(ranks, 1, n + 1000) and _tiles(ranks, 2, n)
