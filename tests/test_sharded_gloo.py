"""The N > 1 path on CPU: two ``gloo`` ranks shard the query axis, exchange the reduction vectors
with all-reduce and must reproduce the single-process report bit for bit.  The GPU engine is
replaced by the oracle-backed test double (tests/oracle_engine.py); the exchange, the shard
arithmetic, pccm_finish_sum and the metric DAG are the product's own code."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script, nproc, env, timeout):
    """Launch `nproc` gloo ranks of `script`; one retry on a fresh port, and ONLY if stderr shows a bind / rendezvous failure (the
    probed port can be taken between the probe and torchrun's bind on a busy host); any other failure is returned as it is."""
    import socket
    proc = None
    for _ in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        if proc.returncode == 0:
            break
        err = proc.stderr.lower()
        if not any(t in err for t in ("eaddrinuse", "address already in use", "rendezvouserror", "rendezvous")):
            break                      # a worker failed: surface it, never retry an assertion away
    return proc

WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PCCM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCCM_ROOT"], "tests"))
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle_engine import OracleEngine

dist.init_process_group("gloo")
n = int(os.environ["PCCM_N"])
rng = np.random.default_rng(42)
a = rng.random((n, 3), dtype=np.float32); b = rng.random((n + 37, 3), dtype=np.float32)
na = rng.standard_normal((n, 3)); nb = rng.standard_normal((n + 37, 3))
ca = rng.integers(0, 256, (n, 3)) / 255.0; cb = rng.integers(0, 256, (n + 37, 3)) / 255.0
pair = CloudPair(PointCloud(a, na, ca), PointCloud(b, nb, cb), extent=[1, 1, 1], normal_index="neighbour",
                 group=dist.group.WORLD, shard_mode=os.environ["PCCM_MODE"], _engine=OracleEngine())
res = MetricCalculator(pair).calculate(transform_options(CalculateOptions("ycc", True, True))).as_dict()
col = np.asarray(pair.get_right_neighbour_distances())          # all-gathered column
ev = np.asarray(pair.get_left_error_vector())
out = {"rank": dist.get_rank(), "shard": pair._engine.shard_range(0), "shards": [pair._engine.shard_range(d) for d in (0, 1, 2)],
       "rows": [[list(map(str, k)), [float(x).hex() for x in np.atleast_1d(v)]] for k, v in res.items()],
       "col_sum": float(np.sum(col)).hex(), "col_len": len(col), "ev_sum": float(np.sum(ev)).hex(),
       "calls": sorted(set(c[0] for c in pair._engine.calls if c[0].startswith("reduce")))}
with open(os.path.join(os.environ["PCCM_OUT"], f"rank{dist.get_rank()}.json"), "w") as fh:
    json.dump(out, fh)
dist.barrier()                      # (orderly teardown: a rank that leaves while a peer's gloo thread still talks to it aborts that peer)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n,world,mode", [(1000, 2, "direction"), (20011, 2, "direction"), (20011, 4, "direction"),
                                          (20011, 8, "direction"), (5000, 3, "direction"), (20011, 2, "rows"), (40000, 4, "rows")])
def test_gloo_ranks_match_single_process(tmp_path, n, world, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from open_pcc_metric_amd.calculator import MetricCalculator
    from open_pcc_metric_amd.cloud_pair import CloudPair
    from open_pcc_metric_amd.options import CalculateOptions, transform_options
    from open_pcc_metric_amd.point_cloud import PointCloud
    from oracle_engine import OracleEngine

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PCCM_ROOT=ROOT, PCCM_N=str(n), MASTER_ADDR="127.0.0.1", PCCM_OUT=str(tmp_path), PCCM_MODE=mode,
               OMP_NUM_THREADS="2")
    proc = _torchrun(script, world, env, 600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    outs = sorted((json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)), key=lambda o: o["rank"])
    assert [o["rank"] for o in outs] == list(range(world))
    assert all(o["rows"] == outs[0]["rows"] for o in outs)         # every rank holds the full result
    # row ownership: for every direction the ranks' shards tile [0, n_iter) exactly once, in rank order, on 128-row units
    for d, n_iter in ((0, n), (1, n + 37), (2, n)):
        owned = [tuple(o["shards"][d]) for o in outs if o["shards"][d][1] > o["shards"][d][0]]
        assert owned[0][0] == 0 and owned[-1][1] == n_iter
        assert all(a[1] == b[0] and a[1] % 128 == 0 for a, b in zip(owned, owned[1:]))
    # the exchange: one number per 8192-row chunk when every rank that shares a direction can own whole chunks (then the
    # shards start on chunk boundaries), else one per 128-row leaf -- the same choice on every rank
    sub = world if mode == "rows" else (world + 1) // 2
    want_calls = ["reduce_chunks"] if n >= sub * 8192 else ["reduce"]
    assert all(o["calls"] == want_calls for o in outs), [o["calls"] for o in outs]
    if want_calls == ["reduce_chunks"]:
        assert all(sh[0] % 8192 == 0 for o in outs for sh in o["shards"])
    if mode == "direction":
        # the split is by direction first: no rank owns rows of both the left and the right direction
        assert all(not (o["shards"][0][1] > o["shards"][0][0] and o["shards"][1][1] > o["shards"][1][0]) for o in outs)

    rng = np.random.default_rng(42)
    a = rng.random((n, 3), dtype=np.float32); b = rng.random((n + 37, 3), dtype=np.float32)
    na = rng.standard_normal((n, 3)); nb = rng.standard_normal((n + 37, 3))
    ca = rng.integers(0, 256, (n, 3)) / 255.0; cb = rng.integers(0, 256, (n + 37, 3)) / 255.0
    pair = CloudPair(PointCloud(a, na, ca), PointCloud(b, nb, cb), extent=[1, 1, 1], normal_index="neighbour",
                     _engine=OracleEngine())
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions("ycc", True, True))).as_dict()
    want = [[list(map(str, k)), [float(x).hex() for x in np.atleast_1d(v)]] for k, v in res.items()]
    assert any(k[0] == "ColorMSE" for k in res)                      # the colour rows are part of the comparison
    assert outs[0]["rows"] == want                                   # bit-identical to one process
    assert outs[0]["col_len"] == n + 37
    assert outs[0]["col_sum"] == float(np.sum(np.asarray(pair.get_right_neighbour_distances()))).hex()
    assert outs[0]["ev_sum"] == float(np.sum(np.asarray(pair.get_left_error_vector()))).hex()


Q1_WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PCCM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCCM_ROOT"], "tests"))
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle_engine import OracleEngine

dist.init_process_group("gloo")
n = int(os.environ["PCCM_N"])
rng = np.random.default_rng(7)
a = rng.random((n + 300, 3), dtype=np.float32); b = rng.random((n, 3), dtype=np.float32)     # A larger than B
na = rng.standard_normal((n + 300, 3)); nb = rng.standard_normal((n, 3))
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1], group=dist.group.WORLD,
                 shard_mode=os.environ["PCCM_MODE"], _engine=OracleEngine())
outcome = "no error"
try:
    MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, False, True)))   # default normal_index="row"
except IndexError as exc:
    outcome = "IndexError: " + str(exc)
with open(os.path.join(os.environ["PCCM_OUT"], f"q1_rank{dist.get_rank()}.json"), "w") as fh:
    json.dump({"rank": dist.get_rank(), "outcome": outcome, "shard": pair._engine.shard_range(0)}, fh)
dist.barrier()                      # (orderly teardown: a rank that leaves while a peer's gloo thread still talks to it aborts that peer)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["rows", "direction"])
def test_row_indexed_normals_out_of_range_raise_on_every_rank(tmp_path, mode):
    """Reference quirk Q1 (metric.py:148-152: normals_other[i] with i up to len(iterating cloud)) under sharding:
    only the LAST rank's shard reaches past the other cloud's normals, yet every rank must raise the reference's
    IndexError -- a per-shard decision would leave the low ranks waiting in the all-reduce.
    Split by direction, the rank that searches the other direction owns no row of the offending column at all and must
    raise just the same."""
    n = 3000
    script = tmp_path / "q1_worker.py"
    script.write_text(Q1_WORKER)
    env = dict(os.environ, PCCM_ROOT=ROOT, PCCM_N=str(n), MASTER_ADDR="127.0.0.1", PCCM_OUT=str(tmp_path), PCCM_MODE=mode)
    proc = _torchrun(script, 2, env, 300)                         # a hang would trip the timeout
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    outs = [json.load(open(tmp_path / f"q1_rank{r}.json")) for r in (0, 1)]
    if mode == "rows":
        assert outs[0]["shard"][1] <= n < outs[1]["shard"][1]      # rank 0's own rows are all in range, rank 1's are not
    else:
        assert outs[0]["shard"] == [0, n + 300] and outs[1]["shard"] == [0, 0]   # rank 1 owns no row of that column
    for o in outs:
        assert o["outcome"].startswith("IndexError"), o
