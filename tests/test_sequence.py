"""evaluate_pairs: pairs are the unit of work; ranks take every W-th pair and gather the finished rows."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PCCM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCCM_ROOT"], "tests"))
from open_pcc_metric_amd.options import CalculateOptions
from open_pcc_metric_amd.point_cloud import PointCloud
from open_pcc_metric_amd.sequence import evaluate_pairs
from oracle_engine import OracleEngine
dist.init_process_group("gloo")
loaded = []
def frame(k):
    def load():
        loaded.append(k)
        rng = np.random.default_rng(100 + k)
        a, b = rng.random((400 + 7 * k, 3)), rng.random((400 + 7 * k, 3))
        return PointCloud(a, rng.standard_normal(a.shape)), PointCloud(b, rng.standard_normal(b.shape))
    return load
import open_pcc_metric_amd.sequence as seq
_CloudPair = seq.CloudPair
seq.CloudPair = lambda o, r, device=None, **kw: _CloudPair(o, r, _engine=OracleEngine(), **kw)
rows = evaluate_pairs([frame(k) for k in range(5)], CalculateOptions(None, True, True), group=True, extent=[1, 1, 1])
out = {"rank": dist.get_rank(), "loaded": loaded,
       "rows": [[[list(map(str, k)), float(v).hex()] for k, v in r.items()] for r in rows]}
with open(os.path.join(os.environ["PCCM_OUT"], f"rank{dist.get_rank()}.json"), "w") as fh:
    json.dump(out, fh)
dist.barrier()                      # (orderly teardown: a rank that leaves while a peer's gloo thread still talks to it aborts that peer)
dist.destroy_process_group()
'''


def test_pairs_are_split_over_ranks_and_gathered(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from open_pcc_metric_amd.calculator import MetricCalculator
    from open_pcc_metric_amd.cloud_pair import CloudPair
    from open_pcc_metric_amd.options import CalculateOptions, transform_options
    from open_pcc_metric_amd.point_cloud import PointCloud
    from oracle_engine import OracleEngine

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCCM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", PCCM_OUT=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    outs = sorted((json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1)), key=lambda o: o["rank"])
    assert sorted(outs[0]["loaded"]) == [0, 2, 4] and sorted(outs[1]["loaded"]) == [1, 3]   # only the rank's own share is loaded
    assert outs[0]["rows"] == outs[1]["rows"] and len(outs[0]["rows"]) == 5     # every rank holds all reports, in order
    opts = transform_options(CalculateOptions(None, True, True))
    for k in range(5):
        rng = np.random.default_rng(100 + k)
        a, b = rng.random((400 + 7 * k, 3)), rng.random((400 + 7 * k, 3))
        pa, pb = PointCloud(a, rng.standard_normal(a.shape)), PointCloud(b, rng.standard_normal(b.shape))
        pair = CloudPair(pa, pb, extent=[1, 1, 1], _engine=OracleEngine())
        want = [[list(map(str, key)), float(v).hex()] for key, v in MetricCalculator(pair).calculate(opts).as_dict().items()]
        assert outs[0]["rows"][k] == want


@pytest.mark.gpu
def test_evaluate_pairs_on_one_gpu_matches_single_reports():
    from open_pcc_metric_amd.calculator import MetricCalculator
    from open_pcc_metric_amd.cloud_pair import CloudPair
    from open_pcc_metric_amd.options import CalculateOptions, transform_options
    from open_pcc_metric_amd.point_cloud import PointCloud
    from open_pcc_metric_amd.sequence import evaluate_pairs
    rng = np.random.default_rng(3)
    frames = []
    for k in range(4):
        a = rng.random((20000 + 100 * k, 3), dtype=np.float32)
        b = (a + rng.normal(0, 1e-3, a.shape)).astype(np.float32)
        frames.append((PointCloud(a, rng.standard_normal(a.shape)), PointCloud(b, rng.standard_normal(b.shape))))
    options = CalculateOptions(None, True, True)
    rows = evaluate_pairs(frames, options, extent=[1, 1, 1])
    for (pa, pb), got in zip(frames, rows):
        with CloudPair(pa, pb, extent=[1, 1, 1]) as pair:
            want = MetricCalculator(pair).calculate(transform_options(options)).as_dict()
        assert list(got) == list(want) and all(np.float64(got[k]) == np.float64(want[k]) for k in want)
