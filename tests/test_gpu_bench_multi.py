"""bench.py's N > 1 path on the test box's single GPU:

* two ``gloo`` ranks (``python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --backend gloo``): the
  launch contract of the driver, the direction-first split, the barrier / max-over-ranks timing and the JSON line;
* RCCL (backend "nccl") with the one rank a single GPU allows: process-group initialisation on the device, the pinned +
  device staging of ``Collective.allreduce`` and a barrier -- the code ``bench.py --gpus N --backend nccl`` runs on every
  rank.  A real N-GPU curve can only be measured by the driver on a multi-GPU node (recorded as unmeasured here)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_two_gloo_ranks_print_one_valid_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--points", "300000", "--backend", "gloo"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2"))
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                          # rank 0 prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "strong" and line["unit"] == "Mpoints/s"
    assert line["value"] > 0 and abs(line["value"] - 2 * 300000 / (line["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * line["value"]
    assert line["config"]["sharding"] == "direction x2"
    ranks = line["per_rank_kernel_us_per_step"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert ranks[0]["rows"] == [[0, 300000], [0, 0]] and ranks[1]["rows"] == [[0, 0], [0, 300000]]
    assert all("grid_query" in r and "grid_build" in r for r in ranks)
    solo = line["independent_pairs"]                                # one whole pair per rank and step, no collective
    assert solo["scaling"] == "weak" and solo["pairs_per_step"] == 2
    assert abs(solo["value"] - 2 * 2 * 300000 / (solo["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * solo["value"]


def test_bench_launches_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts the ranks itself as a child process
    (before any GPU call) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--points", "200000",
           "--backend", "gloo", "--no-extras"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, OMP_NUM_THREADS="2"))
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo" and line["steps"] == 4
    assert line["config"]["sharding"] == "direction x2" and line["value"] > 0


NCCL_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PCCM_ROOT"])
import torch, torch.distributed as dist
from open_pcc_metric_amd.collective import Collective
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))        # what bench.py --backend nccl does per rank
coll = Collective(dist.group.WORLD)
coll.device = 0
assert coll._backend == "nccl" and coll.world == 1
coll.world = 2        # one GPU, one rank: pretend there is a peer so that the exchange code really runs (identity result)
for n in (10, 70000):
    x = np.random.default_rng(n).random(n)
    assert np.array_equal(coll.allreduce(x, "sum"), x)
    assert np.array_equal(coll.allreduce(x, "max"), x)
t = torch.tensor([3.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                       # the timing reduction of bench.py
dist.barrier()
assert float(t) == 3.0
dist.destroy_process_group()
print("NCCL-ONE-RANK-OK")
'''


def test_rccl_exchange_path_on_one_rank(tmp_path):
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, PCCM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0 and "NCCL-ONE-RANK-OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]
