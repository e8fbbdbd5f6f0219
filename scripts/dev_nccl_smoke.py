"""Developer scratch: one-rank RCCL sanity (init, all_reduce, barrier, a sharded report) next to a libpccm context."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from open_pcc_metric_amd.collective import Collective
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
e = nat.Engine(0)
c = Collective(dist.group.WORLD)
c.world = 2          # pretend, so that the collective code path really runs (single rank: identity results)
x = np.arange(10, dtype=np.float64)
print("sum", c.allreduce(x, "sum")[:3], "max", c.allreduce(np.array([1.0, -2.0]), "max"))
for n in (10, 270000):
    x = np.random.default_rng(n).random(n)
    for _ in range(3):
        t = time.perf_counter(); y = c.allreduce(x, "sum"); dt = time.perf_counter() - t
    print("allreduce", n, "doubles:", round(dt * 1e6, 1), "us", bool((x == y).all()))
dist.barrier()
# the sharded report path with a world of one pretending to be sharded: exchange = identity, results = unsharded
rng = np.random.default_rng(1)
a = rng.random((100000, 3), dtype=np.float32); b = rng.random((100000, 3), dtype=np.float32)
na = rng.standard_normal((100000, 3)); nb = rng.standard_normal((100000, 3))
opts = transform_options(CalculateOptions(None, True, True))
ref = MetricCalculator(CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])).calculate(opts).as_dict()
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1], group=dist.group.WORLD)
pair._coll.world = 2     # rank 0 of a pretended 2: the engine still holds the whole cloud (set_shard was (0, 1))
got = MetricCalculator(pair).calculate(opts).as_dict()
diff = [k for k in ref if not np.array_equal(np.atleast_1d(ref[k]), np.atleast_1d(got[k]))]
# the pretended second rank contributes all-zero extrema slots, so only Hausdorff-type rows (a max against 0) may differ
print("sharded-path rows that differ (expected: none with 'MSE' or plain 'GeoPSNR'):", diff)
t = torch.tensor([1.0], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("ok", float(t))
dist.destroy_process_group()
