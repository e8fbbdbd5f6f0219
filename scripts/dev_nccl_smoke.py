"""Developer scratch: one-rank RCCL sanity (init, all_reduce, barrier) next to a libpccm context."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from open_pcc_metric_amd.collective import Collective
from open_pcc_metric_amd import _native as nat
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
e = nat.Engine(0)
c = Collective(dist.group.WORLD)
c.world = 2          # pretend, so that the collective code path really runs (single rank: identity results)
x = np.arange(10, dtype=np.float64)
print("sum", c.allreduce(x, "sum")[:3], "max", c.allreduce(np.array([1.0, -2.0]), "max"))
print("gather", c.allgather_rows(np.ones((3, 2)), [3, 3]).shape if False else "skipped")
dist.barrier()
t = torch.tensor([1.0], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("ok", float(t))
dist.destroy_process_group()
