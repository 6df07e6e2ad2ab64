"""Exercises dm-vio_amd/sharding.Collective on the RCCL backend with a single rank (device tensors, pinned staging, all_gather_into_tensor)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as graft
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
graft.load_package()
import dmvio_amd.sharding as sh
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
c = sh.Collective(dist, dev)
c.world = 2   # force the multi-rank code path: with world == 1 the wrapper short-circuits to identities
x = np.arange(10, dtype=np.float64)
try:
    a = c.allreduce_sum(x)
    assert np.allclose(a, x), a
    host, devb = c._buf("ag_in", 5, torch.float64)
    ohost, odev = c._buf("ag_out", 5, torch.float64)
    host.numpy()[:] = np.arange(5); devb.copy_(host)
    dist.all_gather_into_tensor(odev, devb); ohost.copy_(odev)
    assert np.allclose(ohost.numpy(), np.arange(5))
    m = c.allreduce_max(3.0)
    assert m == 3.0
    print("RCCL single-rank exchange path OK")
finally:
    dist.destroy_process_group()
