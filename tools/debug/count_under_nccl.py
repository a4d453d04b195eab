"""Why is s2k_count_device slower in a process that initialised RCCL?  Times the same call before / after init_process_group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from s2k_loader import import_package
pkg = import_package()
eng = pkg.Engine(0)
dev = torch.device("cuda", 0)
n = 140_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
keys = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
def t(tag):
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nd = eng.count_device(keys.data_ptr(), n)
        torch.cuda.synchronize()
        print(tag, i, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), nd, flush=True)
t("before init")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
t("after init_process_group")
x = torch.ones(1, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
t("after first all_reduce")
dist.barrier(); torch.cuda.synchronize()
t("after barrier")
dist.destroy_process_group()
