"""One-rank RCCL probe: process group on GPU 0, one all-reduce, one barrier (tests/test_gpu_run.py
uses it to tell a box without a working RCCL from a defect of this package)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
t0=time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
x=torch.arange(1<<20, dtype=torch.float32, device="cuda")
y=x.clone()
w=dist.all_reduce(y, async_op=True); w.wait(); torch.cuda.synchronize()
print("allreduce ok", torch.equal(x,y), "t=%.1fs"%(time.time()-t0))
dist.barrier(); torch.cuda.synchronize()
print("barrier ok t=%.1fs"%(time.time()-t0))
print("nccl version", torch.cuda.nccl.version())
dist.destroy_process_group()
