import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
g = torch.arange(1<<20, dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.mul_(2)
    w = dist.all_reduce(g[1000:500000], op=dist.ReduceOp.AVG, async_op=True)
w.wait(); torch.cuda.synchronize()
print("ok", float(g[1000]), float(g[999]), dist.get_backend())
dist.destroy_process_group()
