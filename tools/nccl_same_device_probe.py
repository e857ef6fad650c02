import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
t = torch.ones(4, device="cuda:0") * (rank + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "allreduce ->", t.tolist(), flush=True)
a = torch.full((8,), float(rank), device="cuda:0"); b = torch.empty(8, device="cuda:0")
peer = (rank + 1) % world
ops = [dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, (rank - 1) % world)]
for r in dist.batch_isend_irecv(ops): r.wait()
torch.cuda.synchronize()
print("rank", rank, "recv ->", b[0].item(), flush=True)
dist.destroy_process_group()
