import os, sys, time, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=rank, world_size=world)
a = torch.full((64, 8192), float(rank), device=dev); b = torch.zeros_like(a)
peer = 1 - rank
try:
    ops = [dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, peer)]
    for w in dist.batch_isend_irecv(ops): w.wait()
    torch.cuda.synchronize()
    print(rank, "first exchange ok", float(b[0, 0]), flush=True)
    t0 = time.perf_counter()
    for _ in range(200):
        for w in dist.batch_isend_irecv(ops): w.wait()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(rank, f"cpu enqueue per exchange {1e6*(t1-t0)/200:.1f} us, total per exchange {1e6*(t2-t0)/200:.1f} us", flush=True)
except Exception as e:
    print(rank, "FAILED:", repr(e)[:300], flush=True)
dist.destroy_process_group()
