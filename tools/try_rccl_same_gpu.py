import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce ok {t.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: RCCL on a shared device failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
