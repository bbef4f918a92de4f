import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.multiprocessing as mp

def worker(rank, world, port, N, which, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cglb_amd.distributed import Comm, HipSymLocalOps, HipLocalOps, SymShardedCGLB, ShardedCGLB, row_partition
    from cglb_amd.hip_context import HipContext
    from cglb_amd.data import synthetic_problem, trained_like_hypers
    torch.cuda.set_device(0)
    X, y, Z = synthetic_problem(N, 8, 32, seed=7)
    h = trained_like_hypers(8)
    per, parts = row_partition(N, world)
    ctx = HipContext(X, y, 32, "rbf", row_range=parts[rank])
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    if which == "sym":
        drv = SymShardedCGLB(HipSymLocalOps(ctx), Comm())
    else:
        drv = ShardedCGLB(HipLocalOps(ctx), Comm())
    res = drv.objective_and_grad(True, 1.0, 100, 40)
    if rank == 0:
        q.put((which, res.bound, res.lower, res.upper, res.logdet, res.steps, res.residual_error))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    N = int(sys.argv[1]); world = int(sys.argv[2])
    from cglb_amd.hip_context import HipContext
    from cglb_amd.data import synthetic_problem, trained_like_hypers
    X, y, Z = synthetic_problem(N, 8, 32, seed=7)
    h = trained_like_hypers(8)
    ctx = HipContext(X, y, 32, "rbf")
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    v = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    f = ctx.objective_and_grad(v, True, 1.0, 100, 40)
    print("fused", f.bound, f.lower, f.upper, f.logdet, f.steps, f.residual_error)
    for which in ("row", "sym"):
        c = mp.get_context("spawn"); q = c.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [c.Process(target=worker, args=(r, world, port, N, which, q)) for r in range(world)]
        [p.start() for p in ps]
        print(q.get(timeout=300))
        [p.join() for p in ps]
