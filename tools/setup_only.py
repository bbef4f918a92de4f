import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
X, y, Z = synthetic_problem(100000, 8, 1024, 0)
h = trained_like_hypers(8)
ctx = HipContext(X, y, 1024, "rbf")
ctx.set_option("chol_mode", int(os.environ.get("CHOL", "1")))
if "AAT_BLOCK" in os.environ:
    ctx.set_option("aat_block", int(os.environ["AAT_BLOCK"]))
ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
for _ in range(3):
    ctx.setup()
torch.cuda.synchronize()
import time
for _ in range(3):
    t0 = time.perf_counter(); ctx.setup(); torch.cuda.synchronize(); print(f"setup {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
v = torch.zeros(100000, dtype=torch.float64, device=ctx.device)
for run_cg in (True, False):
    for _ in range(2):
        v.zero_()
        t0 = time.perf_counter(); r = ctx.objective_and_grad(v, run_cg); torch.cuda.synchronize()
        print(f"objective_and_grad run_cg={run_cg}: {1e3*(time.perf_counter()-t0):.2f} ms steps={r.steps}", flush=True)
