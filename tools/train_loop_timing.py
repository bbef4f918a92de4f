"""Developer tool: end-to-end cost of the training loop through the cglb.backend mirror at the headline shape:
create_model (GPU inducing-point selection) + optimize (SciPy L-BFGS-B over the HIP objective) for a few steps, against the
bare evaluation rate of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cglb_amd.backend import BACKENDS, CGLBConfig, INDUCING_VARIABLE_CONFIGS, KERNEL_CONFIGS
from cglb_amd.backend.callbacks import Logger
from cglb_amd.data import synthetic_problem

N, D, M = int(os.environ.get("N", 100000)), int(os.environ.get("D", 8)), int(os.environ.get("M", 1024))
steps = int(os.environ.get("STEPS", 6))
X, y, _ = synthetic_problem(N, D, 8, 0)
be = BACKENDS["hip"]
be.configure_backend(logdir="/tmp/cglb_train_timing", keops=False)
be.set_default_float("fp64"); be.set_default_jitter("fp64")
t0 = time.perf_counter()
model = be.create_model(CGLBConfig(kernel=KERNEL_CONFIGS[os.environ.get("KERNEL", "rbf")](), inducing_variable=INDUCING_VARIABLE_CONFIGS["cv"](M)), (X, y))
torch.cuda.synchronize(); t_create = time.perf_counter() - t0
data = ((X, y), (X[:1000], y[:1000]))
logger = Logger("/tmp/cglb_train_timing", be.metrics_fn(model, data), lambda: be.model_parameters(model), holdout_interval=10**9, verbose=False)
# wall time of every objective+gradient call inside optimize (the first ones carry one-time costs: rocBLAS kernels are loaded lazily)
eval_ms, eval_steps = [], []
_orig = model.hip.objective_and_grad
def _timed(*a, **k):
    t = time.perf_counter(); r = _orig(*a, **k); eval_ms.append(1e3 * (time.perf_counter() - t)); eval_steps.append(r.steps); return r
model.hip.objective_and_grad = _timed
prof = None
if os.environ.get("PROFILE"):  # host-side view: where the wall time of an evaluation goes (C-ABI calls show as _FuncPtr entries)
    import cProfile
    prof = cProfile.Profile(); prof.enable()
t0 = time.perf_counter()
results = be.optimize(model, data, steps, logger, "scipy")
torch.cuda.synchronize(); t_opt = time.perf_counter() - t0
if prof is not None:
    import pstats
    prof.disable(); pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
nfev = sum(r.nfev for r in results); nit = sum(r.nit for r in results)
print(f"N={N} D={D} M={M}: create_model {t_create:.2f} s (incl. GPU inducing-point selection); optimize {nit} iterations / {nfev} evaluations "
      f"in {t_opt:.2f} s = {1e3*t_opt/max(nfev,1):.1f} ms per evaluation (median of the C-ABI call alone {np.median(eval_ms):.1f} ms, "
      f"first call {eval_ms[0]:.0f} ms, mean CG steps {np.mean(eval_steps):.1f}); "
      f"final loss {results[-1].fun:.4f}", flush=True)
