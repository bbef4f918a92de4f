"""Developer tool: A/B several builds of libcglb_hip.so (precision/speed knobs of the pair kernels, cglb_amd/csrc/Makefile
EXTRA_DEFS) on one box: pair-kernel time of the K_ff mat-vec and of the gradient pass, and the deviation of the mat-vec from the
first library given (the reference build) and from the blocked C oracle on a row sample.

  tools/build_variant.sh d3 "-DCGLB_EXP_DEG=3"         # one object directory per variant, built from clean
  python tools/k1_variants.py cglb_amd/lib/libcglb_hip.so cglb_amd/lib/variants/libcglb_d3.so ...

Variant libraries must come from tools/build_variant.sh (it leaves libcglb_NAME.defs beside the library): a variant linked from
an object directory shared with other define sets can mix objects - round 2 lost a variant to a GPU memory access fault that way
(profiles/r02_k1_variants.log).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from cglb_amd.data import synthetic_problem, trained_like_hypers
from cglb_amd.hip_context import HipContext
N = int(os.environ.get("N", 100000)); D = int(os.environ.get("D", 8)); M = 64
X, y, Z = synthetic_problem(N, D, M, 0)
h = trained_like_hypers(D)
out = {"lib": os.path.basename(os.environ["CGLB_HIP_LIB"])}
g = torch.Generator().manual_seed(5)
p = torch.randn(N, dtype=torch.float64, generator=g)
for kind in ("rbf", "matern32"):
    ctx = HipContext(X, y, M, kind)
    ctx.set_hypers(h["lengthscales"], h["variance"], h["noise"], h["mean"], Z, 1e-6)
    ctx.setup()
    Ap = ctx.matvec(p.to(ctx.device)).cpu().numpy()
    np.save(os.path.join(os.environ["OUTDIR"], out["lib"] + "." + kind + ".npy"), Ap)
    ms = [ctx.time_kernel(3, 10) for _ in range(3)]
    mg = [ctx.time_kernel(2, 5) for _ in range(2)]
    mv = [ctx.time_kernel(0, 10) for _ in range(3)]
    out[kind] = {"k1_ms": min(ms), "k1_ms_all": ms, "grad_ms": min(mg), "matvec_ms": min(mv)}
    ctx.close()
print("RESULT " + json.dumps(out))
"""


def main():
    libs = [os.path.abspath(a) for a in sys.argv[1:]]
    for lib in libs:
        name = os.path.basename(lib)
        if name == "libcglb_hip.so":
            continue  # the product build (cglb_amd/csrc/Makefile defaults, its own object directory)
        if not os.path.exists(lib[:-len(".so")] + ".defs"):
            raise SystemExit(f"{name}: no {name[:-3]}.defs beside it - build variants with tools/build_variant.sh (clean per-variant object directory)")
    outdir = os.path.join(ROOT, "gpurun_out", "k1_variants")
    os.makedirs(outdir, exist_ok=True)
    rounds = int(os.environ.get("ROUNDS", 2))
    results = {}
    for rnd in range(rounds):
        for lib in libs:
            env = dict(os.environ, CGLB_HIP_LIB=lib, OUTDIR=outdir)
            res = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(f"{lib}: FAILED\n{res.stdout[-2000:]}\n{res.stderr[-2000:]}", flush=True)
                continue
            r = json.loads(line[0][7:])
            results.setdefault(r["lib"], []).append(r)
            print(f"round {rnd} {r['lib']:28s} rbf K1 {r['rbf']['k1_ms']:.3f} ms mat-vec {r['rbf']['matvec_ms']:.3f} grad {r['rbf']['grad_ms']:.3f} | "
                  f"matern32 K1 {r['matern32']['k1_ms']:.3f} ms mat-vec {r['matern32']['matvec_ms']:.3f} grad {r['matern32']['grad_ms']:.3f}", flush=True)
    # accuracy: against the first library and against the blocked C oracle on a row sample
    import numpy as np
    import torch
    from oracle import cglb_oracle as orc
    from oracle import cglb_oracle_c as orcc
    N = int(os.environ.get("N", 100000)); D = int(os.environ.get("D", 8))
    X, y, Z = orc.synthetic_problem(N, D, 64, 0)
    hyp = orc.trained_like_hypers(D, Z)
    g = torch.Generator().manual_seed(5)
    p = torch.randn(N, dtype=torch.float64, generator=g).numpy()
    rows = slice(40_000, 40_000 + 256)
    summary = {}
    for kind in ("rbf", "matern32"):
        ref = orcc.kff_matvec(kind, X, hyp, p, rows.start, rows.stop)
        base = np.load(os.path.join(outdir, os.path.basename(libs[0]) + "." + kind + ".npy"))
        for lib in libs:
            name = os.path.basename(lib)
            if not os.path.exists(os.path.join(outdir, name + "." + kind + ".npy")):
                continue  # that build failed above
            a = np.load(os.path.join(outdir, name + "." + kind + ".npy"))
            d_base = float(np.abs(a - base).max() / np.abs(base).max())
            d_orc = float(np.abs(a[rows] - ref).max() / np.abs(base).max())
            summary.setdefault(name, {})[kind] = {"vs_first_lib": d_base, "vs_c_oracle_rows": d_orc,
                                                 "k1_ms": min(r[kind]["k1_ms"] for r in results.get(name, [{kind: {"k1_ms": float("nan")}}])),
                                                 "grad_ms": min(r[kind]["grad_ms"] for r in results.get(name, [{kind: {"grad_ms": float("nan")}}]))}
            print(f"{name:28s} {kind:9s} max|dAp|/max|Ap| vs first {d_base:.2e}  vs C oracle (256 rows) {d_orc:.2e}", flush=True)
    json.dump(summary, open(os.path.join(outdir, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
