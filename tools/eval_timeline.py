"""Developer tool: kernel timeline of ONE steady-state objective+gradient evaluation out of a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/train_loop_timing.py` writes DIR/NAME_results.db).
Usage: python tools/eval_timeline.py DIR/NAME_results.db [index of the evaluation from the end, default 2] [--full]
Consecutive launches of one family (rocBLAS GEMMs, Cholesky steps, PCG iterations) are merged into one line."""
import re, sqlite3, sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
full = "--full" in sys.argv
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "kuu_kernel" in r[0]]   # one per cglb_setup
a, b = starts[-back - 1], starts[-back]
ev = rows[a - 3:b - 3]   # an evaluation starts three small kernels (operand scaling) before kuu_kernel
t0 = ev[0][1]
print(f"{len(ev)} launches, span {(ev[-1][2] - t0) / 1e6:.3f} ms, busy {sum(r[2] - r[1] for r in ev) / 1e6:.3f} ms")


def family(n):
    if n.startswith("Cijk") or "rocblas" in n or "rocsolver" in n: return "rocBLAS/rocSOLVER"
    if "chol_" in n: return "cholesky"
    if re.search(r"kff_sym|gemv_|tri_rowdot|precond_z|update_|finalize|weight_operand|residual|copyBuffer|sub_scalar|dot_kernel", n): return "pcg / vector"
    return None


groups = []
for n, s, e in ev:
    f = None if full else family(n)
    n = n.replace("void ", "")
    if f and groups and groups[-1][0] == f:
        g = groups[-1]; g[2] = e; g[3] += e - s; g[4] += 1
    else:
        groups.append([f, s, e, e - s, 1, n])
for f, s, e, busy, cnt, n in groups:
    label = f"[{f}] {cnt} launches" if f and cnt > 1 else n[:100]
    print(f"{(s - t0) / 1e3:10.1f} us  span {(e - s) / 1e3:8.1f}  busy {busy / 1e3:8.1f}  {label}")
