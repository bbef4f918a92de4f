"""Developer tool: which rocBLAS kernels serve a triangular panel solve of the setup's shape when issued through torch
(out-of-place, its own handle/workspace) - to compare with the library's own rocblas_dtrsm call."""
import torch, time
M, N = 1024, 100000
torch.manual_seed(0)
L = torch.tril(torch.randn(M, M, dtype=torch.float64, device="cuda")) * 0.01 + torch.eye(M, dtype=torch.float64, device="cuda")
B = torch.randn(M, N, dtype=torch.float64, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("row-major panel (M,N) contiguous: %.2f ms" % t(lambda: torch.linalg.solve_triangular(L, B, upper=False)))
