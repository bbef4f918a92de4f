// Greedy inducing-point selection: the reference initialises Z with robustgp.ConditionalVariance(sample=False)
// (config.py:62-65; third-party, absent) = greedy maximisation of the conditional variance = pivoted Cholesky of K_ff.
//   d = diag K + jitter;  j_0 = argmax d
//   step m:  e = (K[:, j_m] + jitter 1_{j_m} - C[:m]^T C[:m, j_m]) / sqrt(d_{j_m});  C[m] = e;  d = max(d - e^2, 0);  d[j_m] = 0
//            j_{m+1} = argmax d   (lowest index on ties)
// O(N M^2) flops but HBM-bound: step m streams the m rows of C written so far (sum_m m N w bytes = 419 GB at N = 100k, M = 1024).
// Everything stays on the device: one streaming kernel + one single-block pivot kernel per step, no host synchronisation.
#include "devmath.h"
#include "dispatch.h"

struct SelPivot {
    long long j;   // current pivot
    double dj;     // sqrt(d[j])
};

// One thread per point.  cjv[t] = C[t][j] (gathered by the pivot kernel) is wave-uniform -> scalar loads.
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void select_step_kernel(const T* __restrict__ Xs, int64_t n, T* __restrict__ C, int m, T* __restrict__ d,
                                                          const SelPivot* __restrict__ piv, const T* __restrict__ cjv, T var, T jitter,
                                                          T* __restrict__ pval, long long* __restrict__ pidx, int Dw) {
    __shared__ T sval[4];
    __shared__ long long sidx[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = piv->j;
    const T dj = (T)piv->dj;
    T best = T(-1);
    long long bidx = 0x7fffffffffffffffLL;
    if (i < n) {
        T d2 = 0;
        if constexpr (DP == 0) {  // wide inputs: run-time width (rows are read sequentially: whole cache lines are used)
            for (int q = 0; q < Dw; ++q) {
                const T df = Xs[i * Dw + q] - Xs[j * Dw + q];
                d2 = tfma<T>(df, df, d2);
            }
        } else {
#pragma unroll
            for (int q = 0; q < DP; ++q) {
                const T df = Xs[i * DP + q] - Xs[j * DP + q];
                d2 = tfma<T>(df, df, d2);
            }
        }
        T col = var * kappa_from_d2<T, KIND>(d2);
        if (i == j) col += jitter;
        // dot = sum_{t<m} C[t][i] cjv[t]: 8 independent partial sums keep 8 coalesced row reads in flight per thread
        T acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int t = 0;
        for (; t + 8 <= m; t += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = tfma<T>(C[(int64_t)(t + u) * n + i], cjv[t + u], acc[u]);
        }
        for (int u = 0; t < m; ++t, ++u) acc[u] = tfma<T>(C[(int64_t)t * n + i], cjv[t], acc[u]);
        const T dot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        const T e = (col - dot) / dj;
        C[(int64_t)m * n + i] = e;
        T dn = d[i] - e * e;
        dn = dn > T(0) ? dn : T(0);
        if (i == j) dn = T(0);
        d[i] = dn;
        best = dn;
        bidx = i;
    }
    // block argmax, lowest index on ties
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_xor(best, off, 64);
        const long long oi = __shfl_xor(bidx, off, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = best; sidx[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < bidx)) { best = sval[w]; bidx = sidx[w]; }
        pval[blockIdx.x] = best;
        pidx[blockIdx.x] = bidx;
    }
}

// d = var + jitter, per-block argmax partials for the first pivot
template <typename T>
__global__ __launch_bounds__(256) void select_init_kernel(int64_t n, T var, T jitter, T* __restrict__ d, T* __restrict__ pval,
                                                          long long* __restrict__ pidx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = var + jitter;
    if (threadIdx.x == 0) {  // all entries equal: the block's lowest index wins
        pval[blockIdx.x] = var + jitter;
        pidx[blockIdx.x] = (long long)blockIdx.x * blockDim.x;
    }
}

// Single block: next pivot from the partials, its sqrt(d), and the gathered column cjv[t] = C[t][j], t < rows.
template <typename T>
__global__ __launch_bounds__(256) void select_pivot_kernel(const T* __restrict__ pval, const long long* __restrict__ pidx, int nparts,
                                                           const T* __restrict__ C, int64_t n, int rows, SelPivot* __restrict__ piv,
                                                           T* __restrict__ cjv, long long* __restrict__ chosen, int slot) {
    __shared__ T sval[4];
    __shared__ long long sidx[4];
    __shared__ long long jsh;
    T best = T(-1);
    long long bidx = 0x7fffffffffffffffLL;
    for (int q = threadIdx.x; q < nparts; q += blockDim.x) {
        const T v = pval[q];
        const long long ix = pidx[q];
        if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_xor(best, off, 64);
        const long long oi = __shfl_xor(bidx, off, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = best; sidx[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < bidx)) { best = sval[w]; bidx = sidx[w]; }
        jsh = bidx;
        piv->j = bidx;
        piv->dj = best > T(0) ? sqrt((double)best) : 1.0;  // variance exhausted (fewer distinct points than m): e = 0, no NaN
        chosen[slot] = bidx;
    }
    __syncthreads();
    const long long j = jsh;
    for (int t = threadIdx.x; t < rows; t += blockDim.x) cjv[t] = C[(int64_t)t * n + j];
}

template <typename T, int DP>
__global__ __launch_bounds__(256) void select_gather_kernel(const T* __restrict__ X, int D, const long long* __restrict__ chosen, int M,
                                                            T* __restrict__ Z) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < M * D) Z[idx] = X[chosen[idx / D] * D + idx % D];
}

template <typename T>
__global__ __launch_bounds__(256) void select_trace_kernel(const T* __restrict__ d, int64_t n, double* __restrict__ out) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)d[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[0] = s;
}

template <typename T, int KIND, int DP>
static int select_impl(cglb_ctx* c, double variance, double jitter, long long* chosen_dev, void* Z_out, double* trace_dev) {
    const int64_t n = c->N;
    const int M = (int)(c->M < n ? c->M : n);
    const int nparts = (int)((n + 255) / 256);
    T *C = nullptr, *d = nullptr, *cjv = nullptr, *pval = nullptr;
    long long* pidx = nullptr;
    SelPivot* piv = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(C); (void)hipFree(d); (void)hipFree(cjv); (void)hipFree(pval); (void)hipFree(pidx); (void)hipFree(piv);
    };
#define SEL_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) { cleanup(); return cglb_fail(c, CGLB_ERR_HIP, hipGetErrorString(_e)); } \
    } while (0)
    SEL_HIP(hipMalloc(&C, (size_t)M * n * sizeof(T)));
    SEL_HIP(hipMalloc(&d, (size_t)n * sizeof(T)));
    SEL_HIP(hipMalloc(&cjv, (size_t)M * sizeof(T)));
    SEL_HIP(hipMalloc(&pval, (size_t)nparts * sizeof(T)));
    SEL_HIP(hipMalloc(&pidx, (size_t)nparts * sizeof(long long)));
    SEL_HIP(hipMalloc(&piv, sizeof(SelPivot)));
    hipLaunchKernelGGL((select_init_kernel<T>), dim3(nparts), dim3(256), 0, c->stream, n, (T)variance, (T)jitter, d, pval, pidx);
    hipLaunchKernelGGL((select_pivot_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)pval, (const long long*)pidx, nparts, (const T*)C, n,
                       0, piv, cjv, chosen_dev, 0);
    for (int m = 0; m < M; ++m) {  // the step of the last pivot only serves the reported trace (all M points conditioned on)
        hipLaunchKernelGGL((select_step_kernel<T, KIND, DP>), dim3(nparts), dim3(256), 0, c->stream, (const T*)c->Xs, n, C, m, d,
                           (const SelPivot*)piv, (const T*)cjv, (T)variance, (T)jitter, pval, pidx, c->D);
        if (m + 1 < M)
            hipLaunchKernelGGL((select_pivot_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)pval, (const long long*)pidx, nparts,
                               (const T*)C, n, m + 1, piv, cjv, chosen_dev, m + 1);
    }
    if (Z_out)
        hipLaunchKernelGGL((select_gather_kernel<T, DP>), dim3((unsigned)((M * c->D + 255) / 256)), dim3(256), 0, c->stream, (const T*)c->X, c->D,
                           (const long long*)chosen_dev, M, (T*)Z_out);
    hipLaunchKernelGGL((select_trace_kernel<T>), dim3(1), dim3(256), 0, c->stream, (const T*)d, n, trace_dev);
    SEL_HIP(hipGetLastError());
    SEL_HIP(hipStreamSynchronize(c->stream));
#undef SEL_HIP
    cleanup();
    return CGLB_OK;
}

int launch_select_inducing(cglb_ctx* c, double variance, double jitter, long long* chosen_dev, void* Z_out, double* trace_dev) {
    if (is_wide(c)) { CGLB_DISPATCH_T(c->dtype, CGLB_DISPATCH_KIND(c->kind, return (select_impl<T, KIND, 0>(c, variance, jitter, chosen_dev, Z_out, trace_dev)))); }
    CGLB_DISPATCH_ALL(c, return (select_impl<T, KIND, DP>(c, variance, jitter, chosen_dev, Z_out, trace_dev)));
    return CGLB_OK;
}
