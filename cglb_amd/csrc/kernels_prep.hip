// Operand preparation and the dense kernel blocks of the common terms
// (reference: models.py:196-201 — K_uf, K_uu + jitter I).
#include "devmath.h"
#include "dispatch.h"

// xs[i][d] = (x[i][d] - c_d) * scale_d (zero padded to DP), xa[i] = RBF: -|xs|^2/2, Matern32: |xs|^2
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void prep_scaled_kernel(const T* __restrict__ X, int64_t n, int D, ScaleParams sp,
                                                          T* __restrict__ Xs, T* __restrict__ xa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s2 = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        T v = 0;
        if (d < D) v = (T)(((double)X[i * D + d] - sp.center[d]) * sp.scale[d]);
        Xs[i * DP + d] = v;
        s2 = tfma<T>(v, v, s2);
    }
    xa[i] = (KIND == CGLB_RBF) ? T(-0.5) * s2 : s2;
}

// hot operand set: RBF exponent -|xh_i-xh_j|^2/2 and Matern exponent -2|xh_i-xh_j| are in 1/T octave (T = 2^CGLB_TAB_BITS);
// Matern-3/2 carries HALF the nominal scale because the hot-loop square root returns 2 sqrt (devmath.h: sqrt_hot)
double cglb_hot_scale(const cglb_ctx* c) { return (c->kind == CGLB_RBF) ? sqrt(CGLB_HOT_UNITS) : 0.5 * CGLB_HOT_UNITS; }

int launch_prep_scaled(cglb_ctx* c, const void* Xraw, int64_t n, void* Xs_out, void* xa_out, bool hot) {
    if (is_wide(c)) return wide_prep_scaled(c, Xraw, n, Xs_out, xa_out, nullptr);  // one operand set: the wide path has no "hot" units
    ScaleParams sp;
    double kscale = (c->kind == CGLB_RBF) ? sqrt(CGLB_LOG2E) : CGLB_SQRT3 * CGLB_LOG2E;
    if (hot) kscale *= cglb_hot_scale(c);
    for (int d = 0; d < CGLB_MAX_D_NARROW; ++d) {
        sp.center[d] = d < c->D ? c->xmean[d] : 0.0;
        sp.scale[d] = d < c->D ? kscale / c->ls[d] : 0.0;
    }
    if (n == 0) return CGLB_OK;
    const int grid = (int)((n + 255) / 256);
    CGLB_DISPATCH_ALL(c, hipLaunchKernelGGL((prep_scaled_kernel<T, KIND, DP>), dim3(grid), dim3(256), 0, c->stream,
                                            (const T*)Xraw, n, c->D, sp, (T*)Xs_out, (T*)xa_out));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Kuf block: out[m * nloc + n] = var * kappa(z_m, x_{r0+n}), direct differences on the scaled operands
// (exact zero on coincident points).  Thread = one column n, loops over a chunk of rows m; z_m is
// wave-uniform (scalar loads), stores are coalesced along n.
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void kuf_kernel(const T* __restrict__ Zs, const T* __restrict__ Xs, int64_t r0,
                                                  int64_t nloc, int64_t lda, int M, int mchunk, T var, T* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m0 = blockIdx.y * mchunk;
    const int m1 = min(M, m0 + mchunk);
    const int64_t nn = n < nloc ? n : nloc - 1;
    T x[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = Xs[(r0 + nn) * DP + d];
    for (int m = m0; m < m1; ++m) {
        T d2 = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const T df = Zs[(int64_t)m * DP + d] - x[d];
            d2 = tfma<T>(df, df, d2);
        }
        const T k = var * kappa_from_d2<T, KIND>(d2);
        if (n < nloc) out[(int64_t)m * lda + n] = k;
    }
}

int launch_kuf(cglb_ctx* c) {
    if (c->nloc == 0) return CGLB_OK;
    if (is_wide(c)) return wide_kuf(c);
    const int mchunk = 32;
    dim3 grid((unsigned)((c->nloc + 255) / 256), (unsigned)((c->M + mchunk - 1) / mchunk));
    CGLB_DISPATCH_ALL(c, hipLaunchKernelGGL((kuf_kernel<T, KIND, DP>), grid, dim3(256), 0, c->stream, (const T*)c->Zs,
                                            (const T*)c->Xs, c->r0, c->nloc, c->lda, c->M, mchunk, (T)c->var, (T*)c->At));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Kuu + jitter I, full symmetric M x M.
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void kuu_kernel(const T* __restrict__ Zs, int M, T var, T jitter, T* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * M) return;
    const int i = (int)(idx / M), j = (int)(idx % M);
    T d2 = 0;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        const T df = Zs[(int64_t)i * DP + d] - Zs[(int64_t)j * DP + d];
        d2 = tfma<T>(df, df, d2);
    }
    T k = var * kappa_from_d2<T, KIND>(d2);
    if (i == j) k += jitter;
    out[idx] = k;
}

int launch_kuu(cglb_ctx* c) {
    if (is_wide(c)) return wide_kuu(c);
    const int64_t tot = (int64_t)c->M * c->M;
    const int grid = (int)((tot + 255) / 256);
    CGLB_DISPATCH_ALL(c, hipLaunchKernelGGL((kuu_kernel<T, KIND, DP>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->Zs,
                                            c->M, (T)c->var, (T)c->jitter, (T*)c->Lc));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}
