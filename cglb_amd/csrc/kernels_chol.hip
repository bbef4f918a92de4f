// Blocked lower Cholesky of the two small M x M matrices of the common terms (K_uu + jitter I and B = A A^T + I,
// reference models.py:202, :210).  rocSOLVER's potrf spends 2.9 ms on a 1024^2 matrix (latency of ~40 small kernels with a
// 155-us unblocked panel step).  This right-looking version runs three small kernels per 64-column step.  They are written as
// ROLLED loops on purpose: a one-wave, fully unrolled factorisation of the 64 x 64 block (rows in registers, operands by
// v_readlane or LDS broadcast) is 50-60 KB of straight-line code that runs once on a cold instruction cache, and measured
// 38 us (readlane form) / 90 us (LDS form) - the time of fetching the instructions, not of executing them.
//   diag    64 x 64 diagonal block, 512 threads: lane = row, wave = column group; one barrier per column
//   panel   rows below it, one wave per row (lane = column), the finished entry broadcast with v_readlane (no barrier)
//   update  trailing rank-64 update A22 -= P P^T, one 64 x 64 tile of the lower triangle per workgroup
//           (rocBLAS syrk splits this shape into a diagonal kernel and 5-7 small GEMMs: ~45 us of launches per step).
// Column-major, in place; the strict upper triangle is not referenced or modified.
#include "devmath.h"
#include "dispatch.h"

#ifndef CHOL_NB
#define CHOL_NB 64
#endif
static_assert(CHOL_NB == 64, "the kernels map the 64 rows / columns of a block to the lanes of a wave");

// sqrt(s) and 1/sqrt(s) of a positive pivot: v_rsq_f64 seed, two Newton steps, one residual correction each (both to ~1 ulp)
__device__ __forceinline__ void pivot_root(double s, double& root, double& rinv) {
    double y = __builtin_amdgcn_rsq(s);
    const double h = 0.5 * s;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-(h * y), y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    double g = s * y;
    g = __builtin_fma(__builtin_fma(-g, g, s), 0.5 * y, g);
    root = g;
    rinv = __builtin_fma(__builtin_fma(-g, y, 1.0), y, y);
}
__device__ __forceinline__ void pivot_root(float s, float& root, float& rinv) {
    root = __builtin_sqrtf(s);
    rinv = 1.0f / root;
}
// 1/s for the rank-1 update (the only arithmetic between reading a pivot and the next column): v_rcp_f64 + two Newton steps
__device__ __forceinline__ double pivot_rcp(double s) {
    double y = __builtin_amdgcn_rcp(s);
    y = __builtin_fma(__builtin_fma(-s, y, 1.0), y, y);
    return __builtin_fma(__builtin_fma(-s, y, 1.0), y, y);
}
__device__ __forceinline__ float pivot_rcp(float s) { return 1.0f / s; }

// Factor the nb x nb diagonal block at (k0, k0).  Thread (r = lane, w = wave) holds A[r][w + NW k], k = 0..NC-1, in NC registers
// (NW = 8 waves, NC = 8 columns).  Step j (column j = NW jq + jj lives in wave jj, register jq): the owning wave puts the column as it stands ("raw":
// a_rj after the updates of columns < j) into LDS; after ONE barrier every thread reads the pivot p = a_jj, its row entry a_rj
// and the entries a_cj of its own columns c > j, and applies a_rc -= (a_rj a_cj) / p  (= L_rj L_cj).  The square roots are
// taken after the loop (L_rj = a_rj / sqrt(p_j) from the raw columns kept in the registers and the pivots kept in LDS), so that
// the chain from one column to the next is LDS write -> barrier -> LDS read -> reciprocal -> 2 flops.  Registers of the strict
// upper triangle collect garbage that is never read.  The raw-column buffer is double-buffered: a wave may write column j + 1
// while a slower one still reads column j.
// A short block (nb < 64, last block of a ragged matrix) is padded with the identity.  info[0] receives k0 + j + 1 for the
// first non-positive pivot (only if still 0); the block is then left unwritten.  Dblk receives a dense transposed copy of the
// factored block (Dblk[t * 64 + c] = L[c][t] for c > t, zero elsewhere) plus the reciprocal diagonal for the panel kernel.
#ifndef CHOL_DIAG_WAVES
#define CHOL_DIAG_WAVES 8  // waves of the diagonal-block kernel: each thread holds 64 / waves columns of its row (16 / 8 / 4 waves: 792 / 730 / 808 us per 1024^2 factorisation - the per-column work every wave repeats competes for issue slots with the updates)
#endif
template <typename T>
__global__ __launch_bounds__(64 * CHOL_DIAG_WAVES) void chol_diag_kernel(T* __restrict__ A, int n, int k0, int nb, int* __restrict__ info,
                                                         T* __restrict__ Dblk) {
    __shared__ T colraw[2][CHOL_NB];
    __shared__ T pivs[CHOL_NB];
    const int r = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = CHOL_DIAG_WAVES, NC = CHOL_NB / NW;  // waves, columns per thread
    T a[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int cc = w + NW * k;
        T v = (cc == r) ? T(1) : T(0);  // identity padding
        if (r < nb && cc < nb) v = (cc <= r) ? A[(size_t)(k0 + cc) * n + k0 + r] : T(0);
        a[k] = v;
    }
    int bad = 0;
#pragma unroll
    for (int jq = 0; jq < NC; ++jq) {
#pragma unroll 1
        for (int jj = 0; jj < NW; ++jj) {
            const int j = NW * jq + jj;
            T* buf = colraw[j & 1];
            if (w == jj) buf[r] = a[jq];
            __syncthreads();
            const T piv = buf[j];
            const T xr = buf[r];
            const bool ok = piv > T(0);  // false for NaN too; every thread sees the same pivot
            bad = (!ok && bad == 0) ? j + 1 : bad;
            if (threadIdx.x == 0) pivs[j] = ok ? piv : T(1);
            const T rp = pivot_rcp(ok ? piv : T(1));
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (k < jq) continue;
                const T upd = tfma<T>(-(xr * buf[w + NW * k]), rp, a[k]);  // the product does not wait for the reciprocal
                a[k] = (k > jq || w > jj) ? upd : a[k];
            }
        }
    }
    if (bad != 0) {
        if (threadIdx.x == 0 && info[0] == 0) info[0] = k0 + bad;
        return;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int cc = w + NW * k;
        T root, rinv;
        pivot_root(pivs[cc], root, rinv);
        const T l = (r == cc) ? root : (r > cc ? a[k] * rinv : T(0));
        if (r < nb && cc <= r && cc < nb) A[(size_t)(k0 + cc) * n + k0 + r] = l;
        Dblk[cc * CHOL_NB + r] = (r > cc) ? l : T(0);  // strictly lower part, transposed: Dblk[t][c] = L[c][t]
        if (r == cc) Dblk[CHOL_NB * CHOL_NB + cc] = rinv;
    }
}

__device__ __forceinline__ double readlane_t(double v, int lane) {  // lane: wave-uniform, may be a run-time value
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_t(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Panel below the diagonal block: X L_kk^T = P.  ONE WAVE PER ROW of P, lane c holds P[row][c]; 16 rows per workgroup share one
// LDS copy of the factor block (stored transposed by the diag kernel: column t of L_kk is contiguous, conflict-free).
// Right-looking substitution: x_t = p_t / L_tt is final in lane t, reaches the other lanes through v_readlane (run-time lane
// index: the loop stays rolled), and every later entry takes p_c -= x_t L[c][t]; the chain from step to step is mul - readlane -
// fma, the LDS reads of a block of 8 steps are issued ahead of it.  (Global loads of the columns with a register prefetch
// queue measured 21 us per panel: the wait counters are drained at every trip of a rolled loop.)
template <typename T>
__global__ __launch_bounds__(1024) void chol_panel_kernel(T* __restrict__ A, int n, int k0, const T* __restrict__ Dblk) {
    __shared__ T Ls[CHOL_NB * CHOL_NB + CHOL_NB];
    for (int i = threadIdx.x; i < CHOL_NB * CHOL_NB + CHOL_NB; i += 1024) Ls[i] = Dblk[i];
    const int lane = threadIdx.x & 63;
    const int row = k0 + CHOL_NB + blockIdx.x * 16 + (threadIdx.x >> 6);
    T p = row < n ? A[(size_t)(k0 + lane) * n + row] : T(0);
    __syncthreads();
    if (row >= n) return;  // whole waves leave together
    const T myrinv = Ls[CHOL_NB * CHOL_NB + lane];
#pragma unroll 1
    for (int t0 = 0; t0 < CHOL_NB; t0 += 8) {
        T lq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lq[i] = Ls[(t0 + i) * CHOL_NB + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + i;
            const T xt = readlane_t(p * myrinv, t);
            p = (lane == t) ? xt : (lane > t ? tfma<T>(-xt, lq[i], p) : p);
        }
    }
    A[(size_t)(k0 + lane) * n + row] = p;
}

// Trailing update A22 -= P P^T (lower triangle), P = the 64 panel columns just solved (rows k0 + 64 ... n - 1).  Workgroup
// (I, J), J <= I, owns the 64 x 64 tile of A22 at block row I, block column J: both 64 x 64 panel slices go to LDS ([k][row]:
// the 16 rows a quarter-wave reads are contiguous, the other operand is a 4-address broadcast), each thread holds a 4 x 4
// register tile of products summed in fixed order; the tile of A is loaded together with the panel slices (one memory round trip
// covers both) and the sum is subtracted once at the end (one rounding at the magnitude of A per step, which fp32 needs).  Rows and columns beyond n (ragged M) are zero-filled and not stored.
template <typename T>
__global__ __launch_bounds__(256) void chol_update_kernel(T* __restrict__ A, int n, int k0) {
    const int I = blockIdx.x, J = blockIdx.y;
    if (J > I) return;
    __shared__ T Pi[CHOL_NB][CHOL_NB];
    __shared__ T Pj[CHOL_NB][CHOL_NB];
    const int t = threadIdx.x;
    const int base = k0 + CHOL_NB;
    const int ti = t & 15, tj = t >> 4;
    T c0[4][4], acc[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int cj = base + J * CHOL_NB + tj + 16 * b;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ri = base + I * CHOL_NB + ti + 16 * a;
            c0[a][b] = (ri < n && cj < n && ri >= cj) ? A[(size_t)cj * n + ri] : T(0);
            acc[a][b] = T(0);
        }
    }
    {
        const int rl = t & 63;
        const int ri = base + I * CHOL_NB + rl, rj = base + J * CHOL_NB + rl;
#pragma unroll
        for (int q = 0; q < CHOL_NB / 4; ++q) {
            const int k = (t >> 6) + 4 * q;
            Pi[k][rl] = ri < n ? A[(size_t)(k0 + k) * n + ri] : T(0);
            Pj[k][rl] = rj < n ? A[(size_t)(k0 + k) * n + rj] : T(0);
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < CHOL_NB; ++k) {
        T pi[4], pj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) pi[a] = Pi[k][ti + 16 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) pj[b] = Pj[k][tj + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = tfma<T>(pi[a], pj[b], acc[a][b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int cj = base + J * CHOL_NB + tj + 16 * b;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ri = base + I * CHOL_NB + ti + 16 * a;
            if (ri < n && cj < n && ri >= cj) A[(size_t)cj * n + ri] = c0[a][b] - acc[a][b];
        }
    }
}

template <typename T>
static int cholesky_impl(cglb_ctx* c, T* A, int* info_slot) {
    const int n = c->M;
    if (!c->chol_blk) HIP_CHECK(c, hipMalloc(&c->chol_blk, (CHOL_NB * CHOL_NB + CHOL_NB) * sizeof(double)));
    T* Dblk = (T*)c->chol_blk;
    HIP_CHECK(c, hipMemsetAsync(info_slot, 0, sizeof(int), c->stream));
    for (int k0 = 0; k0 < n; k0 += CHOL_NB) {
        const int nb = n - k0 < CHOL_NB ? n - k0 : CHOL_NB;
        hipLaunchKernelGGL((chol_diag_kernel<T>), dim3(1), dim3(64 * CHOL_DIAG_WAVES), 0, c->stream, A, n, k0, nb, info_slot, Dblk);
        CGLB_LAUNCH_CHECK(c);
        const int rest = n - k0 - nb;
        if (rest > 0) {  // nb == CHOL_NB here: a short block is the last one (rest == 0)
            hipLaunchKernelGGL((chol_panel_kernel<T>), dim3((rest + 15) / 16), dim3(1024), 0, c->stream, A, n, k0, (const T*)Dblk);
            CGLB_LAUNCH_CHECK(c);
            const int tn = (rest + CHOL_NB - 1) / CHOL_NB;
            hipLaunchKernelGGL((chol_update_kernel<T>), dim3(tn, tn), dim3(256), 0, c->stream, A, n, k0);
            CGLB_LAUNCH_CHECK(c);
        }
    }
    return CGLB_OK;
}

// In-place lower Cholesky of the column-major M x M matrix `A`; *info_slot (device int) = 0 or 1 + index of the first
// non-positive pivot.  A failed factorisation leaves garbage below the failing block (callers check info before use).
int launch_cholesky_lower(cglb_ctx* c, void* A, int* info_slot) {
    CGLB_DISPATCH_T(c->dtype, return cholesky_impl<T>(c, (T*)A, info_slot));
}
