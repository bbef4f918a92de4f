// K5 — gradient passes (what torch.autograd.grad(loss, variables) yields in
// pytorch/optimizer.py:95-98 with v detached, models.py:257-274), written as explicit bilinear forms:
//   grad_kff : sum_ij u_i (dK_ff/dl_d)_ij v_j                       (N^2 pairs, vector-fp64 bound)
//   grad_kuf : sum_mn G[m][n] dK_uf[m][n]/d{l_d, var, z_m}          (M x N, streams the adjoint panel)
// Direct differences are used here (the per-dimension delta^2 is needed anyway), so coincident points
// contribute exactly 0 and the Matern-3/2 derivative needs no sqrt guard:
//   dk/dl_d = var * h * delta_d^2 / l_d,  dk/dx1_d = -var * h * delta_d / l_d,
//   delta_d = (x1_d - x2_d)/l_d,  h = kappa (RBF) | 3 exp(-sqrt3 r) (Matern-3/2).
// Operands are the scaled ones of K1 (xs = (x-c)/l*kscale); the 1/(l_d kscale^2) factors are applied
// in the finalize kernels.
#include "devmath.h"
#include "dispatch.h"

// ---- N^2 pass ---------------------------------------------------------------------------------------
// Thread owns R rows; the column operand (xs_j, v_j, u_j) is wave-uniform (scalar loads).
// part[(blk0 + by * gridDim.x + bx) * DP + d] = sum over the block's pairs of w_ij h_ij (xs_id - xs_jd)^2
//   SYM = false: rows x columns [0, ncols) of another index range, w_ij = u_i v_j.
//   SYM = true : the square block rows x rows.  The form is symmetric under i <-> j, so only columns at or beyond the
//                block's first row are visited: the diagonal 256R x 256R block in full with w = u_i v_j, everything to its
//                right once with w = u_i v_j + u_j v_i  (halves the pair evaluations).
typedef float gk_f2 __attribute__((ext_vector_type(2)));  // two rows of a lane side by side (fp32 path of grad_kff_kernel)
template <typename T, int KIND, int DP, int R, bool SYM, int PREC>
__global__ __launch_bounds__(256) void grad_kff_kernel(const T* __restrict__ XsRow, const T* __restrict__ uRow, const T* __restrict__ vRow,
                                                       int64_t nrows, const T* __restrict__ XsCol, const T* __restrict__ vCol,
                                                       const T* __restrict__ uCol, int64_t ncols, int64_t jchunk, int64_t blk0,
                                                       int rb_stride, int rb_offset, double* __restrict__ part,
                                                       const double* __restrict__ exp_tab) {
    __shared__ double smem[16];
    __shared__ double tab[CGLB_TAB_SIZE];
    load_exp_table(tab, exp_tab);
    const int64_t rblock = ((int64_t)blockIdx.x * rb_stride + rb_offset) * (256 * R);  // cyclic over ranks when rb_stride > 1
    const int64_t rbase = rblock + threadIdx.x;
    if constexpr (sizeof(T) == 4 && R == 2) {
        // fp32: the two rows of a lane side by side, so that differences, squares, the distance and the D accumulations run as v_pk_*_f32
        // (4 packed instructions per dimension and row PAIR against 4 per dimension and row); the column operand is broadcast to both halves
        // from its SGPR.  The pair weight (hardware exp2) stays per row.  Measured at N = 400 000: D = 16 101.2 -> 90.9 ms, D = 8 56.0 -> 49.8 ms;
        // padded width 28 loses (42.4 -> 48.4 ms at N = 200 000: registers), so wider rows keep one row per lane.
        gk_f2 xi2[DP], acc2[DP];
        float ui2[2], vi2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t row = rbase + (int64_t)k * 256;
            const int64_t rr = row < nrows ? row : nrows - 1;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                xi2[d][k] = XsRow[rr * DP + d];
                acc2[d][k] = 0.f;
            }
            ui2[k] = row < nrows ? uRow[rr] : 0.f;
            vi2[k] = (SYM && row < nrows) ? vRow[rr] : 0.f;
        }
        int64_t j0 = (int64_t)blockIdx.y * jchunk;
        const int64_t j1 = (j0 + jchunk < ncols) ? j0 + jchunk : ncols;
        const int64_t sym_from = rblock + 256 * R;
        if (SYM && j0 < rblock) j0 = rblock;
        for (int64_t j = j0; j < j1; ++j) {
            const float vj = vCol[j];
            float wu = 0.f;
            if (SYM) wu = (j >= sym_from) ? uCol[j] : 0.f;  // wave-uniform
            gk_f2 sq[DP], d2 = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const float x = XsCol[j * DP + d];
                const gk_f2 df = xi2[d] - gk_f2{x, x};
                sq[d] = df * df;
                d2 += sq[d];
            }
            gk_f2 hv;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float w = ui2[k] * vj;
                if (SYM) w = __builtin_fmaf(vi2[k], wu, w);
                hv[k] = hfac_hot_from_d2<float, KIND, true, PREC>(d2[k], tab) * w;
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) acc2[d] = __builtin_elementwise_fma(hv, sq[d], acc2[d]);
        }
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            double s = (double)acc2[d][0] + (double)acc2[d][1];
            s = block_sum(s, smem);
            if (threadIdx.x == 0) part[(blk0 + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * DP + d] = s;
        }
        return;
    }
    T xi[R][DP], acc[R][DP], ui[R], vi[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t row = rbase + (int64_t)k * 256;
        const int64_t rr = row < nrows ? row : nrows - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            xi[k][d] = XsRow[rr * DP + d];
            acc[k][d] = 0;
        }
        ui[k] = row < nrows ? uRow[rr] : T(0);
        vi[k] = (SYM && row < nrows) ? vRow[rr] : T(0);
    }
    int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < ncols) ? j0 + jchunk : ncols;
    const int64_t sym_from = rblock + 256 * R;
    if (SYM && j0 < rblock) j0 = rblock;
    for (int64_t j = j0; j < j1; ++j) {
        const T vj = vCol[j];
        T wu = 0;
        if (SYM) wu = (j >= sym_from) ? uCol[j] : T(0);  // wave-uniform
        T xj[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) xj[d] = XsCol[j * DP + d];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            T sq[DP];
            T d2 = 0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const T df = xi[k][d] - xj[d];
                sq[d] = df * df;
                d2 += sq[d];
            }
            T w = ui[k] * vj;
            if (SYM) w = tfma<T>(vi[k], wu, w);
            const T hv = hfac_hot_from_d2<T, KIND, true, PREC>(d2, tab) * w;
#pragma unroll
            for (int d = 0; d < DP; ++d) acc[k][d] = tfma<T>(hv, sq[d], acc[k][d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s += (double)acc[k][d];
        s = block_sum(s, smem);
        if (threadIdx.x == 0) part[(blk0 + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * DP + d] = s;
    }
}

// Gram-form variant of the symmetric N^2 pass (SYM only, unclamped exponent range - the common case; selected by launch_grad_kff*):
// the pair value comes from the Gram chain seeded with the row norm exactly as in the mat-vec (D fma + 1 instead of D sub, D mul,
// D add), and the per-dimension sums are built from moments,
//   sum_j hv_ij (x_id - x_jd)^2 = x_id^2 S0_i - 2 x_id S1_id + S2_id,   S0 = sum hv, S1_d = sum hv x_jd, S2_d = sum hv x_jd^2,
// with x_jd a wave-uniform scalar operand.  S0 and S1 are per row (they are weighted by x_id afterwards); S2 enters only through its sum
// over rows, i.e. through the column sums of the pair weights, which are formed 8 columns at a time by a transposition through LDS (below):
// D + 2 accumulate instructions per pair and ~1 for the transposition (D = 8: ~31 in all, against 46 for direct differences).
// The expansion cancels when |x_d| >> |x_id - x_jd| for the pairs that carry weight (lengthscale far below the data range): the
// operands are centred, so the loss is ~log10((range / l)^2) of 16 digits - irrelevant against the 1e-8 the optimiser needs.
#ifndef CGLB_GRAM_W3_DP
#define CGLB_GRAM_W3_DP 28  // padded width whose instance is pinned to 3 waves per SIMD (175 VGPRs unpinned: 2 waves; pinned 28 B of scratch: 7.26 -> 5.62 ms at N = 60k; the same at width 32 costs 130 B of scratch: 7.4 -> 8.5 ms)
#endif
#define GRAD_TR_LD 65  // leading dimension of a wave's 8 x 64 transposition scratch (odd: the column reads spread over the banks)
template <typename T, int KIND, int DP, int R, int PREC, bool CLAMP>
__global__ __launch_bounds__(256, (DP == CGLB_GRAM_W3_DP ? 3 : 1)) void grad_kff_gram_kernel(const T* __restrict__ Xh, const T* __restrict__ Xhsq, const T* __restrict__ ah,
                                                            const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ uc,
                                                            const T* __restrict__ vc, int64_t row0, int64_t n,
                                                            int64_t jchunk, int64_t blk0, int rb_stride, int rb_offset,
                                                            double* __restrict__ part, const double* __restrict__ exp_tab, T bias) {
    __shared__ double smem[16];
    __shared__ double tab[CGLB_TAB_SIZE];
    load_exp_table(tab, exp_tab);
    // RBF, unclamped range: folded column norm as in the symmetric mat-vec - h_ij = 2^(a_i + x_i.x_j) w_j with w_j = 2^(a_j) carried by the
    // column-side copies uc = u o w, vc = v o w of the two vectors (the row side uses u, v themselves), so the per-pair add of a_j is dropped
    constexpr bool FOLD = (KIND == CGLB_RBF) && !CLAMP && sizeof(T) == 8;
    // Matern-3/2, fast level, unclamped range: positivity bias in the row seeds instead of a clamp per pair (devmath.h CGLB_M32_BIAS_*)
    constexpr bool BIASED = (KIND != CGLB_RBF) && !CLAMP && PREC != CGLB_PREC_EXACT && sizeof(T) == 8;
    const int64_t rblock = ((int64_t)blockIdx.x * rb_stride + rb_offset) * (256 * R);  // cyclic over ranks when rb_stride > 1
    const int64_t rbase = rblock + threadIdx.x;
    __shared__ T trbuf[4 * 8 * GRAD_TR_LD];
    T* __restrict__ tr = trbuf + (threadIdx.x >> 6) * (8 * GRAD_TR_LD);
    const int lane = threadIdx.x & 63;
    // second moments: through column sums (COLSUM, below) or, for D <= 4 where that costs more than it saves, per lane with the R rows
    // sharing one accumulator set (S2 enters only through its sum over rows)
    constexpr bool COLSUM = DP > 4;
    constexpr int NQ = COLSUM ? (DP + 7) / 8 : DP;  // COLSUM: dimensions (lane >> 3) + 8 q of column (lane & 7) of a batch; else S2[d]
    T xi[R][DP], S1[R][DP], G2[NQ], S0[R], aseed[R], ui[R], vi[R];
#pragma unroll
    for (int q = 0; q < NQ; ++q) G2[q] = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t row = rbase + (int64_t)k * 256;
        const int64_t rr = row < n ? row : n - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            xi[k][d] = Xh[(row0 + rr) * DP + d];
            S1[k][d] = 0;
        }
        S0[k] = 0;
        const T a = ah[row0 + rr];
        aseed[k] = (KIND == CGLB_RBF) ? a : (BIASED ? T(-0.5) * (a + bias) : T(-0.5) * a);
        // padded rows carry zero weight; Matern-3/2: h = 3 * 2^(-r), the factor 3 rides in the row weights
        const T hscale = (KIND == CGLB_RBF) ? T(1) : T(3);
        ui[k] = row < n ? hscale * u[row0 + rr] : T(0);
        vi[k] = row < n ? hscale * v[row0 + rr] : T(0);
    }
    int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < n) ? j0 + jchunk : n;
    const int64_t sym_from = rblock + 256 * R;
    if (j0 < rblock) j0 = rblock;
    for (int64_t jb = j0; jb < j1; jb += 8) {
        T t[8];  // per column of the batch: the sum of this lane's R pair weights
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int64_t jc = jb + jj;
            t[jj] = 0;
            if (jc < j1) {  // wave-uniform: only the last batch of a chunk is short
                const int64_t j = row0 + jc;
                const T vj = vc[j];
                const T wu = (jc >= sym_from) ? uc[j] : T(0);  // wave-uniform
                const T aj = FOLD ? T(0) : ah[j];
                T xj[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) xj[d] = Xh[j * DP + d];
                T earg[R], h[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    T g = aseed[k];
#pragma unroll
                    for (int d = 0; d < DP; ++d) g = tfma<T>(xi[k][d], xj[d], g);
                    earg[k] = (KIND == CGLB_RBF) ? (FOLD ? g : g + aj) : sqrt_hot<PREC, BIASED>(tfma<T>(T(-2), g, aj));  // RBF: exponent; Matern: r (exponent -r)
                }
                exp2_tab_batch<CLAMP, KIND != CGLB_RBF, PREC, R>(earg, tab, h);  // CLAMP: exponents beyond the table's range (set_hypers decides)
                T cj = 0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    T w = ui[k] * vj;
                    w = tfma<T>(vi[k], wu, w);
                    const T hv = h[k] * w;
                    S0[k] += hv;
                    cj = (k == 0) ? hv : cj + hv;
#pragma unroll
                    for (int d = 0; d < DP; ++d) S1[k][d] = tfma<T>(hv, xj[d], S1[k][d]);
                }
                t[jj] = cj;
                if constexpr (!COLSUM) {
#pragma unroll
                    for (int d = 0; d < DP; ++d) G2[d] = tfma<T>(cj, Xhsq[j * DP + d], G2[d]);
                }
            }
        }
        if constexpr (!COLSUM) continue;
        // Second moments.  sum_ij hv_ij x_jd^2 = sum_j x_jd^2 (sum_i hv_ij): only the COLUMN sums of the pair weights are needed,
        // so the per-pair accumulation S2_d += hv x_jd^2 (D fma per pair and D more scalar operands per column - at D = 24 / 32 they
        // no longer fit the SGPR file: 64 v_readlane/v_writelane per column, a 3.8x slower kernel) is replaced by one transposition
        // of the 8 partials through LDS per batch (as in the symmetric mat-vec: lane (c, g) adds 8 lanes of column c, three
        // xor-shuffles finish the wave's column sum) and NQ fma per lane: lane (c, g) weights x_{jb+c, d}^2 for d = g, g + 8, ...
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tr[jj * GRAD_TR_LD + lane] = t[jj];
        __builtin_amdgcn_wave_barrier();
        const T* __restrict__ src = tr + (lane & 7) * GRAD_TR_LD + (lane & ~7);
        T cs = src[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) cs += src[i];
        __builtin_amdgcn_wave_barrier();
        cs += __shfl_xor(cs, 8, 64);
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        const int64_t jcol = jb + (lane & 7);
        const T* __restrict__ sq = Xhsq + (row0 + (jcol < j1 ? jcol : j1 - 1)) * DP;  // columns past the chunk carry cs == 0
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = (lane >> 3) + 8 * q;
            if (d < DP) G2[q] = tfma<T>(cs, sq[d], G2[q]);
        }
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        double s = COLSUM ? (((lane >> 3) == (d & 7)) ? (double)G2[d >> 3] : 0.0) : (double)G2[COLSUM ? 0 : d];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double x = (double)xi[k][d];
            s += x * (x * (double)S0[k] - 2.0 * (double)S1[k][d]);
        }
        s = block_sum(s, smem);
        if (threadIdx.x == 0) part[(blk0 + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * DP + d] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void square_kernel(const T* __restrict__ x, int64_t n, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * x[i];
}
// Xhsq = Xh .* Xh (after set_hypers): the second-moment operand of the Gram-form gradient pass
int launch_hot_squares(cglb_ctx* c) {
    if (is_wide(c)) return CGLB_OK;
    const int64_t tot = c->N * c->Dp;
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((square_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, (const T*)c->Xh,
                                                 tot, (T*)c->Xhsq));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// out[d] += scale[d] * sum_b part[b*DP + d]   (one block per d)
__global__ __launch_bounds__(256) void grad_dl_finalize_kernel(const double* __restrict__ part, int64_t nblk, int DP, int D,
                                                               ScaleParams sp, double var, double* __restrict__ out, int accumulate) {
    __shared__ double smem[16];
    const int d = blockIdx.x;
    if (d >= D) return;
    double s = 0.0;
    for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) s += part[b * DP + d];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) {
        const double val = s * var * sp.scale[d];  // sp.scale holds 1/(l_d kscale^2) here
        out[d] = accumulate ? out[d] + val : val;
    }
}

int ensure_gpart(cglb_ctx* c, size_t need) {
    if (need > c->gpart_cap) {
        if (c->gpart) HIP_CHECK(c, hipFree(c->gpart));
        c->gpart = nullptr;
        HIP_CHECK(c, hipMalloc((void**)&c->gpart, need));
        c->gpart_cap = need;
    }
    return CGLB_OK;
}

// rows per lane of the Gram-form kernel when D <= 8 (4 rows - 204 VGPRs, 2 waves per SIMD - measured 6.28 ms against 5.55 ms)
#ifndef CGLB_GRAM_R2_MAX_DP
#define CGLB_GRAM_R2_MAX_DP 12  // widest padded row with CGLB_GRAM_ROWS rows per lane in the Gram-form kernel (DP = 12: 8.4 -> 7.2 ms at N = 100k against 1 row)
#endif
#ifndef CGLB_GRAM_R4_MAX_DP
#define CGLB_GRAM_R4_MAX_DP 4  // widest padded row with FOUR rows per lane (D <= 4: -6..-8 % against two)
#endif
#define CGLB_GRAM_ROWS_OF(dp) ((dp) <= CGLB_GRAM_R4_MAX_DP ? 4 : ((dp) <= CGLB_GRAM_R2_MAX_DP ? CGLB_GRAM_ROWS : 1))
#ifndef CGLB_GRAM_ROWS
#define CGLB_GRAM_ROWS 2
#endif

static inline double kscale_of(const cglb_ctx* c) { return (c->kind == CGLB_RBF) ? sqrt(CGLB_LOG2E) : CGLB_SQRT3 * CGLB_LOG2E; }

// column-side copies of the two vectors of the bilinear form for the folded column norm (RBF, unclamped range): vw = v o wh over all
// n_v entries, uw[off + i] = u[i] wh[off + i] over the n_u entries the caller holds.  vw lives in the mat-vec's pwh buffer.
template <typename T>
__global__ __launch_bounds__(256) void grad_weight_kernel(const T* __restrict__ v, int64_t n_v, const T* __restrict__ u, int64_t off, int64_t n_u,
                                                          const T* __restrict__ wh, T* __restrict__ vw, T* __restrict__ uw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_v) vw[i] = v[i] * wh[i];
    if (i < n_u) uw[off + i] = u[i] * wh[off + i];
}
int grad_fold_operands(cglb_ctx* c, const void* v_full, const void* u, int64_t off, int64_t n_u, bool* fold) {
    *fold = c->kind == CGLB_RBF && !c->exp_clamp && c->dtype == CGLB_F64 && c->have_hypers;
    if (!*fold) return CGLB_OK;
    if (!c->uwh) HIP_CHECK(c, hipMalloc(&c->uwh, (size_t)c->N * c->esz));
    c->pwh_src = nullptr;  // pwh no longer holds the weighted direction of the PCG loop
    const int64_t n = c->N > n_u ? c->N : n_u;
    hipLaunchKernelGGL((grad_weight_kernel<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)v_full, c->N,
                       (const double*)u, off, n_u, (const double*)c->wh, (double*)c->pwh, (double*)c->uwh);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// out_dl[d] (device double[D], overwritten) = sum_{i local, j} u_i dK_ij/dl_d v_j
int launch_grad_kff(cglb_ctx* c, const void* v_full, const void* u_local, double* out_dl) {
    if (is_wide(c) && mid_reg(c) && c->nloc == c->N) return launch_grad_kff_mid(c, v_full, u_local, 1, 0, out_dl);
    if (is_wide(c)) return wide_grad_kff(c, v_full, u_local, c->r0, c->nloc, 1, 0, out_dl);
    ScaleParams sp;
    const double ks = kscale_of(c) * cglb_hot_scale(c);  // the N^2 pass runs on the hot operand set
    for (int d = 0; d < CGLB_MAX_D_NARROW; ++d) {
        sp.center[d] = 0;
        sp.scale[d] = d < c->D ? 1.0 / (c->ls[d] * ks * ks) : 0.0;
    }
    if (c->nloc == 0) {
        HIP_CHECK(c, hipMemsetAsync(out_dl, 0, sizeof(double) * c->D, c->stream));
        return CGLB_OK;
    }
    // fp32 keeps direct differences (no digits to spare); the clamped exponent range (large scaled coordinates: wide D at short
    // lengthscales, e.g. the reference's initial l = 1 at D >= 17) runs the same Gram form with the range clamp of the 2^x
    const bool use_gram = c->grad_gram && c->dtype == CGLB_F64;
    bool fold = false;
    if (use_gram) CGLB_TRY(grad_fold_operands(c, v_full, u_local, c->r0, c->nloc, &fold));
    const int R = (c->Dp <= 8 || (c->dtype == CGLB_F32 && c->Dp <= 16)) ? 2 : 1;   // fp32 up to width 16: two rows per lane (packed arithmetic)
    const int Rg = CGLB_GRAM_ROWS_OF(c->Dp);  // the Gram-form kernel (square range only)
    // three column ranges: the square block (symmetric form) and the shard's off-diagonal ranges [0,r0), [r1,N)
    struct Range { int64_t col0, ncols; bool sym; int64_t jsplit, jchunk; } rg[3] = {
        {c->r0, c->nloc, true, 0, 0}, {0, c->r0, false, 0, 0}, {c->r1, c->N - c->r1, false, 0, 0}};
    int64_t nblk = 0;
    for (auto& r : rg) {
        if (r.ncols <= 0) continue;
        const int rr = (r.sym && use_gram) ? Rg : R;
        const int64_t bx = (c->nloc + 256 * rr - 1) / (256 * rr);
        int64_t js = (8192 + bx - 1) / bx;
        if (r.sym) js *= 2;  // about half of the (row block, chunk) cells of the square are left of the diagonal and exit at once
        if (js > 1024) js = 1024;
        if (js > (r.ncols + 63) / 64) js = (r.ncols + 63) / 64;
        if (js < 1) js = 1;
        r.jchunk = (r.ncols + js - 1) / js;
        r.jsplit = (r.ncols + r.jchunk - 1) / r.jchunk;
        nblk += bx * r.jsplit;
    }
    CGLB_TRY(ensure_gpart(c, (size_t)nblk * c->Dp * sizeof(double)));
    int64_t blk0 = 0;
    for (auto& r : rg) {
        if (r.ncols <= 0) continue;
        const int rr = (r.sym && use_gram) ? Rg : R;
        const int64_t bx = (c->nloc + 256 * rr - 1) / (256 * rr);
        dim3 grid((unsigned)bx, (unsigned)r.jsplit);
#define GK_LAUNCH(RR, SYMV)                                                                                                        \
    hipLaunchKernelGGL((grad_kff_kernel<T, KIND, DP, RR, SYMV, PREC>), grid, dim3(256), 0, c->stream, (const T*)c->Xh + c->r0 * DP,        \
                       (const T*)u_local, (const T*)v_full + c->r0, c->nloc, (const T*)c->Xh + r.col0 * DP, (const T*)v_full + r.col0, \
                       (const T*)u_local, r.ncols, r.jchunk, blk0, 1, 0, c->gpart, (const double*)c->exp_tab)
#define GG_LAUNCH1(RR, CL)                                                                                                            \
    hipLaunchKernelGGL((grad_kff_gram_kernel<T, KIND, DP, RR, PREC, CL>), grid, dim3(256), 0, c->stream, (const T*)c->Xh, (const T*)c->Xhsq, \
                       (const T*)c->xah, (const T*)u_local - c->r0, (const T*)v_full, (CL || !fold) ? (const T*)u_local - c->r0 : (const T*)c->uwh,       \
                       (CL || !fold) ? (const T*)v_full : (const T*)c->pwh, c->r0, c->nloc, r.jchunk, blk0, 1, 0, c->gpart,               \
                       (const double*)c->exp_tab, (T)c->m32_bias)
#define GG_LAUNCH(RR) do { if (c->exp_clamp) { GG_LAUNCH1(RR, true); } else { GG_LAUNCH1(RR, false); } } while (0)
        if (r.sym && use_gram) {
            // u_local is indexed by global row inside the kernel (row0 + local), hence the shifted base pointer
            CGLB_DISPATCH_ALL(c, CGLB_DISPATCH_PREC(c, GG_LAUNCH(CGLB_GRAM_ROWS_OF(DP))));
        } else if (r.sym) { CGLB_DISPATCH_ALL(c, CGLB_DISPATCH_PREC(c, if constexpr (DP <= 8 || (sizeof(T) == 4 && DP <= 16)) { GK_LAUNCH(2, true); } else { GK_LAUNCH(1, true); })); }
        else { CGLB_DISPATCH_ALL(c, CGLB_DISPATCH_PREC(c, if constexpr (DP <= 8 || (sizeof(T) == 4 && DP <= 16)) { GK_LAUNCH(2, false); } else { GK_LAUNCH(1, false); })); }
#undef GG_LAUNCH
#undef GG_LAUNCH1
#undef GK_LAUNCH
        CGLB_LAUNCH_CHECK(c);
        blk0 += bx * r.jsplit;
    }
    hipLaunchKernelGGL(grad_dl_finalize_kernel, dim3(c->D), dim3(256), 0, c->stream, (const double*)c->gpart, nblk, c->Dp, c->D, sp,
                       c->var, out_dl, 0);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Cyclic form for the multi-GPU path: the whole N x N form in its symmetric version, row blocks rb == par_rank (mod par_world).
// u_full, v_full: all N entries.  out_dl (device double[D], overwritten) is this rank's PARTIAL sum (all-reduced by the caller).
int launch_grad_kff_cyclic(cglb_ctx* c, const void* v_full, const void* u_full, double* out_dl) {
    if (is_wide(c) && mid_reg(c)) return launch_grad_kff_mid(c, v_full, u_full, c->par_world, c->par_rank, out_dl);
    if (is_wide(c)) return wide_grad_kff(c, v_full, u_full, 0, c->N, c->par_world, c->par_rank, out_dl);
    ScaleParams sp;
    const double ks = kscale_of(c) * cglb_hot_scale(c);
    for (int d = 0; d < CGLB_MAX_D_NARROW; ++d) {
        sp.center[d] = 0;
        sp.scale[d] = d < c->D ? 1.0 / (c->ls[d] * ks * ks) : 0.0;
    }
    const bool use_gram = c->grad_gram && c->dtype == CGLB_F64;
    bool fold = false;
    if (use_gram) CGLB_TRY(grad_fold_operands(c, v_full, u_full, 0, c->N, &fold));
    const int R = use_gram ? CGLB_GRAM_ROWS_OF(c->Dp) : ((c->Dp <= 8 || (c->dtype == CGLB_F32 && c->Dp <= 16)) ? 2 : 1);
    const int64_t nb = (c->N + 256 * R - 1) / (256 * R);
    const int64_t bx = c->par_rank < nb ? (nb - c->par_rank + c->par_world - 1) / c->par_world : 0;
    if (bx == 0) {
        HIP_CHECK(c, hipMemsetAsync(out_dl, 0, sizeof(double) * c->D, c->stream));
        return CGLB_OK;
    }
    int64_t js = 2 * ((8192 + bx - 1) / bx);
    if (js > 1024) js = 1024;
    if (js > (c->N + 63) / 64) js = (c->N + 63) / 64;
    if (js < 1) js = 1;
    const int64_t jchunk = (c->N + js - 1) / js;
    const int64_t jsplit = (c->N + jchunk - 1) / jchunk;
    const int64_t nblk = bx * jsplit;
    CGLB_TRY(ensure_gpart(c, (size_t)nblk * c->Dp * sizeof(double)));
    dim3 grid((unsigned)bx, (unsigned)jsplit);
#define GKC_LAUNCH(RR)                                                                                                            \
    hipLaunchKernelGGL((grad_kff_kernel<T, KIND, DP, RR, true, PREC>), grid, dim3(256), 0, c->stream, (const T*)c->Xh, (const T*)u_full,  \
                       (const T*)v_full, c->N, (const T*)c->Xh, (const T*)v_full, (const T*)u_full, c->N, jchunk, (int64_t)0,       \
                       c->par_world, c->par_rank, c->gpart, (const double*)c->exp_tab)
#define GGC_LAUNCH1(RR, CL)                                                                                                          \
    hipLaunchKernelGGL((grad_kff_gram_kernel<T, KIND, DP, RR, PREC, CL>), grid, dim3(256), 0, c->stream, (const T*)c->Xh, (const T*)c->Xhsq, \
                       (const T*)c->xah, (const T*)u_full, (const T*)v_full, (CL || !fold) ? (const T*)u_full : (const T*)c->uwh,                   \
                       (CL || !fold) ? (const T*)v_full : (const T*)c->pwh, (int64_t)0, c->N, jchunk, (int64_t)0, c->par_world, c->par_rank, \
                       c->gpart, (const double*)c->exp_tab, (T)c->m32_bias)
#define GGC_LAUNCH(RR) do { if (c->exp_clamp) { GGC_LAUNCH1(RR, true); } else { GGC_LAUNCH1(RR, false); } } while (0)
    if (use_gram) { CGLB_DISPATCH_ALL(c, CGLB_DISPATCH_PREC(c, GGC_LAUNCH(CGLB_GRAM_ROWS_OF(DP)))); }
    else { CGLB_DISPATCH_ALL(c, CGLB_DISPATCH_PREC(c, if constexpr (DP <= 8 || (sizeof(T) == 4 && DP <= 16)) { GKC_LAUNCH(2); } else { GKC_LAUNCH(1); })); }
#undef GGC_LAUNCH
#undef GGC_LAUNCH1
#undef GKC_LAUNCH
    CGLB_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(grad_dl_finalize_kernel, dim3(c->D), dim3(256), 0, c->stream, (const double*)c->gpart, nblk, c->Dp, c->D, sp,
                       c->var, out_dl, 0);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// ---- M x ncols pass over an adjoint panel --------------------------------------------------------------
// grid (M, nsplit); block (m, s) handles columns [s*chunk,(s+1)*chunk) of row m.
// G (adjoint, row m contiguous, ld = ldg) plus optional rank-1 term cvec[m]*wvec[n].
// part[(m*nsplit+s)*(2*DP+1) + {0..DP-1: dl, DP..2DP-1: dz, 2DP: df}]
template <typename T, int KIND, int DP>
__global__ __launch_bounds__(256) void grad_panel_kernel(const T* __restrict__ G, int64_t ldg, const T* __restrict__ cvec,
                                                         const T* __restrict__ wvec, const T* __restrict__ ZsRow,
                                                         const T* __restrict__ XsCol, int64_t ncols, int64_t chunk,
                                                         double* __restrict__ part) {
    __shared__ double smem[16];
    const int m = blockIdx.x;
    const int64_t n0 = (int64_t)blockIdx.y * chunk;
    const int64_t n1 = (n0 + chunk < ncols) ? n0 + chunk : ncols;
    T z[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) z[d] = ZsRow[(int64_t)m * DP + d];
    const T cm = cvec ? cvec[m] : T(0);
    double adl[DP], adz[DP], adf = 0.0;
#pragma unroll
    for (int d = 0; d < DP; ++d) { adl[d] = 0.0; adz[d] = 0.0; }
    for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x) {
        T g = CGLB_STREAM_LOAD(G + (int64_t)m * ldg + n);
        if (cvec) g = tfma<T>(cm, wvec[n], g);
        T df[DP], d2 = 0;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            df[d] = z[d] - XsCol[n * DP + d];
            d2 = tfma<T>(df[d], df[d], d2);
        }
        const T kap = kappa_from_d2<T, KIND>(d2);
        const T h = (KIND == CGLB_RBF) ? kap : hfac_from_d2<T, KIND>(d2);
        const T W = g * h;
        adf += (double)g * (double)kap;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const T wd = W * df[d];
            adz[d] += (double)wd;
            adl[d] += (double)(wd * df[d]);
        }
    }
    double* out = part + ((int64_t)m * gridDim.y + blockIdx.y) * (2 * DP + 1);
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        const double s = block_sum(adl[d], smem);
        if (threadIdx.x == 0) out[d] = s;
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        const double s = block_sum(adz[d], smem);
        if (threadIdx.x == 0) out[DP + d] = s;
    }
    const double s = block_sum(adf, smem);
    if (threadIdx.x == 0) out[2 * DP] = s;
}

// Finalize: out = packed gradient (dl[D], dvar, dnoise, dmean, dZ[M*D]); accumulates into it.
//   dl[d]   += var/(l_d ks^2) * sum_{m,s} adl
//   dvar    += sum_{m,s} adf
//   dZ[m,d] += -zfactor * var/(l_d ks) * sum_s adz
__global__ __launch_bounds__(256) void grad_panel_finalize_kernel(const double* __restrict__ part, int M, int nsplit, int DP, int D,
                                                                  ScaleParams sp, double var, double zfactor, double* __restrict__ out) {
    __shared__ double smem[16];
    const int W = 2 * DP + 1;
    const int b = blockIdx.x;
    if (b < D) {  // dl[b]
        double s = 0.0;
        for (int64_t k = threadIdx.x; k < (int64_t)M * nsplit; k += blockDim.x) s += part[k * W + b];
        s = block_sum(s, smem);
        if (threadIdx.x == 0) out[b] += s * var * sp.scale[b];
    } else if (b == D) {  // dvar
        double s = 0.0;
        for (int64_t k = threadIdx.x; k < (int64_t)M * nsplit; k += blockDim.x) s += part[k * W + 2 * DP];
        s = block_sum(s, smem);
        if (threadIdx.x == 0) out[D] += s;
    } else {  // dZ rows: blocks D+1 ... ; each thread one (m, d)
        const int64_t idx = (int64_t)(b - D - 1) * blockDim.x + threadIdx.x;
        if (idx < (int64_t)M * D) {
            const int m = (int)(idx / D), d = (int)(idx % D);
            double s = 0.0;
            for (int k = 0; k < nsplit; ++k) s += part[((int64_t)m * nsplit + k) * W + DP + d];
            out[D + 3 + idx] += -zfactor * var * sp.center[d] * s;  // sp.center holds 1/(l_d ks) here
        }
    }
}

static int grad_panel(cglb_ctx* c, const void* G, int64_t ldg, const void* cvec, const void* wvec, const void* XsCol, int64_t ncols,
                      double zfactor, double* out) {
    if (ncols == 0) return CGLB_OK;
    ScaleParams sp;
    const double ks = kscale_of(c);
    for (int d = 0; d < CGLB_MAX_D_NARROW; ++d) {
        sp.scale[d] = d < c->D ? 1.0 / (c->ls[d] * ks * ks) : 0.0;
        sp.center[d] = d < c->D ? 1.0 / (c->ls[d] * ks) : 0.0;
    }
    int nsplit = 1;
    while ((int64_t)c->M * nsplit < 2048 && nsplit < 64 && ncols / (nsplit * 2) >= 2048) nsplit *= 2;
    const int64_t chunk = (ncols + nsplit - 1) / nsplit;
    nsplit = (int)((ncols + chunk - 1) / chunk);
    const int W = 2 * c->Dp + 1;
    CGLB_TRY(ensure_gpart(c, (size_t)c->M * nsplit * W * sizeof(double)));
    dim3 grid((unsigned)c->M, (unsigned)nsplit);
    CGLB_DISPATCH_ALL(c, hipLaunchKernelGGL((grad_panel_kernel<T, KIND, DP>), grid, dim3(256), 0, c->stream, (const T*)G, ldg,
                                            (const T*)cvec, (const T*)wvec, (const T*)c->Zs, (const T*)XsCol, ncols, chunk, c->gpart));
    CGLB_LAUNCH_CHECK(c);
    const int zblocks = (int)(((int64_t)c->M * c->D + 255) / 256);
    hipLaunchKernelGGL(grad_panel_finalize_kernel, dim3(c->D + 1 + zblocks), dim3(256), 0, c->stream, (const double*)c->gpart, c->M,
                       nsplit, c->Dp, c->D, sp, c->var, zfactor, out);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Kuf part: adjoint = c->Guf (+ cvec[m] * w_local[n]); columns = local rows of X.
int launch_grad_kuf(cglb_ctx* c, const void* cvec, const void* w_local, double* out) {
    if (is_wide(c))
        return wide_grad_panel(c, c->Guf, c->lda, cvec, w_local, (const char*)c->Xs + (size_t)c->r0 * c->Dp * c->esz, (const char*)c->xa + (size_t)c->r0 * c->esz,
                               (const char*)c->Xsq + (size_t)c->r0 * c->Dp * c->esz, c->nloc, 1.0, out);
    return grad_panel(c, c->Guf, c->lda, cvec, w_local, (const char*)c->Xs + (size_t)c->r0 * c->Dp * c->esz, c->nloc, 1.0, out);
}

// Kuu part: adjoint = Guu (symmetric M x M) - c c^T/2 passed as rank-1 (cvec, wvec = -c/2); columns = Z.
// z_m appears in row and column of K_uu, hence the factor 2 on dZ.
int launch_grad_kuu(cglb_ctx* c, const void* Guu, const void* cvec, const void* mhalf_c, double* out) {
    if (is_wide(c)) return wide_grad_panel(c, Guu, c->M, cvec, mhalf_c, c->Zs, c->za, c->Zsq, c->M, 2.0, out);
    return grad_panel(c, Guu, c->M, cvec, mhalf_c, c->Zs, c->M, 2.0, out);
}
