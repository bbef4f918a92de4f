// K5, mid widths (32 < D <= 96, fp64): the symmetric Gram-form N^2 pass of kernels_grad.hip with a matrix row shared by two (four) lanes
// and the column operands broadcast across lanes.  A translation unit of its own: its instances build in parallel with the narrow ones.
#include "devmath.h"
#include "dispatch.h"

#define GRAD_TR_LD 65  // leading dimension of a wave's 8 x 64 transposition scratch (as in kernels_grad.hip)

// Mid-width form of the symmetric Gram-form pass (32 < D <= 96, fp64; cglb_internal.h: mid_dim).  A row's operand and its first moments
// no longer fit one lane (4 DP registers), so a matrix row is shared by SPLIT lanes (lane l of each half of the wave, SPLIT = 2; of each
// 16-lane row, SPLIT = 4, for DP = 96), each holding 1 / SPLIT of the dimensions: x_i and S1_i over HD = DP / SPLIT of them.  The column
// operands travel in VGPRs, 8 coordinates to a register pair (lane l holds coordinate 8 s + l % 8 of its part), and both the Gram chain
// and the moment accumulation take coordinate k from lane k of their 16-lane row through v_fmac_f64 row_newbcast (devmath.h:
// fmac_bcast) - the slices loaded for the chain are reused by the moments, and slice s of the next column is requested into the registers
// of slice s as soon as the moments have used them.  The partial Gram sums meet through v_permlane32_swap (and v_permlane16_swap); the
// pair weight is then computed redundantly in all parts (~16 of the ~(2 DP + 38) / 2 lane-instructions per pair and lane at SPLIT = 2).
// Column sums (second moments) count each row once: only part 0 contributes.
//   part[(by * gridDim.x + bx) * DP + d] as in the kernels above; rows per workgroup: 256 / SPLIT.
template <int KIND, int DP, int PREC, bool CLAMP, int SPLIT>
__global__ __launch_bounds__(256, 2) void grad_kff_mid_kernel(const double* __restrict__ Xh, const double* __restrict__ ah, const double* __restrict__ u,
                                                              const double* __restrict__ v, const double* __restrict__ uc, const double* __restrict__ vc,
                                                              int64_t n, int64_t jchunk, int rb_stride, int rb_offset, double* __restrict__ part,
                                                              const double* __restrict__ exp_tab, double bias) {
    using T = double;
    constexpr int HD = DP / SPLIT, CH = 8, NCH = HD / CH, NQ = (DP + 7) / 8, WROWS = 64 / SPLIT;
    static_assert((SPLIT == 2 || SPLIT == 4) && HD % CH == 0, "part width must be a multiple of 8");
    constexpr bool FOLD = (KIND == CGLB_RBF) && !CLAMP;
    constexpr bool BIASED = (KIND != CGLB_RBF) && !CLAMP && PREC != CGLB_PREC_EXACT;
    __shared__ double smem[16];
    __shared__ double tab[CGLB_TAB_SIZE];
    __shared__ T trbuf[4 * 8 * GRAD_TR_LD];
    // DP = 96 with a row shared by two lanes needs 265 registers with the second-moment accumulators in VGPRs - one wave per SIMD; kept in
    // LDS instead (a private column of 12 slots per lane: no conflicts, no barrier; 36 LDS operations per batch of 8 columns) it fits 2 waves
    constexpr bool G2LDS = (DP > 80 && SPLIT == 2);
    __shared__ T g2buf[G2LDS ? 4 * NQ * 64 : 1];
    load_exp_table(tab, exp_tab);
    const int lane = threadIdx.x & 63, prt = lane / WROWS, l8 = lane & 7;   // prt: which 1 / SPLIT of the dimensions this lane holds
    T* __restrict__ g2 = g2buf + (G2LDS ? (threadIdx.x >> 6) * (NQ * 64) + lane : 0);
    T* __restrict__ tr = trbuf + (threadIdx.x >> 6) * (8 * GRAD_TR_LD);
    const int64_t rblock = ((int64_t)blockIdx.x * rb_stride + rb_offset) * (4 * WROWS);
    const int64_t row = rblock + (threadIdx.x >> 6) * WROWS + (lane & (WROWS - 1));
    const int64_t rr = row < n ? row : n - 1;
    T xi[HD], S1[HD], G2[G2LDS ? 1 : NQ];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        xi[d] = Xh[rr * DP + prt * HD + d];
        S1[d] = 0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if constexpr (G2LDS) g2[q * 64] = 0;
        else G2[q] = 0;
    }
    T S0 = 0;
    const T a = ah[rr];
    const T aseed = prt ? T(0) : ((KIND == CGLB_RBF) ? a : (BIASED ? T(-0.5) * (a + bias) : T(-0.5) * a));  // the seed enters the sum of the parts once
    const T hscale = (KIND == CGLB_RBF) ? T(1) : T(3);
    const T ui = row < n ? hscale * u[rr] : T(0);
    const T vi = row < n ? hscale * v[rr] : T(0);
    int64_t j0 = (int64_t)blockIdx.y * jchunk;
    const int64_t j1 = (j0 + jchunk < n) ? j0 + jchunk : n;
    const int64_t sym_from = rblock + 4 * WROWS;
    if (j0 < rblock) j0 = rblock;
    const T* __restrict__ xcol = Xh + prt * HD + l8;   // + j * DP + 8 s: this lane's element of slice s of column j
    T xsl[NCH];
    if (j0 < j1) {
#pragma unroll
        for (int sl = 0; sl < NCH; ++sl) xsl[sl] = xcol[j0 * DP + sl * CH];
    }
    for (int64_t jb = j0; jb < j1; jb += 8) {
        T t[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int64_t jc = jb + jj;
            t[jj] = 0;
            if (jc < j1) {  // wave-uniform: only the last batch of a chunk is short
                const int64_t jn = (jc + 1 < j1) ? jc + 1 : jc;  // next column (or a harmless re-read)
                const T vj = vc[jc];
                const T wu = (jc >= sym_from) ? uc[jc] : T(0);
                const T aj = FOLD ? T(0) : ah[jc];
                T g = aseed, g1 = T(0);   // two interleaved chains: the dependent-issue latency of one (~9 cycles) is not covered by two waves
#pragma unroll
                for (int sl = 0; sl < NCH; ++sl) BcastChain2<0, CH>::run(g, g1, xsl[sl], &xi[sl * CH]);
                g += g1;
                if constexpr (SPLIT == 4) g = sum_row_pairs(g);
                g = sum_halves(g);
                const T earg[1] = {(KIND == CGLB_RBF) ? (FOLD ? g : g + aj) : sqrt_hot<PREC, BIASED>(tfma<T>(T(-2), g, aj))};
                T h[1];
                exp2_tab_batch<CLAMP, KIND != CGLB_RBF, PREC, 1>(earg, tab, h);
                T w = ui * vj;
                w = tfma<T>(vi, wu, w);
                const T hv = h[0] * w;
                S0 += hv;
                t[jj] = prt ? T(0) : hv;
#pragma unroll
                for (int sl = 0; sl < NCH; ++sl) {
                    BcastAxpy<0, CH>::run(&S1[sl * CH], xsl[sl], hv);
                    xsl[sl] = xcol[jn * DP + sl * CH];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // second moments through the column sums of the pair weights, as in grad_kff_gram_kernel; the squares are formed here
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
        const T* __restrict__ xq = Xh + (jcol < j1 ? jcol : j1 - 1) * DP;  // columns past the chunk carry cs == 0
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = (lane >> 3) + 8 * q;
            if (d < DP) {
                const T x = xq[d];
                if constexpr (G2LDS) g2[q * 64] = tfma<T>(cs, x * x, g2[q * 64]);
                else G2[q] = tfma<T>(cs, x * x, G2[q]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        double s = ((lane >> 3) == (d & 7)) ? (G2LDS ? g2[(d >> 3) * 64] : G2[G2LDS ? 0 : (d >> 3)]) : 0.0;
        const int dl = d % HD;     // the part that holds dimension d adds the row terms
        if (prt == d / HD) s += xi[dl] * (xi[dl] * S0 - 2.0 * S1[dl]);
        s = block_sum(s, smem);
        if (threadIdx.x == 0) part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * DP + d] = s;
    }
}

// out[d] = var * scale_d / hot^2 * sum_b part[b * DP + d], scale_d = 1 / (l_d kscale^2) from the device copy (kernels_wide.hip: upload_scales)
__global__ __launch_bounds__(256) void grad_dl_finalize_mid_kernel(const double* __restrict__ part, int64_t nblk, int DP, int D, const double* __restrict__ scale,
                                                                   double inv_hot2, double var, double* __restrict__ out) {
    __shared__ double smem[16];
    const int d = blockIdx.x;
    if (d >= D) return;
    double s = 0.0;
    for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) s += part[b * DP + d];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[d] = s * var * scale[d] * inv_hot2;
}


// Lanes sharing a matrix row at padded width 96: 2 with the second moments in LDS (16.8 ms at N = 50k; 24.7 with 4).  The Matern-3/2
// instance of the exact level (two-step square root) does not fit 256 registers that way (101 spilled: 41 ms) and keeps 4 (25.5 ms).
static inline int grad_mid_split(const cglb_ctx* c) {
    if (c->Dh != 96) return 2;
    return (c->kind != CGLB_RBF && c->precision == CGLB_PREC_EXACT) ? 4 : 2;
}
// Mid-width contexts: the whole symmetric form register-resident (rb_stride / rb_offset: the cyclic deal of the 128-row blocks)
int launch_grad_kff_mid(cglb_ctx* c, const void* v_full, const void* u_full, int world, int rank, double* out_dl) {
    bool fold = false;
    CGLB_TRY(grad_fold_operands(c, v_full, u_full, 0, c->N, &fold));
    const int split = grad_mid_split(c);
    const int brows = 256 / split;   // rows of a workgroup
    const int64_t nb = (c->N + brows - 1) / brows;
    const int64_t bx = rank < nb ? (nb - rank + world - 1) / world : 0;
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
    CGLB_TRY(ensure_gpart(c, (size_t)nblk * c->Dh * sizeof(double)));
    dim3 grid((unsigned)bx, (unsigned)jsplit);
    const bool lowprec = c->precision != CGLB_PREC_EXACT;   // level 2 runs as level 1 here
#define GM_LAUNCH1(DPV, PR, CL) do { if (DPV == 96 && split == 4) { GM_LAUNCH0(DPV, PR, CL, (DPV == 96 ? 4 : 2)); } else { GM_LAUNCH0(DPV, PR, CL, 2); } } while (0)
#define GM_LAUNCH0(DPV, PR, CL, SP)                                                                                                                    \
    hipLaunchKernelGGL((grad_kff_mid_kernel<KIND, DPV, PR, CL, SP>), grid, dim3(256), 0, c->stream, (const double*)c->Xh, (const double*)c->xah,         \
                       (const double*)u_full, (const double*)v_full, (CL || !fold) ? (const double*)u_full : (const double*)c->uwh,                    \
                       (CL || !fold) ? (const double*)v_full : (const double*)c->pwh, c->N, jchunk, world, rank, c->gpart, (const double*)c->exp_tab, \
                       c->m32_bias)
#define GM_LAUNCH(DPV) do { if (c->exp_clamp) { if (lowprec) GM_LAUNCH1(DPV, CGLB_PREC_FAST, true); else GM_LAUNCH1(DPV, CGLB_PREC_EXACT, true); }     \
                            else { if (lowprec) GM_LAUNCH1(DPV, CGLB_PREC_FAST, false); else GM_LAUNCH1(DPV, CGLB_PREC_EXACT, false); } } while (0)
    CGLB_DISPATCH_KIND(c->kind, {
        switch (c->Dh) {
            case 48: GM_LAUNCH(48); break;
            case 64: GM_LAUNCH(64); break;
            case 80: GM_LAUNCH(80); break;
            case 96: GM_LAUNCH(96); break;
            default: return cglb_fail(c, CGLB_ERR_BAD_ARG, "unsupported mid width");
        }
    });
#undef GM_LAUNCH
#undef GM_LAUNCH1
#undef GM_LAUNCH0
    CGLB_LAUNCH_CHECK(c);
    const double hot = cglb_hot_scale(c);
    hipLaunchKernelGGL(grad_dl_finalize_mid_kernel, dim3(c->D), dim3(256), 0, c->stream, (const double*)c->gpart, nblk, c->Dh, c->D, (const double*)c->wsmall,
                       1.0 / (hot * hot), c->var, out_dl);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

