// K1, fp64 fast path — implicit K_ff mat-vec with the pair exponent on the matrix pipe.
//
// The per-pair cost of the plain kernel (kernels_kff.hip) is ~24 vector-fp64 instructions, 9 of them the
// Gram chain a_i + a_j + xs_i.xs_j.  That chain is a true contraction over k = D+2 (the two norm terms
// ride along as two extra "dimensions": x~_i = [xs_i, a_i, 1], y~_j = [xs_j, 1, a_j]), so it goes to
// v_mfma_f64_16x16x4_f64: one 16x16 tile of exponents per ceil((D+2)/4) MFMAs, issued on the matrix pipe
// which runs concurrently with the VALU pipe.  What is left on the VALU per pair is 2^x (14 instr) and the
// accumulate fma: 15 instead of 24.
//
// Fragment layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md, 3): A: lane l holds A[l&15][l>>4],
// B: lane l holds B[l>>4][l&15], C/D: 4 doubles per lane, D[(l>>4) + 4*reg][l&15].
// The operands are pre-permuted into that order by frag_prep_kernel, so every fragment load is one fully
// coalesced 512-B wave load:  F[(block16 * KA + kstep) * 64 + lane] = vec[block16*16 + (lane&15)][4*kstep + (lane>>4)].
//
// Each wave owns RB*16 rows (A fragments + RB*4 row accumulators live in VGPRs for the whole launch) and
// streams 16-column sub-tiles of its column chunk.  Row sums are reduced over the 16 lanes of a group once,
// at the end.  Column chunks -> partial slab [jsplit][nrows], combined in fixed order (reproducible).
#include "devmath.h"
#include "dispatch.h"

typedef double double4v __attribute__((ext_vector_type(4)));

// ---- fragment preparation ---------------------------------------------------------------------------------
// side 0 (rows, A operand): x~ = [c*xs, a, 1]   side 1 (cols, B operand): y~ = [xs, 1, a]
// RBF: c = 1, a = -|xs|^2/2 (as stored in xa).  Matern32: c = -2, a = |xs|^2  ->  x~.y~ = |xs_i - xs_j|^2.
template <int KIND>
__global__ __launch_bounds__(256) void frag_prep_kernel(const double* __restrict__ Xs, const double* __restrict__ xa, int64_t n, int D, int DP,
                                                        int KA, int side, double* __restrict__ F, int64_t nblk16) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nblk16 * KA * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t bs = idx >> 6;
    const int s = (int)(bs % KA);
    const int64_t b = bs / KA;
    const int64_t row = b * 16 + (lane & 15);
    const int k = 4 * s + (lane >> 4);
    double v = 0.0;
    if (row < n) {
        if (k < D) {
            v = Xs[row * DP + k];
            if (side == 0 && KIND == CGLB_MATERN32) v *= -2.0;
        } else if (k == D) {
            v = (side == 0) ? xa[row] : 1.0;
        } else if (k == D + 1) {
            v = (side == 0) ? 1.0 : xa[row];
        }
    }
    F[idx] = v;
}

int launch_frag_prep(cglb_ctx* c) {
    if (c->dtype != CGLB_F64) return CGLB_OK;
    const int KA = (c->D + 2 + 3) / 4;
    const int64_t nblk16 = (c->N + 15) / 16;
    const size_t need = (size_t)nblk16 * KA * 64 * sizeof(double);
    if (need > c->frag_cap) {
        if (c->fragA) HIP_CHECK(c, hipFree(c->fragA));
        if (c->fragB) HIP_CHECK(c, hipFree(c->fragB));
        c->fragA = c->fragB = nullptr;
        HIP_CHECK(c, hipMalloc(&c->fragA, need));
        HIP_CHECK(c, hipMalloc(&c->fragB, need));
        c->frag_cap = need;
    }
    const int64_t total = nblk16 * KA * 64;
    const int grid = (int)((total + 255) / 256);
    for (int side = 0; side < 2; ++side) {
        double* F = (double*)(side == 0 ? c->fragA : c->fragB);
        if (c->kind == CGLB_RBF)
            hipLaunchKernelGGL((frag_prep_kernel<CGLB_RBF>), dim3(grid), dim3(256), 0, c->stream, (const double*)c->Xs, (const double*)c->xa, c->N,
                               c->D, c->Dp, KA, side, F, nblk16);
        else
            hipLaunchKernelGGL((frag_prep_kernel<CGLB_MATERN32>), dim3(grid), dim3(256), 0, c->stream, (const double*)c->Xs, (const double*)c->xa,
                               c->N, c->D, c->Dp, KA, side, F, nblk16);
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

// ---- the mat-vec kernel -------------------------------------------------------------------------------------
template <int KIND, bool CLAMP> __device__ __forceinline__ double kappa_from_exponent(double t) {
    if (KIND == CGLB_RBF) {
        return exp2_hot<CLAMP>(t);  // t = a_i + a_j + xs_i.xs_j  (<= 0 up to round-off)
    } else {
        const double r = sqrt_pos(fmax(t, 0.0));  // t = |xs_i - xs_j|^2 in the scaled units
        return __builtin_fma(r, CGLB_LN2, 1.0) * exp2_hot<CLAMP>(-r);
    }
}

// grid: (ceil(nrows / (64*RB)), jsplit); block: 256 threads = 4 waves, wave w owns row blocks [..] of 16*RB rows.
// FA: row-side fragments starting at the first local row block (row0 must be a multiple of 16).
template <int KIND, int KA, int RB, bool CLAMP>
__global__ __launch_bounds__(256) void kff_mfma_kernel(const double* __restrict__ FA, int64_t nrows, const double* __restrict__ FB,
                                                       const double* __restrict__ p, int64_t N, int64_t jtiles_per_chunk,
                                                       double* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t rblk0 = ((int64_t)blockIdx.x * 4 + wave) * RB;  // first 16-row block of this wave
    const int64_t nrblk = (nrows + 15) >> 4;
    double a[RB][KA];
    double4v acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t blk = rblk0 + rb < nrblk ? rblk0 + rb : nrblk - 1;  // clamp: duplicates are dropped at the store
#pragma unroll
        for (int s = 0; s < KA; ++s) a[rb][s] = FA[(blk * KA + s) * 64 + lane];
        acc[rb] = (double4v){0.0, 0.0, 0.0, 0.0};
    }
    const int64_t njt = (N + 15) >> 4;
    const int64_t jt0 = (int64_t)blockIdx.y * jtiles_per_chunk;
    const int64_t jt1 = jt0 + jtiles_per_chunk < njt ? jt0 + jtiles_per_chunk : njt;
    const int cidx = lane & 15;
    // software prefetch of the next column sub-tile
    double bcur[KA], pcur = 0.0;
    if (jt0 < jt1) {
#pragma unroll
        for (int s = 0; s < KA; ++s) bcur[s] = FB[(jt0 * KA + s) * 64 + lane];
        const int64_t j = jt0 * 16 + cidx;
        pcur = j < N ? p[j] : 0.0;
    }
    for (int64_t jt = jt0; jt < jt1; ++jt) {
        double bnext[KA], pnext = 0.0;
        const int64_t jn = jt + 1 < jt1 ? jt + 1 : jt;
#pragma unroll
        for (int s = 0; s < KA; ++s) bnext[s] = FB[(jn * KA + s) * 64 + lane];
        {
            const int64_t j = jn * 16 + cidx;
            pnext = j < N ? p[j] : 0.0;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            double4v t = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KA; ++s) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb][s], bcur[s], t, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[rb][q] = __builtin_fma(kappa_from_exponent<KIND, CLAMP>(t[q]), pcur, acc[rb][q]);
        }
#pragma unroll
        for (int s = 0; s < KA; ++s) bcur[s] = bnext[s];
        pcur = pnext;
    }
    // row sums: reduce over the 16 lanes of each lane group (the column index), then lane (l&15)==0 stores
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = acc[rb][q];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            const int64_t row = (rblk0 + rb) * 16 + (lane >> 4) + 4 * q;
            if (cidx == 0 && rblk0 + rb < nrblk && row < nrows) part[(int64_t)blockIdx.y * nrows + row] = v;
        }
    }
}

template <int KIND, int KA>
static int mfma_launch(cglb_ctx* c, const double* FA, int64_t nrows, const double* p_full, double* part, int64_t* jsplit_out) {
    constexpr int RB = 4;
    const int64_t bx = (nrows + 64 * RB * 4 - 1) / (64 * RB * 4) * 1;  // 4 waves x RB*16 rows = 256 rows per block (RB = 4)
    const int64_t rows_per_block = 4 * RB * 16;
    const int64_t gx = (nrows + rows_per_block - 1) / rows_per_block;
    (void)bx;
    const int64_t njt = (c->N + 15) / 16;
    int64_t jsplit = c->kff_jsplit > 0 ? c->kff_jsplit : (8192 + gx - 1) / gx;
    if (jsplit > 512) jsplit = 512;
    if (jsplit > njt) jsplit = njt;
    if (jsplit < 1) jsplit = 1;
    const int64_t jtpc = (njt + jsplit - 1) / jsplit;
    jsplit = (njt + jtpc - 1) / jtpc;
    *jsplit_out = jsplit;
    dim3 grid((unsigned)gx, (unsigned)jsplit);
    if (c->exp_clamp)
        hipLaunchKernelGGL((kff_mfma_kernel<KIND, KA, RB, true>), grid, dim3(256), 0, c->stream, FA, nrows, (const double*)c->fragB, p_full, c->N,
                           jtpc, part);
    else
        hipLaunchKernelGGL((kff_mfma_kernel<KIND, KA, RB, false>), grid, dim3(256), 0, c->stream, FA, nrows, (const double*)c->fragB, p_full, c->N,
                           jtpc, part);
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

// Launches the pair kernel for the local row shard (rows r0..r1 of the training set) into c->kpart; returns jsplit.
// Requires r0 % 16 == 0 (row fragments are addressed by 16-row blocks).
int launch_kff_mfma_pairs(cglb_ctx* c, const double* p_full, int64_t* jsplit_out) {
    if (c->dtype != CGLB_F64 || (c->r0 & 15)) return cglb_fail(c, CGLB_ERR_STATE, "mfma path needs fp64 and a 16-aligned row shard");
    const int KA = (c->D + 2 + 3) / 4;
    const int64_t nrows = c->nloc;
    const int64_t njt = (c->N + 15) / 16;
    const int64_t rows_per_block = 256;
    const int64_t gx = (nrows + rows_per_block - 1) / rows_per_block;
    int64_t js = c->kff_jsplit > 0 ? c->kff_jsplit : (8192 + gx - 1) / gx;
    if (js > 512) js = 512;
    if (js > njt) js = njt;
    const size_t need = (size_t)js * nrows * sizeof(double);
    if (need > c->kpart_cap) {
        if (c->kpart) HIP_CHECK(c, hipFree(c->kpart));
        c->kpart = nullptr;
        HIP_CHECK(c, hipMalloc(&c->kpart, need));
        c->kpart_cap = need;
    }
    if (!c->frag_valid) {
        CGLB_TRY(launch_frag_prep(c));
        c->frag_valid = true;
    }
    const double* FA = (const double*)c->fragA + (c->r0 / 16) * KA * 64;
    double* part = (double*)c->kpart;
#define MF(KAV)                                                                                      \
    case KAV:                                                                                        \
        if (c->kind == CGLB_RBF) return mfma_launch<CGLB_RBF, KAV>(c, FA, nrows, p_full, part, jsplit_out); \
        return mfma_launch<CGLB_MATERN32, KAV>(c, FA, nrows, p_full, part, jsplit_out);
    switch (KA) {
        MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(8) MF(9)
        default: return cglb_fail(c, CGLB_ERR_BAD_ARG, "unsupported dimension for the mfma path");
    }
#undef MF
}
