// K1, symmetric form — the square block K[rows, rows] of the implicit mat-vec, each kernel value used twice.
//
// K_ff is symmetric, so for i < j one evaluation of kappa_ij serves out_i += kappa_ij p_j and out_j += kappa_ij p_i.
// The pair evaluation (Gram chain + 2^x, ~22 vector-fp64 instructions) dominates, so halving the number of
// evaluations is worth one extra fma per pair plus a cross-lane reduction:
//   * a wave owns 64*R rows (R per lane, operands in VGPRs) and streams a chunk of columns j >= its first row;
//     x_j / a_j / p_j are wave-uniform scalar loads exactly as in the plain kernel (kernels_kff.hip);
//   * row sums stay in the lane that owns the row (no reduction);
//   * the column contribution sum_i kappa_ij p_i of a column j is a sum ACROSS lanes.  Columns are taken in
//     batches of 16; the 16 per-lane partials are transposed through a small LDS scratch (8 columns at a time) so that
//     each lane adds 8 lanes of one column and three shuffles finish the sum: ~1 fp64 instruction per column and lane
//     instead of 18 for sixteen separate wave reductions (an in-register transpose-reduce, round 1, cost ~5);
//   * nothing is accumulated with atomics: every (slot, element) of the partial slabs is written by exactly one
//     workgroup and the combine kernel adds the valid slots in fixed order, so results are bitwise reproducible.
// Slab layout (element type double), n = number of rows of the square block:
//   Prow[k][li], k < nchunk : row sums of chunk k (valid for k >= first chunk of row block rb(i)); li = compact index of row i among
//                            the row blocks of this rank (all rows on one GPU), leading dimension prow_ld
//   Pcol[g][j], g < ceil(nrb/4) : column sums produced by the GROUP of row blocks 4g .. 4g+3 of this rank (one workgroup: its four
//                            waves stage their column sums in LDS and the workgroup stores their fixed-order sum); valid for the
//                            groups that hold a block left of rb(j); blocks at or right of rb(j) contribute exact zeros
// The diagonal 64R x 64R blocks are evaluated in full and contribute row sums only.
//
// Folded column norm (RBF, unclamped range): kappa_ij = 2^(a_i + x_i.x_j) * w_j with w_j = 2^(a_j), so the per-pair add of
// a_j is dropped: the row sums use the pre-weighted operand pw_j = p_j w_j (one N-element kernel per mat-vec) and the 16
// column sums of a batch are multiplied by w_j after the cross-lane reduction.  The unweighted factor reaches
// 2^(|x_j|^2/2); cglb_set_hypers keeps the whole exponent range inside +-1000 octaves (fp64) / +-100 (fp32) or selects CLAMP.
#include "devmath.h"
#include "dispatch.h"
#include <algorithm>

#define SYM_BATCH 16
typedef float sym_f2 __attribute__((ext_vector_type(2)));  // two rows of a lane side by side: v_pk_fma_f32 (fp32 path)

#define SYM_CHUNK_MAX 1024  // column chunk of a work item: at most this many columns (their sums are staged in LDS)
#ifndef CGLB_SYM_TR_REG
#define CGLB_SYM_TR_REG 0
#endif
// minimum waves per SIMD asked of the compiler for the fp64 instances (padded width DP, R rows per lane); 1 = no constraint
#define CGLB_SYM_WAVES(DP, R) (((DP) <= 8 && (R) <= 4) ? 3 : ((DP) == 32 ? 4 : ((DP) == 96 ? CGLB_SYM_WAVES_96 : 1)))
#ifndef CGLB_SYM_BCAST_DP
#define CGLB_SYM_BCAST_DP 33   // narrowest padded width (fp64) whose column operands travel in VGPRs and reach the fma by row_newbcast (below)
#endif
#ifndef CGLB_SYM_WAVES_96
#define CGLB_SYM_WAVES_96 2
#endif
#ifndef CGLB_SYM_R4_MAX_DP
#define CGLB_SYM_R4_MAX_DP 12  // widest padded row that still gets 4 rows per lane in fp64 (DP = 12: 3.67 -> 3.40 ms at N = 100k against 2 rows)
#endif
#ifndef CGLB_SYM_LATE_DP
#define CGLB_SYM_LATE_DP 24  // padded row width from which the column operands are fetched after the Gram chain (below)
#endif
// column sums of a batch: in-register transpose-reduce (true) or through the LDS scratch (false).  One row per lane took the register form
// because those instances ran 4 waves per SIMD; the broadcast-operand instances (2-3 waves) have the LDS to spare (CGLB_SYM_MID_TR_REG=1: A/B).
#ifndef CGLB_SYM_MID_TR_REG
#define CGLB_SYM_MID_TR_REG 0
#endif
#define SYM_TR_IN_REG(T, DP, R) (CGLB_SYM_TR_REG || ((R) == 1 && !(sizeof(T) == 8 && (DP) >= CGLB_SYM_BCAST_DP && !CGLB_SYM_MID_TR_REG)))
#define SYM_TR_LD 65         // leading dimension of the 8 x 64 transposition scratch of a wave (odd: the column reads spread over the banks)

// One work item: rows of block `rb` against the columns of chunk `k` that lie at or right of the block's first row.  Row sums go to
// Prow; the column sums (transposed use of the kernel values) go to `cs`, this wave's LDS array indexed by column - k * chunk.
template <typename T, int KIND, int DP, int R, bool CLAMP, int PREC>
__device__ __forceinline__ void kff_sym_item(const T* __restrict__ Xs, const T* __restrict__ xa, const T* __restrict__ p, const T* __restrict__ pc,
                                             int64_t row0, int64_t n, int64_t chunk, int64_t rb, int64_t k, int64_t cslot, int64_t prow_ld,
                                             T* __restrict__ Prow, T* __restrict__ cs, T* __restrict__ tr, const double* __restrict__ tab, int lane, T bias) {
    constexpr bool FOLD = (KIND == CGLB_RBF) && !CLAMP;
    // Matern-3/2, fast level, unclamped range: squared distances kept positive by a bias in the row seeds instead of a clamp per pair
    constexpr bool BIASED = (KIND != CGLB_RBF) && !CLAMP && PREC != CGLB_PREC_EXACT && sizeof(T) == 8;
    constexpr int RBROWS = 64 * R;
    const int64_t rbase = rb * RBROWS;
    // fp32: rows in pairs, so that the Gram chain and the two accumulations run as v_pk_fma_f32 (4.8 nominal cycles per pair of
    // fma against 2 x 2.94 unpacked); the column operand is broadcast to both halves through op_sel straight from the SGPR
    constexpr bool PACKED = (sizeof(T) == 4) && (R % 2 == 0);
    // Mid width (32 < DP <= 96, fp64, R = 1): the row operand fills 2 DP VGPRs and a column's operands no longer fit the scalar register
    // file (first attempt: 16-wide slices through a double buffer of 64 SGPRs - the scalar loads of a 650-KB column chunk miss the scalar
    // cache and one slice of lead, 64 cycles, does not cover them: 11.4 ms at N = 50k, D = 77 against 11.8 through the Gram tiles).  The
    // column operands therefore travel in VGPRs: slice s of a column is ONE register pair, lane l holding coordinate 16 s + l % 16 (a vector
    // load of 128 bytes), and the fma takes coordinate k from lane k of its 16-lane row (devmath.h: fmac_bcast).  Vector loads return in
    // order, so the compiler's vmcnt waits for the look-ahead are exact.
    constexpr bool MID = sizeof(T) == 8 && DP >= CGLB_SYM_BCAST_DP;
    constexpr int CH = !MID ? 16 : (DP % 16 == 0 ? 16 : (DP % 8 == 0 ? 8 : 4));   // coordinates per slice register (a divisor of DP)
    constexpr int NCH = MID ? DP / CH : 1, DJ = MID ? 1 : DP;
#ifndef CGLB_SYM_MID_CHAINS2_DP
#define CGLB_SYM_MID_CHAINS2_DP 33
#endif
    constexpr bool MID2 = MID && DP >= CGLB_SYM_MID_CHAINS2_DP;   // two interleaved chains instead of one dependent chain
    static_assert(!MID || (R == 1 && DP % CH == 0), "broadcast-operand instances: fp64, one row per lane");
    const int l16 = lane & (CH - 1);
    constexpr int RP = PACKED ? R / 2 : 1;
    constexpr int RU = PACKED ? 1 : R;  // the unpacked row operands exist only on the other path
    T xi[RU][DP], ai[R], pr[R], acc[R];
    sym_f2 xi2[RP][DP], ai2[RP], pr2[RP], acc2[RP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = rbase + r * 64 + lane;
        const int64_t rr = row < n ? row : n - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const T v = Xs[(row0 + rr) * DP + d];
            if (PACKED) xi2[r / 2][d][r % 2] = (float)v;
            else xi[r % RU][d] = v;
        }
        const T a = xa[row0 + rr];
        ai[r] = (KIND == CGLB_RBF) ? a : (BIASED ? T(-0.5) * (a + bias) : T(-0.5) * a);
        pr[r] = row < n ? p[row0 + rr] : T(0);
        acc[r] = 0;
    }
    if (PACKED) {
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            ai2[rp] = sym_f2{(float)ai[2 * rp], (float)ai[2 * rp + 1]};
            pr2[rp] = sym_f2{(float)pr[2 * rp], (float)pr[2 * rp + 1]};
            acc2[rp] = sym_f2{0.f, 0.f};
        }
    }
    int64_t j0 = k * chunk;
    if (j0 < rbase) j0 = rbase;
    int64_t j1 = (k + 1) * chunk;
    if (j1 > n) j1 = n;
    const int64_t sym_from = rbase + RBROWS;  // columns at or beyond this get the transposed contribution
    // j0 and chunk are multiples of 16, so only the very last batch of the block (j1 == n) can be short
    const int64_t jfull = j0 + ((j1 - j0) / SYM_BATCH) * SYM_BATCH;
    // Column operands are software-pipelined one column ahead: the scalar loads of column j+1 are issued before the
    // arithmetic of column j, so their latency (and that of the table reads, which share the lgkm counter) is covered.
    T xj[DJ], aj = T(0), pj;
    T xsl[NCH];  // MID: the slices of the current column
    if (j0 < jfull) {
        const int64_t j = row0 + j0;
        if (!FOLD) aj = xa[j];
        pj = pc[j];
        if constexpr (MID) {
#pragma unroll
            for (int sl = 0; sl < NCH; ++sl) xsl[sl] = Xs[j * DP + sl * CH + l16];
        } else {
#pragma unroll
            for (int d = 0; d < DP; ++d) xj[d] = Xs[j * DP + d];
        }
    }
    for (int64_t jb = j0; jb < jfull; jb += SYM_BATCH) {
        const T* __restrict__ xsj = Xs + (row0 + jb) * DP;  // wave-uniform bases: s_load with immediate offsets
        const T* __restrict__ xaj = xa + row0 + jb;
        const T* __restrict__ pjv = pc + row0 + jb;
        const int64_t nb = (jb + SYM_BATCH < jfull) ? SYM_BATCH : 0;  // first column of the next batch (or a harmless re-read)
        T t[SYM_BATCH];
#pragma unroll
        for (int jj = 0; jj < SYM_BATCH; ++jj) {
            // prefetch the next column.  D <= 16: before the Gram chain (a whole column ahead).  Wider rows: AFTER the chain, into the
            // registers its operands just left - two columns of operands (4 DP SGPRs) do not fit the scalar register file, and the
            // compiler then parks them in VGPR lanes (D = 32: 581 spilled SGPRs, 70 v_readlane/v_writelane per column: 6.4 ms for a
            // mat-vec at N = 60k that takes 3.2 without); the loads then have the 2^x and the accumulation of the column to land.
            constexpr bool LATE = DP >= CGLB_SYM_LATE_DP;
            T xn[DJ], an = T(0), pn;
            const int64_t o = (jj + 1 < SYM_BATCH) ? jj + 1 : nb;
            if constexpr (MID) {
                if (!FOLD) an = xaj[o];
                pn = pjv[o];
            } else if (!LATE) {
                if (!FOLD) an = xaj[o];
                pn = pjv[o];
#pragma unroll
                for (int d = 0; d < DP; ++d) xn[d] = xsj[o * DP + d];
                __builtin_amdgcn_sched_barrier(0);  // issue the prefetch first; it is consumed a whole column later
            }
            T gram[R];
            if constexpr (MID) {
                // Slice s of the NEXT column is requested into the register pair slice s of this column just left: it is needed one whole
                // column later, and no second set of registers is spent on the look-ahead.
                if constexpr (MID2) {
                    T g0 = ai[0], g1 = T(0);
#pragma unroll
                    for (int sl = 0; sl < NCH; ++sl) {
                        BcastChain2<0, CH>::run(g0, g1, xsl[sl], &xi[0][sl * CH]);
                        xsl[sl] = xsj[o * DP + sl * CH + l16];
                    }
                    gram[0] = g0 + g1;
                } else {
                    T g = ai[0];
#pragma unroll
                    for (int sl = 0; sl < NCH; ++sl) {
                        BcastChain<0, CH>::run(g, xsl[sl], &xi[0][sl * CH]);
                        xsl[sl] = xsj[o * DP + sl * CH + l16];
                    }
                    gram[0] = g;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (PACKED) {
#pragma unroll
                for (int rp = 0; rp < RP; ++rp) {
                    sym_f2 g = ai2[rp];
#pragma unroll
                    for (int d = 0; d < DP; ++d) g = __builtin_elementwise_fma(xi2[rp][d], sym_f2{(float)xj[d], (float)xj[d]}, g);
                    gram[2 * rp] = (T)g.x;
                    gram[2 * rp + 1] = (T)g.y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    T g = ai[r];
#pragma unroll
                    for (int d = 0; d < DP; ++d) g = tfma<T>(xi[r % RU][d], xj[d], g);
                    gram[r] = g;
                }
            }
            if constexpr (LATE && !MID) {
                __builtin_amdgcn_sched_barrier(0);
                if (!FOLD) an = xaj[o];
                pn = pjv[o];
#pragma unroll
                for (int d = 0; d < DP; ++d) xn[d] = xsj[o * DP + d];
                __builtin_amdgcn_sched_barrier(0);
            }
            KappaPend<T> kp[R];
            kappa_hot_begin_batch<T, KIND, CLAMP, FOLD || BIASED, PREC, R>(gram, aj, tab, kp);
            __builtin_amdgcn_sched_barrier(0);  // all R table reads are in flight here ...
#pragma unroll
            for (int r = 0; r < R; ++r) kappa_hot_poly<T, KIND, PREC>(kp[r]);
            __builtin_amdgcn_sched_barrier(0);  // ... and are first needed here, R polynomials later
            T tj = 0;
            if (PACKED) {
                sym_f2 tj2 = {0.f, 0.f};
#pragma unroll
                for (int rp = 0; rp < RP; ++rp) {
                    const sym_f2 kap2 = {(float)kappa_hot_end<T, KIND>(kp[2 * rp]), (float)kappa_hot_end<T, KIND>(kp[2 * rp + 1])};
                    acc2[rp] = __builtin_elementwise_fma(kap2, sym_f2{(float)pj, (float)pj}, acc2[rp]);
                    tj2 = __builtin_elementwise_fma(kap2, pr2[rp], tj2);
                }
                tj = (T)(tj2.x + tj2.y);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T kap = kappa_hot_end<T, KIND>(kp[r]);
                    acc[r] = tfma<T>(kap, pj, acc[r]);
                    tj = tfma<T>(kap, pr[r], tj);
                }
            }
            t[jj] = tj;
            // Pin the accumulators at the end of every column: without this the optimizer sinks the 2*R accumulate
            // fmas of all 16 unrolled columns below the batch (64 kernel values kept live = 128 extra VGPRs), and the
            // scheduler hoists the scalar operand loads of all 16 columns (320 SGPRs, spills).
            if (PACKED) {
#pragma unroll
                for (int rp = 0; rp < RP; ++rp) asm volatile("" : "+v"(acc2[rp]));
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            }
            asm volatile("" : "+v"(t[jj]));
            __builtin_amdgcn_sched_barrier(0);
            aj = an;
            pj = pn;
            if constexpr (!MID) {
#pragma unroll
                for (int d = 0; d < DP; ++d) xj[d] = xn[d];
            }
        }
        if (jb >= sym_from) {  // wave-uniform
            // Column sums of the batch = sums ACROSS the 64 lanes of t[0..15].
            if constexpr (SYM_TR_IN_REG(T, DP, R)) {
                // R == 1 (D > 16): the in-register transpose-reduce of round 1 (4 select/shuffle/add stages that halve the number of live
                // vectors, then two butterfly adds) - those instances run 4 waves per SIMD, which the LDS scratch of the other form would
                // cut to 3 (D = 24: 2.57 -> 2.91 ms).  Also for A/B builds (EXTRA_DEFS=-DCGLB_SYM_TR_REG=1).
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool hi = (lane >> s) & 1;
#pragma unroll
                    for (int q = 0; q < (SYM_BATCH >> (s + 1)); ++q) {
                        const T a = t[2 * q], b = t[2 * q + 1];
                        const T keep = hi ? b : a;
                        const T send = hi ? a : b;
                        t[q] = keep + __shfl_xor(send, 1 << s, 64);
                    }
                }
                T v = t[0];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lane < SYM_BATCH) cs[jb - k * chunk + lane] = v;
            } else {
                // Transposed through LDS, 8 columns at a time: every lane writes its 8 partials (row jj of `tr`, stride SYM_TR_LD:
                // conflict-free), then lane (c = lane & 7, g = lane >> 3) adds the 8 lanes 8g..8g+7 of column c in fixed order and three
                // xor-shuffles add the 8 groups: 7 + 3 adds and no selects per 32 pairs, against 83 VALU instructions per 64 pairs for the
                // in-register form.  One wave, in-order LDS: no barrier; the wave_barrier calls only pin the compiler's order.
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) tr[jj * SYM_TR_LD + lane] = t[8 * half + jj];
                    __builtin_amdgcn_wave_barrier();
                    const T* __restrict__ src = tr + (lane & 7) * SYM_TR_LD + (lane & ~7);
                    T v = src[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) v += src[i];
                    __builtin_amdgcn_wave_barrier();
                    v += __shfl_xor(v, 8, 64);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane < 8) cs[jb - k * chunk + 8 * half + lane] = v;
                }
            }
        }
    }
    if (PACKED) {  // the packed row sums continue unpacked in the tail / are stored below
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            acc[2 * rp] = (T)acc2[rp].x;
            acc[2 * rp + 1] = (T)acc2[rp].y;
        }
    }
    // ragged tail of the block (fewer than 16 columns): one column at a time, plain wave reduction
    for (int64_t jc = jfull; jc < j1; ++jc) {
        const int64_t j = row0 + jc;
        const T aj = FOLD ? T(0) : xa[j];
        const T pj = pc[j];
        T xj[DJ];
        if constexpr (!MID) {
#pragma unroll
            for (int d = 0; d < DP; ++d) xj[d] = Xs[j * DP + d];
        }
        T tj = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T gram = ai[r];
            if constexpr (MID) {   // no prefetch: at most 15 columns per work item come this way
                T ct[NCH];
#pragma unroll
                for (int sl = 0; sl < NCH; ++sl) ct[sl] = Xs[j * DP + sl * CH + l16];
#pragma unroll
                for (int sl = 0; sl < NCH; ++sl) BcastChain<0, CH>::run(gram, ct[sl], &xi[0][sl * CH]);
            } else {
#pragma unroll
            for (int d = 0; d < DP; ++d) gram = tfma<T>(PACKED ? (T)xi2[(r / 2) % RP][d][r % 2] : xi[r % RU][d], xj[d], gram);
            }
            const T kap = kappa_hot_single<T, KIND, CLAMP, FOLD || BIASED, PREC>(gram, aj, tab);  // FOLD: the weight is in pj / applied below
            acc[r] = tfma<T>(kap, pj, acc[r]);
            tj = tfma<T>(kap, pr[r], tj);
        }
        if (jc >= sym_from) {
            const T v = wave_sum(tj);
            if (lane == 0) cs[jc - k * chunk] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = rbase + r * 64 + lane;
        if (row < n) Prow[k * prow_ld + cslot * RBROWS + r * 64 + lane] = acc[r];  // compact: only this rank's row blocks have rows here
    }
}

// A workgroup = 4 waves = 4 consecutive row blocks of this rank against ONE column chunk (`groups[blockIdx.x]` = (group slot, chunk);
// `items[4 * blockIdx.x + wave]` = (row block or -1, chunk)).  Each wave stages the column sums of its item in LDS; after a barrier
// the workgroup adds the four arrays in fixed order and stores ONE column-sum vector per (group, chunk): a quarter of the slab
// elements, writes and combine-kernel reads of one vector per (row block, chunk).
template <typename T, int KIND, int DP, int R, bool CLAMP, int PREC>
// waves per SIMD: fp32 4 (at D = 16 that costs a 184-byte spill and is still faster than 3 waves without: 203 vs 235 ms per mat-vec at
// N = 1M); fp64 (CGLB_SYM_WAVES): 3 for the 4-rows-per-lane instances of D = 5..8 (160-167 VGPRs), 4 at padded width 32 (130 -> 116 VGPRs:
// 3.43 -> 3.13 ms at N = 60k), unconstrained elsewhere (pins at widths 2, 10 and 16 measured -2 % for RBF and up to +13 % for Matern-3/2)
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 4 : CGLB_SYM_WAVES(DP, R))) void kff_sym_kernel(const T* __restrict__ Xs, const T* __restrict__ xa, const T* __restrict__ p,
                                                      const T* __restrict__ pw, const T* __restrict__ wcol,
                                                      int64_t row0, int64_t n, int64_t chunk, const int2* __restrict__ items,
                                                      const int2* __restrict__ groups, int rb_stride, int64_t prow_ld, T* __restrict__ Prow,
                                                      T* __restrict__ Pcol, const double* __restrict__ exp_tab, T bias) {
    __shared__ double tab[CGLB_TAB_SIZE];
    __shared__ T csum[4 * SYM_CHUNK_MAX];
    __shared__ T trbuf[SYM_TR_IN_REG(T, DP, R) ? 1 : 4 * 8 * SYM_TR_LD];  // transposition scratch of the four waves (unused for R == 1)
    load_exp_table(tab, exp_tab);  // before the early exit below: every thread reaches the barrier inside
    constexpr bool FOLD = (KIND == CGLB_RBF) && !CLAMP;
    const T* __restrict__ pc = FOLD ? pw : p;  // column-side operand
    const int lane = threadIdx.x & 63;
    const int2 grp = groups[blockIdx.x];
    if (__builtin_amdgcn_readfirstlane(grp.x) < 0) return;  // padding workgroup of the XCD-aware order: the whole block leaves together
    // wave-uniform work item: readfirstlane makes that visible to the compiler, so everything derived from it
    // (column indices, operand addresses) lives in SGPRs and the column operands are fetched with scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int2 it = items[blockIdx.x * 4 + wave];
    T* __restrict__ cs = csum + wave * SYM_CHUNK_MAX;
    for (int64_t cidx = lane; cidx < chunk; cidx += 64) cs[cidx] = T(0);  // columns this item does not reach contribute nothing
    if (__builtin_amdgcn_readfirstlane(it.x) >= 0) {
        const int64_t rb = __builtin_amdgcn_readfirstlane(it.x), k = __builtin_amdgcn_readfirstlane(it.y);
        const int64_t cslot = rb / rb_stride;  // compact slot: with a cyclic rank distribution only every rb_stride-th block is here
        kff_sym_item<T, KIND, DP, R, CLAMP, PREC>(Xs, xa, p, pc, row0, n, chunk, rb, k, cslot, prow_ld, Prow, cs, trbuf + (SYM_TR_IN_REG(T, DP, R) ? 0 : wave * 8 * SYM_TR_LD), tab, lane, bias);
    }
    __syncthreads();
    const int64_t gslot = __builtin_amdgcn_readfirstlane(grp.x), k = __builtin_amdgcn_readfirstlane(grp.y);
    for (int64_t cidx = threadIdx.x; cidx < chunk; cidx += 256) {
        const int64_t j = k * chunk + cidx;
        if (j < n) {
            const T sum = (csum[cidx] + csum[SYM_CHUNK_MAX + cidx]) + (csum[2 * SYM_CHUNK_MAX + cidx] + csum[3 * SYM_CHUNK_MAX + cidx]);
            Pcol[gslot * n + j] = FOLD ? sum * wcol[row0 + j] : sum;
        }
    }
}

// out[i] = var * ( sum_s plain[s][i] + sum_{k >= k0(i)} Prow[k][i] + sum_{groups g of row blocks < rb(i)} Pcol[g][i] ) + noise * pdiag[i]
// With a cyclic distribution (world > 1) only the row blocks rb == rank (mod world) were processed here: row sums exist for
// the rows of those blocks, column sums come from those blocks only, and only rank 0 adds the noise term (pdiag == null elsewhere).
// Block = 64 elements x 4 slot groups: wave g adds the slots g, g+4, g+8, ... of each of the three slab lists (8 loads in flight
// per lane, slab rows are n elements apart), the four group sums are added in fixed order through LDS -> bitwise reproducible.
// (One thread per element left a rank of a cyclic run with ~400 dependent slab reads on 1/world of its threads: 46 us at world 8.)
template <typename T>
__global__ __launch_bounds__(256) void kff_sym_combine_kernel(const T* __restrict__ plain, int nplain, const T* __restrict__ Prow, int nchunk,
                                                              int64_t prow_ld, const T* __restrict__ Pcol, int64_t n, int64_t chunk, int rbrows, int world,
                                                              int rank,
                                                              T var, T noise, const T* __restrict__ pdiag, T* __restrict__ out,
                                                              double* __restrict__ dotpart) {
    __shared__ double smem[16];
    __shared__ T gsum[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    double contrib = 0.0;
    T a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n) {
        for (int q = g, u = 0; q < nplain; q += 4, u = (u + 1) & 7) a[u] += plain[(int64_t)q * n + i];
        const int64_t rbi = i / rbrows;
        if (rbi % world == rank) {
            const int64_t k0 = (rbi * rbrows) / chunk;
            const int64_t li = (rbi / world) * rbrows + (i - rbi * rbrows);  // compact row index among this rank's row blocks
            int64_t k = k0 + g;
            for (; k + 28 < nchunk; k += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += CGLB_STREAM_LOAD(Prow + (k + 4 * u) * prow_ld + li);
            }
            for (int u = 0; k < nchunk; k += 4, ++u) a[u] += CGLB_STREAM_LOAD(Prow + k * prow_ld + li);
        }
        {
            // this rank's row blocks before block rbi contributed to column i; they are stored four to a slot (one per workgroup)
            const int64_t nbefore = rbi > rank ? (rbi - rank + world - 1) / world : 0;
            const int64_t ns = (nbefore + 3) / 4;
            int64_t c = g;
            for (; c + 28 < ns; c += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += CGLB_STREAM_LOAD(Pcol + (c + 4 * u) * n + i);
            }
            for (int u = 0; c < ns; c += 4, ++u) a[u] += CGLB_STREAM_LOAD(Pcol + c * n + i);
        }
    }
    gsum[g][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (g == 0 && i < n) {
        const T s = (gsum[0][lane] + gsum[1][lane]) + (gsum[2][lane] + gsum[3][lane]);
        T o = var * s;
        if (pdiag) {
            o = tfma<T>(noise, pdiag[i], o);
            contrib = (double)pdiag[i] * (double)o;
        }
        out[i] = o;
    }
    if (dotpart) {
        const double bs = block_sum(contrib, smem);
        if (threadIdx.x == 0) dotpart[blockIdx.x] = bs;
    }
}

// wh[j] = 2^(xah[j] / T) (once per set_hypers) and pw[j] = p[j] wh[j] (once per mat-vec) for the folded column norm
template <typename T>
__global__ __launch_bounds__(256) void hot_weights_kernel(const T* __restrict__ xah, int64_t n, T* __restrict__ wh) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wh[i] = exp2_neg(xah[i] * T(1.0 / CGLB_HOT_UNITS));
}
template <typename T>
__global__ __launch_bounds__(256) void weight_operand_kernel(const T* __restrict__ p, const T* __restrict__ wh, int64_t n, T* __restrict__ pw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pw[i] = p[i] * wh[i];
}
#ifndef CGLB_SYM_MID_TU
int launch_hot_weights(cglb_ctx* c) {
    if (c->kind != CGLB_RBF || (is_wide(c) && c->Dh == 0)) return CGLB_OK;
    const int grid = (int)((c->N + 255) / 256);
    CGLB_DISPATCH_T(c->dtype, hipLaunchKernelGGL((hot_weights_kernel<T>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->xah, c->N, (T*)c->wh));
    CGLB_LAUNCH_CHECK(c);
    return CGLB_OK;
}

#endif  // !CGLB_SYM_MID_TU

static __global__ __launch_bounds__(256) void finalize_sum_sym_kernel(const double* __restrict__ partials, int n, double* __restrict__ out) {
    __shared__ double smem[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partials[i];
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[0] = s;
}


// Work list of the symmetric kernel.  Unit = a GROUP: 4 consecutive row blocks of this rank (local block index lb = 4g .. 4g+3,
// global rb = rank + lb * world) against one column chunk k - one workgroup, one wave per row block.  A group exists for chunk k if its
// first (leftmost) block reaches into the chunk; a block of the group that lies wholly right of the chunk is a -1 entry.
//   items[4 * b + w] = (rb or -1, k),  groups[b] = (g, k) or (-1, -1) for a padding workgroup.
static int ensure_sym_items(cglb_ctx* c, int64_t n, int rbrows, int64_t chunk, int world, int rank, int* nblocks_out, int* nrb_out,
                            int* nchunk_out) {
    const int nrb = (int)((n + rbrows - 1) / rbrows);
    const int nchunk = (int)((n + chunk - 1) / chunk);
    if (c->sym_items && c->sym_n == n && c->sym_rbrows == rbrows && c->sym_chunk == chunk && c->sym_world == world && c->sym_rank == rank &&
        c->sym_order_built == c->sym_order) {
        *nblocks_out = c->sym_nitems; *nrb_out = nrb; *nchunk_out = nchunk;
        return CGLB_OK;
    }
    const int nlb = rank < nrb ? (nrb - rank + world - 1) / world : 0;  // row blocks of this rank
    const int ngroups = (nlb + 3) / 4;
    auto first_chunk = [&](int lb) { return (int)((((int64_t)rank + (int64_t)lb * world) * rbrows) / chunk); };  // chunk holding the block's first row
    std::vector<int2> sorted;  // (g, k)
    if (c->sym_order == 0) {
        // group major, longest rows first (group 0 sweeps the most columns)
        for (int g = 0; g < ngroups; ++g)
            for (int k = first_chunk(4 * g); k < nchunk; ++k) sorted.push_back(make_int2(g, k));
    } else {
        // XCD-aware order.  Workgroups go round-robin to the 8 XCDs (workgroup b -> XCD b % 8), each with its own 4-MB L2, and the
        // streamed side of a workgroup is its column chunk (chunk * (DP + 2) operands, 80 KB at 1024 columns).  Groups are sorted by
        // column chunk and the sorted list is cut into 8 contiguous ranges of equal length, one per XCD: an XCD then only ever
        // streams its own ~1/8 of the columns (L2-resident), instead of every XCD sweeping all of X through the Infinity Cache.
        for (int k = 0; k < nchunk; ++k)
            for (int g = 0; g < ngroups; ++g)
                if (first_chunk(4 * g) <= k) sorted.push_back(make_int2(g, k));
    }
    const size_t T = sorted.size();
    std::vector<int2> order;  // workgroup b -> (g, k) or (-1, -1)
    if (c->sym_order == 0) {
        order = sorted;
    } else {
        const int XCDS = 8;
        const size_t per_xcd = (T + XCDS - 1) / XCDS;  // workgroups per XCD
        order.assign(per_xcd * XCDS, make_int2(-1, -1));
        // inside an XCD's range (a dozen column chunks, ~1 MB of streamed operands that stay in its L2) go group by group,
        // so that the row operands of a group are fetched once per XCD rather than once per workgroup
        for (size_t x = 0; x < (size_t)XCDS; ++x) {
            const size_t lo = x * per_xcd < T ? x * per_xcd : T, hi = (x + 1) * per_xcd < T ? (x + 1) * per_xcd : T;
            std::stable_sort(sorted.begin() + lo, sorted.begin() + hi, [](const int2& a, const int2& b) { return a.x < b.x; });
            for (size_t q = lo; q < hi; ++q) order[(q - lo) * XCDS + x] = sorted[q];  // workgroup index that lands on XCD x
        }
    }
    std::vector<int2> items(order.size() * 4, make_int2(-1, -1));
    double pairs = 0.0;  // evaluated kernel pairs of one launch (cglb_get_stat "k1_pairs_per_launch")
    for (size_t b = 0; b < order.size(); ++b) {
        const int g = order[b].x, k = order[b].y;
        if (g < 0) continue;
        for (int w = 0; w < 4; ++w) {
            const int lb = 4 * g + w;
            items[4 * b + w] = make_int2(-1, k);
            if (lb >= nlb || first_chunk(lb) > k) continue;
            const int rb = rank + lb * world;
            items[4 * b + w] = make_int2(rb, k);
            const int64_t rbase = (int64_t)rb * rbrows;
            const int64_t rows = (n - rbase < rbrows) ? n - rbase : rbrows;
            int64_t j0 = (int64_t)k * chunk, j1 = ((int64_t)k + 1) * chunk;
            if (j0 < rbase) j0 = rbase;
            if (j1 > n) j1 = n;
            if (j1 > j0) pairs += (double)rows * (double)(j1 - j0);
        }
    }
    c->sym_pairs = pairs;
    if (c->sym_items) HIP_CHECK(c, hipFree(c->sym_items));
    c->sym_items = nullptr;
    if (order.empty()) { order.push_back(make_int2(-1, -1)); items.assign(4, make_int2(-1, -1)); }  // keep the allocation non-empty
    // one allocation: the items first, the groups behind them
    HIP_CHECK(c, hipMalloc(&c->sym_items, (items.size() + order.size()) * sizeof(int2)));
    HIP_CHECK(c, hipMemcpyAsync(c->sym_items, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(c, hipMemcpyAsync((int2*)c->sym_items + items.size(), order.data(), order.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->sym_n = n; c->sym_rbrows = rbrows; c->sym_chunk = chunk; c->sym_world = world; c->sym_rank = rank; c->sym_order_built = c->sym_order;
    c->sym_nitems = T > 0 ? (int)order.size() : 0;  // number of workgroups
    *nblocks_out = c->sym_nitems; *nrb_out = nrb; *nchunk_out = nchunk;
    return CGLB_OK;
}

// cyclic == false: the square block of the local row shard [r0, r1) plus plain kernels for the off-diagonal ranges.
// cyclic == true : the whole N x N upper triangle, row blocks rb == par_rank (mod par_world); output is a full-N PARTIAL vector
//                  (var * sums; rank 0 adds the noise term noise * p) that the caller all-reduces into (K_ff + noise I) p.
template <typename T, int KIND, int DP>
static int kff_sym_generic(cglb_ctx* c, const T* p_full, T* out_local, double* pdot_slot, bool cyclic) {
    // rows per lane: bounded by the VGPR budget (R * DP operands of sizeof(T)); fp32 operands are half the size
    constexpr int R = sizeof(T) == 8 ? ((DP <= 4) ? 8 : (DP <= CGLB_SYM_R4_MAX_DP) ? 4 : (DP <= 16 ? 2 : 1)) : ((DP <= 4) ? 8 : (DP <= 16) ? 4 : 2);
    constexpr int RBROWS = 64 * R;
    const int64_t n = cyclic ? c->N : c->nloc;
    const int64_t row0 = cyclic ? 0 : c->r0;
    const int world = cyclic ? c->par_world : 1, rank = cyclic ? c->par_rank : 0;
    // (measured per-rank kernel at N=100k: world 8: 0.48 ms at 128, 0.63 ms at 1024; world 4: 0.82 at 256, 0.86 at 512) -> halve the
    // 1024-column chunk until a rank has >= 16k items; large N keeps 1024 at any world size (fewer, larger slabs)
    int64_t chunk = 1024;
    {
        const double nrb_rank = (double)((n + RBROWS - 1) / RBROWS) / world;
        while (chunk > 128 && nrb_rank * ((double)n / (double)chunk) * 0.5 < 16384.0) chunk /= 2;
    }
    if (c->sym_chunk_opt > 0) chunk = c->sym_chunk_opt;
    chunk = (chunk + SYM_BATCH - 1) / SYM_BATCH * SYM_BATCH;
    if (chunk > SYM_CHUNK_MAX) chunk = SYM_CHUNK_MAX;  // the column sums of a chunk are staged in LDS
    int nblocks = 0, nrb = 0, nchunk = 0;
    CGLB_TRY(ensure_sym_items(c, n, RBROWS, chunk, world, rank, &nblocks, &nrb, &nchunk));
    const int ncslot = (nrb + world - 1) / world;      // row blocks of a rank (upper bound): row-sum slab rows
    const int ngslot = (ncslot + 3) / 4;               // groups of four of them: column-sum slabs
    // off-diagonal column ranges of the row shard ([0,r0) and [r1,N)) go through the plain kernel
    const int64_t nleft = cyclic ? 0 : c->r0, nright = cyclic ? 0 : c->N - c->r1;
    int64_t plain_slots_max = 0;
    if (nleft > 0 || nright > 0) plain_slots_max = 2 * 512;
    const int64_t prow_ld = (int64_t)ncslot * RBROWS;  // compact rows of this rank's row blocks
    const size_t need = (((size_t)plain_slots_max + ngslot) * n + (size_t)nchunk * prow_ld) * sizeof(T);
    if (need > c->kpart_cap) {
        if (c->kpart) HIP_CHECK(c, hipFree(c->kpart));
        c->kpart = nullptr;
        c->kpart_cap = 0;
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(c, hipMemGetInfo(&free_b, &total_b));
        if (need > free_b)  // the partial-sum slabs grow as N^2 / 256 elements per rank: say so instead of failing inside hipMalloc
            return cglb_fail(c, CGLB_ERR_HIP, "K_ff mat-vec needs " + std::to_string(need >> 20) + " MiB of partial-sum slabs (N^2/256 + N^2/chunk elements per rank) but only " +
                                                  std::to_string(free_b >> 20) + " MiB of device memory are free: shard the rows over more GPUs");
        HIP_CHECK(c, hipMalloc(&c->kpart, need));
        c->kpart_cap = need;
    }
    T* plain = (T*)c->kpart;
    int64_t nplain = 0;
    if (nleft > 0) {
        int64_t ns = 0;
        CGLB_TRY(launch_kff_plain_range(c, p_full, 0, c->r0, plain, &ns));
        nplain += ns;
    }
    if (nright > 0) {
        int64_t ns = 0;
        CGLB_TRY(launch_kff_plain_range(c, p_full, c->r1, c->N, plain + nplain * n, &ns));
        nplain += ns;
    }
    T* Prow = plain + nplain * n;
    T* Pcol = Prow + (int64_t)nchunk * prow_ld;
    const int grid = nblocks;
    const int2* items_dev = (const int2*)c->sym_items;
    const int2* groups_dev = items_dev + (size_t)4 * (nblocks > 0 ? nblocks : 1);
    if (grid > 0) {
        if (!c->exp_clamp && KIND == CGLB_RBF) {  // folded column norm: pre-weight the operand over the columns of this block
            // (skipped when the update_p kernel that produced exactly this vector has already written the weighted copy)
            const bool have = c->pwh_src == (const void*)p_full && row0 == 0 && n == c->N;
            if (!have) {
                hipLaunchKernelGGL((weight_operand_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, p_full + row0,
                                   (const T*)c->wh + row0, n, (T*)c->pwh + row0);
                CGLB_LAUNCH_CHECK(c);
            }
        }
        c->pwh_src = nullptr;  // single use: the vector may be modified by the caller afterwards
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->k1_profile) {  // in-situ timing of the dominant kernel (cglb_get_stat)
            if (c->k1_events_used + 2 > c->k1_events.size()) {
                if (c->k1_events.size() >= 4096) CGLB_TRY(k1_profile_collect(c));  // bounded pool: resolve (synchronises) and reuse
                else
                    for (int q = 0; q < 2; ++q) {
                        hipEvent_t ev;
                        HIP_CHECK(c, hipEventCreate(&ev));
                        c->k1_events.push_back(ev);
                    }
            }
            e0 = c->k1_events[c->k1_events_used];
            e1 = c->k1_events[c->k1_events_used + 1];
            c->k1_events_used += 2;
            HIP_CHECK(c, hipEventRecord(e0, c->stream));
        }
        CGLB_DISPATCH_PREC(c, {
            if (c->exp_clamp)
                hipLaunchKernelGGL((kff_sym_kernel<T, KIND, DP, R, true, PREC>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->Xh, (const T*)c->xah,
                                   p_full, (const T*)nullptr, (const T*)nullptr, row0, n, chunk, items_dev, groups_dev, world, prow_ld, Prow,
                                   Pcol, (const double*)c->exp_tab, (T)0);
            else
                hipLaunchKernelGGL((kff_sym_kernel<T, KIND, DP, R, false, PREC>), dim3(grid), dim3(256), 0, c->stream, (const T*)c->Xh, (const T*)c->xah,
                                   p_full, (const T*)c->pwh, (const T*)c->wh, row0, n, chunk, items_dev, groups_dev, world, prow_ld, Prow,
                                   Pcol, (const double*)c->exp_tab, (T)c->m32_bias);
        });
        CGLB_LAUNCH_CHECK(c);
        if (e1) HIP_CHECK(c, hipEventRecord(e1, c->stream));
    }
    if (c->kff_skip_combine) return CGLB_OK;
    const int cgrid = (int)((n + 63) / 64);  // 64 elements per block
    if (pdot_slot && cgrid > DOTPART_CAP) return cglb_fail(c, CGLB_ERR_BAD_ARG, "row shard too large for dot partials");
    hipLaunchKernelGGL((kff_sym_combine_kernel<T>), dim3(cgrid), dim3(256), 0, c->stream, (const T*)plain, (int)nplain, (const T*)Prow, nchunk,
                       prow_ld, (const T*)Pcol, n, chunk, RBROWS, world, rank, (T)c->var, (T)c->noise, cyclic ? (rank == 0 ? p_full : (const T*)nullptr) : p_full + c->r0, out_local,
                       pdot_slot ? c->dotpart : nullptr);
    CGLB_LAUNCH_CHECK(c);
    if (pdot_slot) {
        hipLaunchKernelGGL(finalize_sum_sym_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->dotpart, cgrid, pdot_slot);
        CGLB_LAUNCH_CHECK(c);
    }
    return CGLB_OK;
}

#ifndef CGLB_SYM_MID_TU
int k1_profile_collect(cglb_ctx* c) {
    for (size_t q = 0; q + 1 < c->k1_events_used; q += 2) {
        HIP_CHECK(c, hipEventSynchronize(c->k1_events[q + 1]));
        float ms = 0.f;
        HIP_CHECK(c, hipEventElapsedTime(&ms, c->k1_events[q], c->k1_events[q + 1]));
        c->k1_ms_total += ms;
        c->k1_launches += 1;
    }
    c->k1_events_used = 0;
    return CGLB_OK;
}

#endif  // !CGLB_SYM_MID_TU

#ifdef CGLB_SYM_MID_TU
// mid-width contexts (cglb_internal.h: mid_dim): the same kernel on the Dh-wide hot operand set, Gram chain in slices.  These instances
// are compiled in their own translation unit (kernels_kff_sym_mid.hip includes this file with CGLB_SYM_MID_TU defined), in parallel
// with the narrow ones.
#define CGLB_DISPATCH_DH(dh, ...)                                           \
    switch (dh) {                                                           \
        case 48: { constexpr int DP = 48; __VA_ARGS__; } break;             \
        case 64: { constexpr int DP = 64; __VA_ARGS__; } break;             \
        case 80: { constexpr int DP = 80; __VA_ARGS__; } break;             \
        case 96: { constexpr int DP = 96; __VA_ARGS__; } break;             \
        default: return cglb_fail(c, CGLB_ERR_BAD_ARG, "unsupported mid width"); \
    }

int launch_kff_sym_mid(cglb_ctx* c, const void* p_full, void* out, double* pdot_slot, bool cyclic) {
    using T = double;
    if (!cyclic && c->nloc != c->N) return cglb_fail(c, CGLB_ERR_STATE, "the register-resident mid-width mat-vec covers the full square or the cyclic deal only");
    CGLB_DISPATCH_KIND(c->kind, CGLB_DISPATCH_DH(c->Dh, return (kff_sym_generic<T, KIND, DP>(c, (const T*)p_full, (T*)out, pdot_slot, cyclic))));
    return CGLB_OK;
}

#else   // the narrow translation unit
int launch_kff_sym(cglb_ctx* c, const void* p_full, void* out_local, double* pdot_slot) {
    if (is_wide(c)) return launch_kff_matvec(c, p_full, out_local, pdot_slot);
    CGLB_DISPATCH_ALL(c, return (kff_sym_generic<T, KIND, DP>(c, (const T*)p_full, (T*)out_local, pdot_slot, false)));
    return CGLB_OK;
}

int launch_kff_sym_cyclic(cglb_ctx* c, const void* p_full, void* out_full_partial) {
    // wide inputs: row tiles of the full square dealt round-robin to the ranks; every rank adds the noise term on its own rows
    if (is_wide(c) && mid_reg(c)) return launch_kff_sym_mid(c, p_full, out_full_partial, nullptr, true);
    if (is_wide(c)) return wide_matvec(c, c->Xs, c->xa, 0, c->N, p_full, out_full_partial, true, nullptr, c->par_world, c->par_rank);
    CGLB_DISPATCH_ALL(c, return (kff_sym_generic<T, KIND, DP>(c, (const T*)p_full, (T*)out_full_partial, nullptr, true)));
    return CGLB_OK;
}
#endif  // CGLB_SYM_MID_TU
