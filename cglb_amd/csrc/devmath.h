// Device-side arithmetic helpers for the streaming kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cglb_hip.h"

#define CGLB_LOG2E 1.4426950408889634073599246810019
#define CGLB_LN2 0.69314718055994530941723212145818
#define CGLB_SQRT3 1.7320508075688772935274463415059

// 2^x for x <= 0 (x may be hugely negative -> 0).  fp64: no hardware transcendental on CDNA, so
// round-to-nearest split x = n + r, |r| <= 1/2, degree-11 polynomial for 2^r (max rel. error 1.8e-16
// including Horner round-off; fitted on Chebyshev nodes in extended precision), then ldexp.
// 15 vector-fp64 instructions: rndne, add, 11 fma, cvt, ldexp.
__device__ __forceinline__ double exp2_neg(double x) {
    x = fmax(x, -1100.0);  // keeps the int conversion in range; 2^-1100 underflows to 0 anyway
    const double n = __builtin_rint(x);
    const double r = x - n;
    double p = 0x1.e9ec94f24bf5bp-32;
    p = __builtin_fma(p, r, 0x1.e6228f265d4ebp-28);
    p = __builtin_fma(p, r, 0x1.b524ead100ee2p-24);
    p = __builtin_fma(p, r, 0x1.62bfc2c4b4a97p-20);
    p = __builtin_fma(p, r, 0x1.ffcbfc6e966e3p-17);
    p = __builtin_fma(p, r, 0x1.430913112ed6ap-13);
    p = __builtin_fma(p, r, 0x1.5d87fe78a3a63p-10);
    p = __builtin_fma(p, r, 0x1.3b2ab6fb9f18fp-7);
    p = __builtin_fma(p, r, 0x1.c6b08d704a0c9p-5);
    p = __builtin_fma(p, r, 0x1.ebfbdff82c5afp-3);
    p = __builtin_fma(p, r, 0x1.62e42fefa39efp-1);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }

// Hot-loop variant: degree-10 polynomial (max rel. error 4.3e-16, 2 ulp) and no range clamp.
// Valid for |x| < 2^30 (v_cvt_i32_f64 stays in range; ldexp flushes to 0 far below -1075); the host
// checks that bound on the scaled operands (cglb_set_hypers) and otherwise selects the clamped form.
// 14 vector-fp64 instructions: rndne, add, 10 fma, cvt, ldexp.
template <bool CLAMP> __device__ __forceinline__ double exp2_hot(double x) {
    if (CLAMP) x = fmax(x, -1100.0);
    const double n = __builtin_rint(x);
    const double r = x - n;
    double p = 0x1.e6063d5fed313p-28;
    p = __builtin_fma(p, r, 0x1.b675bd9d9ead9p-24);
    p = __builtin_fma(p, r, 0x1.62bfd477ed5d2p-20);
    p = __builtin_fma(p, r, 0x1.ffcb54050949cp-17);
    p = __builtin_fma(p, r, 0x1.430913096f8e3p-13);
    p = __builtin_fma(p, r, 0x1.5d87fe9d7acc1p-10);
    p = __builtin_fma(p, r, 0x1.3b2ab6fba1de2p-7);
    p = __builtin_fma(p, r, 0x1.c6b08d703ce44p-5);
    p = __builtin_fma(p, r, 0x1.ebfbdff82c598p-3);
    p = __builtin_fma(p, r, 0x1.62e42fefa3a19p-1);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
template <bool CLAMP> __device__ __forceinline__ float exp2_hot(float x) { return __builtin_amdgcn_exp2f(x); }

// sqrt for x >= 0: hardware rsq seed (relative error e0 <= ~2^-23 on fp64), one Goldschmidt step on g (-> ~e0^2) and one
// residual correction with the first-order h (-> ~e0^3, below the rounding of the last fma).  x == 0 returns 0.  7 instructions
// + v_rsq_f64, which alone occupies 4 issue slots.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double xs = fmax(x, 1e-280);
    const double y = __builtin_amdgcn_rsq(xs);
    double g = xs * y;
    double h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    const double d = __builtin_fma(-g, g, xs);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}
__device__ __forceinline__ float sqrt_pos(float x) { return __builtin_sqrtf(fmaxf(x, 0.0f)); }

// ---- precision levels of the pair kernels (cglb_set_option "precision") ------------------------------------------------------
//   CGLB_PREC_EXACT (0): degree-4 table polynomial + two-step square root: kernel values to ~3e-16 (1.5 ulp)
//   CGLB_PREC_FAST  (1): degree-3 polynomial (3.5e-14) + one-step square root (2.1e-14 on r): kernel values to <= ~1e-13 relative.
// north_star asks for 1e-6 on the bound; FAST is the default (one fma per pair less, two for Matern-3/2).  Every fp64 instruction
// of the pair stream is ~5 % of the mat-vec (the kernels are bound by vector-fp64 issue).
#define CGLB_PREC_EXACT 0
#define CGLB_PREC_FAST 1
//   CGLB_PREC_LOW   (2): degree-2 polynomial (1.0e-10): kernel values to ~1e-10 - inside north_star's 1e-6 on the bound, outside the
//                        1e-10 the parity tests hold the default level to: opt-in only (one instruction per pair less than FAST).
#define CGLB_PREC_LOW 2

// Hot-loop square root for a squared distance that may come out slightly negative (Gram form): the clamp to a tiny positive
// number replaces both the max(.,0) and the x > 0 select; sqrt_hot(d2 <= 0) = 2e-140, which the Matern profile maps to 1.
// RETURNS 2 sqrt(x): the hot operand set of Matern-3/2 carries half the nominal scale (cglb_hot_scale), so that the Newton form
//   y = rsq(x) (v_rsq_f64, relative error e0 <= 2^-23, 4 issue slots),  g = x y,  t = 3 - y g,  2 sqrt(x) = g t (1 - 1.5 e0^2)
// needs three instructions after the seed where sqrt(x) itself needs four (the factor 1/2 of y' = y (3 - x y^2) / 2 is the one
// that costs an instruction; a power-of-two scale of the operands is exact).  Error 1.5 e0^2 = 2.1e-14.  At CGLB_PREC_EXACT:
// one Goldschmidt step plus a residual correction with the first-order h (error ~ e0^3, below the rounding of the last fma), doubled.
// POSITIVE: the caller guarantees x > 0 (Matern-3/2 at the fast level: the row seeds of the Gram chain carry a bias that exceeds its
// worst cancellation error, cglb_set_hypers / CGLB_M32_BIAS_*), so the clamp - one instruction per pair - is dropped.
template <int PREC, bool POSITIVE = false> __device__ __forceinline__ double sqrt_hot(double x) {
    const double xs = POSITIVE ? x : fmax(x, 1e-280);
    const double y = __builtin_amdgcn_rsq(xs);
    double g = xs * y;
    if (PREC != CGLB_PREC_EXACT) return g * __builtin_fma(-y, g, 3.0);
    double h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    const double d = __builtin_fma(-g, g, xs);  // residual ~ e0^2 xs: the first-order h (error e0) is accurate enough to apply it
    g = __builtin_fma(d, h, g);
    return g + g;
}
template <int PREC, bool POSITIVE = false> __device__ __forceinline__ float sqrt_hot(float x) { return 2.0f * __builtin_sqrtf(fmaxf(x, 0.0f)); }
// Bias of the Matern-3/2 squared distance in hot units^2 (added through the row seed -(a_i + bias)/2): d2 = a_i + a_j - 2 x_i.x_j.
// With u = 2^-53 and amax = max_i a_i: the seed and the D fused multiply-adds of the Gram chain each round a partial sum of size
// <= 1.5 amax (error of the chain <= (D + 1) 1.5 u amax, doubled by d2 = a_j - 2 gram), and the stored norms a_i, a_j differ from the
// exact |x|^2 of the stored operands by <= D u amax each: |d2_computed - d2_exact| <= (5 D + 3) u amax.  The bias is 1.5x that bound.
// A coincident pair then evaluates to kappa = 1 - 2 (ln2/256)^2 bias instead of 1; the bias is only used while that stays below 3e-13
// (bias <= 2e-8; the headline shape has 1.2e-8), otherwise the clamped variant runs.  Distinct points sit at d2 >~ 1 hot unit^2 and
// do not see it.
#define CGLB_M32_BIAS_FACTOR (1.5 * 1.1102230246251565e-16)   // bias = FACTOR * (5 D + 3) * amax
#define CGLB_M32_BIAS_MAX 2.0e-8

__device__ __forceinline__ double tfma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tfma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Table-driven 2^(xh/T), T = 256: the hot pair kernels keep their operands in units of 1/T octave (xh = T * log2 of the value):
//   xh = n + s, n = floor(xh), 0 <= s < 1;   2^(xh/T) = 2^(n >> 8) * TAB[n & 255] * P(s),
//   TAB[k] = 2^((k + 1/2)/T) (correctly rounded, in LDS), P(s) = 2^((s - 1/2)/T) a polynomial (tools/exp2_poly_fit.py).
// fp64 has no hardware transcendental on CDNA.  Instruction count of one 2^x (every one a 4-cycle issue slot of the SIMD):
//   range reduction 2 (round-down add of 1.5 * 2^52, v_fract_f64)  [was 3: rndne, sub, cvt]
//   table address 1 (SDWA byte extract + shift), ds_read_b64 (LDS pipe, not VALU)
//   polynomial 3 fma at CGLB_PREC_FAST / 4 at CGLB_PREC_EXACT
//   octave scaling 1 (v_lshl_add_u32 into the exponent field), product 1          -> 8 (FAST) / 9 (EXACT), against 14 for a
//   pure polynomial 2^x.  Table size history (round 1, N = 100k D = 8): 64 entries + degree 5 3.69 ms, 256 + degree 4 3.47 ms,
//   4096 + degree 3 3.45 ms (the byte-select addressing only exists for 8 bits: a 12-bit index costs the instruction the shorter
//   polynomial saves).
#define CGLB_TAB_BITS 8
#define CGLB_TAB_SIZE (1 << CGLB_TAB_BITS)
#define CGLB_HOT_UNITS ((double)CGLB_TAB_SIZE)
// P(s) = 2^((s - 1/2) / 256) on 0 <= s < 1; maximum relative error: degree 4 1.7e-17 (+ Horner round-off ~1e-16), degree 3 1.8e-14
template <int PREC> __device__ __forceinline__ double exp2_tab_poly(double s) {
    if (PREC == CGLB_PREC_EXACT) {
        double p = 0x1.3b2ab88f70400p-39;
        p = __builtin_fma(p, s, 0x1.c612fb7dd6528p-29);
        p = __builtin_fma(p, s, 0x1.eb517b4ddbd98p-19);
        p = __builtin_fma(p, s, 0x1.6269464576054p-9);
        return __builtin_fma(p, s, 0x1.ff4eaca4391b6p-1);
    }
    if (PREC == CGLB_PREC_LOW) {  // degree 2: max. rel. error 1.03e-10 (tools/exp2_poly_fit.py)
        double p = 0x1.ebfbe3a9ac80bp-19;
        p = __builtin_fma(p, s, 0x1.6269364acae25p-9);
        return __builtin_fma(p, s, 0x1.ff4eaca51c5ffp-1);
    }
    double p = 0x1.c6b0902b5a0abp-29;
    p = __builtin_fma(p, s, 0x1.eb5162aec6f78p-19);
    p = __builtin_fma(p, s, 0x1.62694646b129dp-9);
    return __builtin_fma(p, s, 0x1.ff4eaca439118p-1);
}
// Range reduction by floor/fract: n = floor(x) is obtained WITHOUT a conversion: t = x + 1.5 * 2^52 added under
// round-toward-minus-infinity leaves floor(x) (two's complement) in the low mantissa word of t; s = v_fract_f64(x) is exact, and
// n and s are consistent by construction (both floor-based, no tie cases).  The fp64 rounding mode (MODE.FP_ROUND[3:2]) is switched
// for exactly these adds inside one asm block, so no other arithmetic sees it (the mode is per wave).  A lane's R values share
// one window: 2 SALU instructions per R adds.
#define CGLB_MAGIC_FLOOR 6755399441055744.0  // 1.5 * 2^52: low 32 mantissa bits zero, ulp 1
template <bool NEG> __device__ __forceinline__ void floor_magic4(double x0, double x1, double x2, double x3, double& t0, double& t1, double& t2,
                                                                 double& t3) {
    const double magic = CGLB_MAGIC_FLOOR;
    if (NEG)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, -%4, %8\n\tv_add_f64 %1, -%5, %8\n\tv_add_f64 %2, -%6, %8\n\tv_add_f64 %3, -%7, %8\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(magic));
    else
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, %4, %8\n\tv_add_f64 %1, %5, %8\n\tv_add_f64 %2, %6, %8\n\tv_add_f64 %3, %7, %8\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(magic));
}
template <bool NEG> __device__ __forceinline__ void floor_magic2(double x0, double x1, double& t0, double& t1) {
    const double magic = CGLB_MAGIC_FLOOR;
    if (NEG)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, -%2, %4\n\tv_add_f64 %1, -%3, %4\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0), "=&v"(t1)
                     : "v"(x0), "v"(x1), "s"(magic));
    else
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, %2, %4\n\tv_add_f64 %1, %3, %4\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0), "=&v"(t1)
                     : "v"(x0), "v"(x1), "s"(magic));
}
template <bool NEG> __device__ __forceinline__ void floor_magic1(double x0, double& t0) {
    const double magic = CGLB_MAGIC_FLOOR;
    if (NEG)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, -%1, %2\n\ts_nop 0\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0)
                     : "v"(x0), "s"(magic));
    else
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
                     "v_add_f64 %0, %1, %2\n\ts_nop 0\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
                     : "=&v"(t0)
                     : "v"(x0), "s"(magic));
}
// floor of R values (NEG: of their negatives) into the low word of t[r]
template <bool NEG, int R> __device__ __forceinline__ void floor_magic(const double (&x)[R], double (&t)[R]) {
    if constexpr (R % 4 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 4) floor_magic4<NEG>(x[r], x[r + 1], x[r + 2], x[r + 3], t[r], t[r + 1], t[r + 2], t[r + 3]);
    } else if constexpr (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) floor_magic2<NEG>(x[r], x[r + 1], t[r], t[r + 1]);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) floor_magic1<NEG>(x[r], t[r]);
    }
}
// The table holds 2^((k + 1/2)/T) with (k << (20 - BITS)) subtracted from the high word, so that ONE integer instruction
//   hi(entry[n & (T-1)]) + (n << (20 - BITS))  =  hi(2^((k + 1/2)/T)) + ((n >> BITS) << 20)
// puts the octave count straight into the exponent field (v_lshl_add_u32 instead of v_ashrrev + v_ldexp_f64).  Unlike ldexp
// this cannot underflow gracefully: callers keep n / T inside [-1000, 1000] (range check in set_hypers or the CLAMP variant,
// which then returns 2^-1000 ~ 1e-301 where ldexp would have returned 0).
#define CGLB_EXP_FLOOR_OCT 1000.0
__device__ __forceinline__ double exp2_tab_scale(double entry, int ni) {
    const int hi = __double2hiint(entry) + (int)((unsigned)ni << (20 - CGLB_TAB_BITS));
    return __hiloint2double(hi, __double2loint(entry));
}
// R values at once: out[r] = 2^(x[r]/T) (NEG: 2^(-x[r]/T))
template <bool CLAMP, bool NEG, int PREC, int R>
__device__ __forceinline__ void exp2_tab_batch(const double (&xin)[R], const double* __restrict__ tab_lds, double (&out)[R]) {
    double x[R], t[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        x[r] = CLAMP ? (NEG ? fmin(xin[r], CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE) : fmax(xin[r], -CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE)) : xin[r];
    floor_magic<NEG, R>(x, t);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double s = __builtin_amdgcn_fract(NEG ? -x[r] : x[r]);
        const int ni = __double2loint(t[r]);
        out[r] = exp2_tab_scale(tab_lds[ni & (CGLB_TAB_SIZE - 1)], ni) * exp2_tab_poly<PREC>(s);
    }
}
template <bool CLAMP, bool NEG, int PREC, int R>
__device__ __forceinline__ void exp2_tab_batch(const float (&xin)[R], const double* __restrict__, float (&out)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = __builtin_amdgcn_exp2f((NEG ? -xin[r] : xin[r]) * (1.0f / (float)CGLB_TAB_SIZE));  // hardware exp2
}
template <bool CLAMP, int PREC> __device__ __forceinline__ double exp2_tab(double xh, const double* __restrict__ tab_lds) {
    const double x1[1] = {xh};
    double o1[1];
    exp2_tab_batch<CLAMP, false, PREC, 1>(x1, tab_lds, o1);
    return o1[0];
}
template <bool CLAMP, int PREC> __device__ __forceinline__ float exp2_tab(float xh, const double* __restrict__) {
    return __builtin_amdgcn_exp2f(xh * (1.0f / (float)CGLB_TAB_SIZE));
}

// acc += (lane 16 (l / 16) + K of bc) * x: v_fmac_f64 with the DPP control row_newbcast (gfx90a+).  One VGPR pair whose lane l holds operand
// l % 16 hands 16 different wave-uniform operands to 16 consecutive fmas at the full fma rate (tools/microbench/dpp_fmac.hip:
// 4.9-5.2 nominal cycles per instruction for both forms) - the operand path of the mid-width kernels (32 < D <= 96: kernels_kff_sym.hip, kernels_grad.hip), whose
// columns no longer fit the scalar register file.
template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, double bc, double x) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(x), "n"(K));
}
template <int K0, int N> struct BcastChain {   // acc += sum_{k < N} bc[lane k] * x[k]
    static __device__ __forceinline__ void run(double& acc, double bc, const double* x) {
        fmac_bcast<K0>(acc, bc, x[K0]);
        if constexpr (K0 + 1 < N) BcastChain<K0 + 1, N>::run(acc, bc, x);
    }
};
template <int K0, int N> struct BcastChain2 {  // two interleaved chains (even / odd k): for the instances that run one wave per SIMD
    static __device__ __forceinline__ void run(double& a0, double& a1, double bc, const double* x) {
        fmac_bcast<K0>(a0, bc, x[K0]);
        fmac_bcast<K0 + 1>(a1, bc, x[K0 + 1]);
        if constexpr (K0 + 2 < N) BcastChain2<K0 + 2, N>::run(a0, a1, bc, x);
    }
};
template <int K0, int N> struct BcastAxpy {    // acc[k] += bc[lane k] * x for k < N (the moment accumulation of the gradient pass)
    static __device__ __forceinline__ void run(double* acc, double bc, double x) {
        fmac_bcast<K0>(acc[K0], bc, x);
        if constexpr (K0 + 1 < N) BcastAxpy<K0 + 1, N>::run(acc, bc, x);
    }
};
// sum of a value over the two 32-lane halves of the wave, returned to both (v_permlane32_swap, gfx950): with old == src the first result is
// the lower half's value in every lane, the second the upper half's
__device__ __forceinline__ double sum_halves(double g) {
    const unsigned lo = (unsigned)__double2loint(g), hi = (unsigned)__double2hiint(g);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// sum over the 16-lane rows 2 q and 2 q + 1, returned to both (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
__device__ __forceinline__ double sum_row_pairs(double g) {
    const unsigned lo = (unsigned)__double2loint(g), hi = (unsigned)__double2hiint(g);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }

// Kernel profile from the scaled operands.
//   RBF:      xs = (x-c)/l*sqrt(log2 e), a = -|xs|^2/2      kappa = 2^(a_i + a_j + xs_i.xs_j)
//   Matern32: xs = (x-c)/l*sqrt3*log2 e, a = |xs|^2          r' = sqrt(max(a_i+a_j-2 xs_i.xs_j,0)),
//             kappa = (1 + r' ln2) 2^(-r')
// `gram` = a_i + xs_i.xs_j (the fma chain is seeded with a_i).  RBF: arg = gram + a_j may come out a few ulp
// above 0 for coincident points; 2^arg is then 1 + O(1e-16), harmless, so no clamp is spent on it.
template <typename T, int KIND, bool CLAMP> __device__ __forceinline__ T kappa_from_gram(T gram, T aj) {
    if (KIND == CGLB_RBF) {
        return exp2_hot<CLAMP>(gram + aj);
    } else {
        // Matern: the chain is seeded with -a_i/2, so gram = -a_i/2 + xs_i.xs_j and d2 = a_j - 2 gram
        T d2 = tfma<T>(T(-2), gram, aj);
        T r = sqrt_pos(tmax<T>(d2, T(0)));
        return tfma<T>(r, T(CGLB_LN2), T(1)) * exp2_hot<CLAMP>(-r);
    }
}

// Hot-unit forms (operands scaled so that exponents are in 1/T octave, see exp2_tab_batch):
//   RBF:      xh = 16 xs, ah = 256 a          kappa = 2^((ah_i + ah_j + xh_i.xh_j)/256)
//   Matern32: xh = 128 xs, ah = 16384 a       rT = 2 sqrt(max(ah_i + ah_j - 2 xh_i.xh_j, 0)),  kappa = (1 + rT ln2/256) 2^(-rT/256)
//             (half the nominal scale: sqrt_hot returns twice the root)
// Two-phase evaluation for software-pipelined loops: `begin` does the range reductions of the R rows a lane owns and issues the R
// table reads, `poly` is independent of the reads, `end` consumes them - so a loop can keep R lookups in flight.
template <typename T> struct KappaPend { T s; T lin; int ni; T tabv; };
// FOLDED, RBF: the column norm a_j is not added here - the caller has folded 2^(a_j/T) into the column operand.
// FOLDED, Matern-3/2: the caller's row seeds carry the positivity bias (sqrt_hot<PREC, true>: no clamp).
template <typename T, int KIND, bool CLAMP, bool FOLDED, int PREC, int R>
__device__ __forceinline__ void kappa_hot_begin_batch(const T (&gram)[R], T aj, const double* __restrict__ tab, KappaPend<T> (&kp)[R]) {
    if constexpr (sizeof(T) == 8) {
        double x[R], t[R];  // RBF: x = exponent; Matern: x = r (the exponent is -r)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (KIND == CGLB_RBF) {
                x[r] = FOLDED ? gram[r] : gram[r] + aj;
                kp[r].lin = T(1);
                if (CLAMP) x[r] = tmax<double>(x[r], -CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE);
            } else {
                const double d2 = tfma<double>(-2.0, gram[r], aj);
                x[r] = sqrt_hot<PREC, FOLDED>(d2);
                kp[r].lin = tfma<double>(x[r], CGLB_LN2 / CGLB_HOT_UNITS, 1.0);
                if (CLAMP) x[r] = tmin<double>(x[r], CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE);
            }
        }
        constexpr bool NEG = KIND != CGLB_RBF;
        floor_magic<NEG, R>(x, t);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            kp[r].s = __builtin_amdgcn_fract(NEG ? -x[r] : x[r]);
            kp[r].ni = __double2loint(t[r]);
            kp[r].tabv = tab[kp[r].ni & (CGLB_TAB_SIZE - 1)];
        }
    } else {  // fp32 has a hardware exp2: no table
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (KIND == CGLB_RBF) {
                kp[r].s = FOLDED ? gram[r] : gram[r] + aj;
                kp[r].lin = T(1);
            } else {
                const T rr = sqrt_hot<PREC>(tfma<T>(T(-2), gram[r], aj));
                kp[r].lin = tfma<T>(rr, T(CGLB_LN2 / CGLB_HOT_UNITS), T(1));
                kp[r].s = -rr;
            }
            kp[r].ni = 0;
            kp[r].tabv = T(0);
        }
    }
}
// polynomial part (independent of the table read): overwrites k.s with P(s) [* lin for Matern]
template <typename T, int KIND, int PREC> __device__ __forceinline__ void kappa_hot_poly(KappaPend<T>& k) {
    T p;
    if constexpr (sizeof(T) == 4) p = (T)__builtin_amdgcn_exp2f((float)k.s * (1.0f / (float)CGLB_TAB_SIZE));
    else p = exp2_tab_poly<PREC>(k.s);
    if (KIND != CGLB_RBF) p *= k.lin;
    k.s = p;
}
template <typename T, int KIND> __device__ __forceinline__ T kappa_hot_end(const KappaPend<T>& k) {
    if (sizeof(T) == 4) return k.s;
    return (T)(exp2_tab_scale((double)k.tabv, k.ni) * (double)k.s);
}
// one pair, same arithmetic as the batched form (ragged tails of the symmetric kernel)
template <typename T, int KIND, bool CLAMP, bool FOLDED, int PREC>
__device__ __forceinline__ T kappa_hot_single(T gram, T aj, const double* __restrict__ tab) {
    T g1[1] = {gram};
    KappaPend<T> kp[1];
    kappa_hot_begin_batch<T, KIND, CLAMP, FOLDED, PREC, 1>(g1, aj, tab, kp);
    kappa_hot_poly<T, KIND, PREC>(kp[0]);
    return kappa_hot_end<T, KIND>(kp[0]);
}

// gradient factor from an exact squared distance in hot units (RBF: d2h = 256 d2s; Matern32: d2h = 16384 d2s, r = 2 sqrt(d2h))
template <typename T, int KIND, bool CLAMP, int PREC> __device__ __forceinline__ T hfac_hot_from_d2(T d2h, const double* __restrict__ tab) {
    if (KIND == CGLB_RBF) {
        return exp2_tab<CLAMP, PREC>(T(-0.5) * d2h, tab);
    } else {
        return T(3) * exp2_tab<CLAMP, PREC>(T(-2) * sqrt_pos(d2h), tab);
    }
}
// cooperative load of the exp2 table into LDS (call from every thread of the block, before any early exit)
__device__ __forceinline__ void load_exp_table(double* tab_lds, const double* __restrict__ tab_global) {
    for (int i = threadIdx.x; i < CGLB_TAB_SIZE; i += blockDim.x) tab_lds[i] = tab_global[i];
    __syncthreads();
}

// Kernel profile from an exact scaled squared distance (direct differences; used off the N^2 path).
//   d2s is in the scaled units above.
template <typename T, int KIND> __device__ __forceinline__ T kappa_from_d2(T d2s) {
    if (KIND == CGLB_RBF) {
        return exp2_neg(T(-0.5) * d2s);
    } else {
        T r = sqrt_pos(d2s);
        return tfma<T>(r, T(CGLB_LN2), T(1)) * exp2_neg(-r);
    }
}
// gradient factor: dk/dl_d = var * hfac * delta_d^2 / l_d with delta in UNscaled-by-kscale units.
template <typename T, int KIND> __device__ __forceinline__ T hfac_from_d2(T d2s) {
    if (KIND == CGLB_RBF) {
        return exp2_neg(T(-0.5) * d2s);
    } else {
        return T(3) * exp2_neg(-sqrt_pos(d2s));
    }
}

// Streaming reads of operands that are touched once per pass and exceed every cache (the 819-MB Nystrom panel and its adjoint, the
// partial-sum slabs): non-temporal loads.  Measured on the preconditioner apply (two passes over the panel): 0.304 -> 0.287 ms on
// the same box (5.4 -> 5.7 TB/s).  A/B builds: EXTRA_DEFS=-DCGLB_STREAM_NT=0.
#ifndef CGLB_STREAM_NT
#define CGLB_STREAM_NT 1
#endif
#if CGLB_STREAM_NT
#define CGLB_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define CGLB_STREAM_LOAD(p) (*(p))
#endif

// ---- reductions ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block sum of a double for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* smem /* >= 16 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) s += smem[i];
    }
    return s;
}

// Every thread of the block gets the (bitwise identical, fixed-order) sum of n device doubles.
__device__ __forceinline__ double block_reduce_array(const double* __restrict__ a, int n, double* smem) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[i];
    s = block_sum(s, smem);
    __shared__ double bcast;
    if (threadIdx.x == 0) bcast = s;
    __syncthreads();
    return bcast;
}
