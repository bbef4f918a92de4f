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

// Hot-loop variant for a squared distance that may come out slightly negative (Gram form): the clamp to a tiny positive
// number replaces both the max(.,0) and the x > 0 select; sqrt_hot(d2 <= 0) = 1e-140, which the Matern profile maps to 1.
__device__ __forceinline__ double sqrt_hot(double x) {
    const double xs = fmax(x, 1e-280);
    const double y = __builtin_amdgcn_rsq(xs);
    double g = xs * y;
    double h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    const double d = __builtin_fma(-g, g, xs);  // residual ~ e0^2 xs: the first-order h (error e0) is accurate enough to apply it
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ float sqrt_hot(float x) { return __builtin_sqrtf(fmaxf(x, 0.0f)); }

__device__ __forceinline__ double tfma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tfma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Table-driven 2^(xh/T), T = 2^CGLB_TAB_BITS: the hot pair kernels keep their operands in units of 1/T octave (xh = T*log2 of the
// value), so  xh = n + s, |s| <= 1/2;  2^(xh/T) = 2^(n >> BITS) * TAB[n & (T-1)] * P(s),  TAB[k] = 2^(k/T) (correctly rounded, in LDS),
//   P = polynomial for 2^(s/T) (1.1e-16).  Total error <= ~3e-16 (1.5 ulp).
// fp64 has no hardware transcendental on CDNA and the pair kernels are bound by vector-fp64 issue, so every instruction here is
// ~4 % of the mat-vec: with T = 256 this is 9 vector-fp64 instructions (rndne, add, 4 fma, cvt, mul, ldexp) + 3 cheap 32-bit
// integer ops + one ds_read_b64, against 14 for the pure polynomial form.
// Table size is a build-time choice: 6 -> 64 entries (512 B) + degree 5, 8 -> 256 entries (2 KB) + degree 4 (default),
// 12 -> 4096 entries (32 KB) + degree 3.  Each step trades one fma per pair for LDS footprint and bank conflicts; measured
// mat-vec at N = 100k, D = 8 (MI355X): 3.69 / 3.47 / 3.45 ms - beyond 256 entries the random-index ds_read_b64 (one per pair,
// four SIMDs sharing the CU's LDS port) eats the fma that the shorter polynomial saves.
#ifndef CGLB_TAB_BITS
#define CGLB_TAB_BITS 8
#endif
#define CGLB_TAB_SIZE (1 << CGLB_TAB_BITS)
#define CGLB_HOT_UNITS ((double)CGLB_TAB_SIZE)
// P(s) = 2^(s / TAB_SIZE) on |s| <= 1/2 (max rel. error 1.1e-16 for each variant, fitted on Chebyshev nodes in extended precision)
template <typename T> __device__ __forceinline__ T exp2_tab_poly(T s) {
#if CGLB_TAB_BITS == 6
    T p = T(0x1.5d8855325a3d0p-40);
    p = tfma_(p, s, T(0x1.3b2ad54ddd7adp-31));
    p = tfma_(p, s, T(0x1.c6b08d7044d9dp-23));
    p = tfma_(p, s, T(0x1.ebfbdff829821p-15));
    p = tfma_(p, s, T(0x1.62e42fefa39efp-7));
#elif CGLB_TAB_BITS == 8
    T p = T(0x1.3b2ad0e3ae6d5p-39);
    p = tfma_(p, s, T(0x1.c6b090db83bfbp-29));
    p = tfma_(p, s, T(0x1.ebfbdff82beffp-19));
    p = tfma_(p, s, T(0x1.62e42fefa39b8p-9));
#elif CGLB_TAB_BITS == 12
    T p = T(0x1.c6b0809952670p-41);
    p = tfma_(p, s, T(0x1.ebfbdffd0ae72p-27));
    p = tfma_(p, s, T(0x1.62e42fefa39f0p-13));
#else
#error "CGLB_TAB_BITS must be 6, 8 or 12"
#endif
    return tfma_(p, s, T(1));
}
// The table holds 2^(k/T) with (k << (20 - BITS)) subtracted from the high word, so that ONE integer instruction
//   hi(entry[n & (T-1)]) + (n << (20 - BITS))  =  hi(2^(k/T)) + ((n >> BITS) << 20)
// puts the octave count straight into the exponent field (v_lshl_add_u32 instead of v_ashrrev + v_ldexp_f64).  Unlike ldexp
// this cannot underflow gracefully: callers keep n / T inside [-1000, 1000] (range check in set_hypers or the CLAMP variant,
// which then returns 2^-1000 ~ 1e-301 where ldexp would have returned 0).
#define CGLB_EXP_FLOOR_OCT 1000.0
__device__ __forceinline__ double exp2_tab_scale(double entry, int ni) {
    const int hi = __double2hiint(entry) + (int)((unsigned)ni << (20 - CGLB_TAB_BITS));
    return __hiloint2double(hi, __double2loint(entry));
}
template <bool CLAMP> __device__ __forceinline__ double exp2_tab(double xh, const double* __restrict__ tab_lds) {
    if (CLAMP) xh = fmax(xh, -CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE);
    const double n = __builtin_rint(xh);
    const double s = xh - n;
    const int ni = (int)n;
    const double t = tab_lds[ni & (CGLB_TAB_SIZE - 1)];
    return exp2_tab_scale(t, ni) * exp2_tab_poly<double>(s);
}
template <bool CLAMP> __device__ __forceinline__ float exp2_tab(float xh, const double* __restrict__) {
    return __builtin_amdgcn_exp2f(xh * (1.0f / (float)CGLB_TAB_SIZE));
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
// `g` receives the gradient factor h/var (RBF: kappa; Matern32: 3*2^(-r')), see kernels_grad.hip.
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

// Hot-unit forms (operands scaled so that exponents are in 1/T octave, see exp2_tab):
//   RBF:      xh = 8 xs, ah = 64 a          kappa = 2^((ah_i + ah_j + xh_i.xh_j)/64)
//   Matern32: xh = 64 xs, ah = 4096 a       r64 = sqrt(max(ah_i + ah_j - 2 xh_i.xh_j, 0)),  kappa = (1 + r64 ln2/64) 2^(-r64/64)
template <typename T, int KIND, bool CLAMP>
__device__ __forceinline__ T kappa_hot_from_gram(T gram, T aj, const double* __restrict__ tab) {
    if (KIND == CGLB_RBF) {
        return exp2_tab<CLAMP>(gram + aj, tab);
    } else {
        T d2 = tfma<T>(T(-2), gram, aj);
        T r = sqrt_hot(d2);
        return tfma<T>(r, T(CGLB_LN2 / CGLB_HOT_UNITS), T(1)) * exp2_tab<CLAMP>(-r, tab);
    }
}
// Two-phase form of kappa_hot_from_gram for software-pipelined loops: `begin` does the range reduction and issues the
// table read, `end` consumes it, so that a loop can start the R lookups of a column before any of them is needed.
template <typename T> struct KappaPend { T s; T lin; int ni; T tabv; };
// FOLDED (RBF only): the column norm a_j is not added here - the caller has folded 2^(a_j/T) into the column operand.
template <typename T, int KIND, bool CLAMP, bool FOLDED = false>
__device__ __forceinline__ KappaPend<T> kappa_hot_begin(T gram, T aj, const double* __restrict__ tab) {
    KappaPend<T> k;
    T x64;
    if (KIND == CGLB_RBF) {
        x64 = FOLDED ? gram : gram + aj;
        k.lin = T(1);
    } else {
        const T d2 = tfma<T>(T(-2), gram, aj);
        const T r = sqrt_hot(d2);
        k.lin = tfma<T>(r, T(CGLB_LN2 / CGLB_HOT_UNITS), T(1));
        x64 = -r;
    }
    if (sizeof(T) == 4) {  // fp32 has a hardware exp2: no table
        k.s = x64; k.ni = 0; k.tabv = T(0);
        return k;
    }
    if (CLAMP) x64 = tmax<T>(x64, T(-CGLB_EXP_FLOOR_OCT * CGLB_TAB_SIZE));
    const T n = __builtin_rint(x64);
    k.s = x64 - n;
    k.ni = (int)n;
    k.tabv = (T)tab[k.ni & (CGLB_TAB_SIZE - 1)];
    return k;
}
// polynomial part (independent of the table read): overwrites k.s with P5(s) [* lin for Matern]
template <typename T, int KIND> __device__ __forceinline__ void kappa_hot_poly(KappaPend<T>& k) {
    T p;
    if (sizeof(T) == 4) p = (T)__builtin_amdgcn_exp2f((float)k.s * (1.0f / (float)CGLB_TAB_SIZE));
    else p = exp2_tab_poly<T>(k.s);
    if (KIND != CGLB_RBF) p *= k.lin;
    k.s = p;
}
template <typename T, int KIND> __device__ __forceinline__ T kappa_hot_end(const KappaPend<T>& k) {
    if (sizeof(T) == 4) return k.s;
    return (T)(exp2_tab_scale((double)k.tabv, k.ni) * (double)k.s);
}

// gradient factor from an exact squared distance in hot units (RBF: d2h = 64 d2s; Matern32: d2h = 4096 d2s)
template <typename T, int KIND, bool CLAMP> __device__ __forceinline__ T hfac_hot_from_d2(T d2h, const double* __restrict__ tab) {
    if (KIND == CGLB_RBF) {
        return exp2_tab<CLAMP>(T(-0.5) * d2h, tab);
    } else {
        return T(3) * exp2_tab<CLAMP>(-sqrt_pos(d2h), tab);
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
