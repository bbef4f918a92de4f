// Developer tool: rate and semantics of v_fmac_f64 with the DPP control row_newbcast (gfx90a+): D += lane(16 * (l / 16) + k of src0) * src1.
// If it issues at the rate of a plain v_fma_f64, one VGPR pair can hand 16 different wave-"uniform" operands to 16 consecutive fmas
// (lane l of every 16-lane row holds operand l), instead of 32 SGPRs fetched by scalar loads.
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/dpp_fmac.hip -o /tmp/dpp_fmac && /tmp/dpp_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define N_ITER 2048

#define FMAC_DPP(acc, bc, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(x))

template <int MODE, int NACC>  // MODE 0: plain v_fma_f64 (VGPR operands), 1: v_fmac_f64_dpp row_newbcast; NACC independent accumulators (1: one dependent chain)
__global__ __launch_bounds__(256) void rate_kernel(double* out, double seed) {
    double acc[8], x[16], bc = seed * (threadIdx.x & 15);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = seed + c;
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = seed * 1e-3 * c + threadIdx.x * 1e-6;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[(rep * 16 + k) % NACC]) : "v"(bc), "v"(x[k]));
            } else {
                FMAC_DPP(acc[(rep * 16 + 0) % NACC], bc, x[0], 0);   FMAC_DPP(acc[(rep * 16 + 1) % NACC], bc, x[1], 1);
                FMAC_DPP(acc[(rep * 16 + 2) % NACC], bc, x[2], 2);   FMAC_DPP(acc[(rep * 16 + 3) % NACC], bc, x[3], 3);
                FMAC_DPP(acc[(rep * 16 + 4) % NACC], bc, x[4], 4);   FMAC_DPP(acc[(rep * 16 + 5) % NACC], bc, x[5], 5);
                FMAC_DPP(acc[(rep * 16 + 6) % NACC], bc, x[6], 6);   FMAC_DPP(acc[(rep * 16 + 7) % NACC], bc, x[7], 7);
                FMAC_DPP(acc[(rep * 16 + 8) % NACC], bc, x[8], 8);   FMAC_DPP(acc[(rep * 16 + 9) % NACC], bc, x[9], 9);
                FMAC_DPP(acc[(rep * 16 + 10) % NACC], bc, x[10], 10); FMAC_DPP(acc[(rep * 16 + 11) % NACC], bc, x[11], 11);
                FMAC_DPP(acc[(rep * 16 + 12) % NACC], bc, x[12], 12); FMAC_DPP(acc[(rep * 16 + 13) % NACC], bc, x[13], 13);
                FMAC_DPP(acc[(rep * 16 + 14) % NACC], bc, x[14], 14); FMAC_DPP(acc[(rep * 16 + 15) % NACC], bc, x[15], 15);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// semantics: out[l] = sum_k src0[lane 16 (l/16) + k] * x_k(l)
__global__ void check_kernel(double* out, const double* a, const double* x) {
    const int l = threadIdx.x;
    double acc = 0.0, bc = a[l];
    double xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = x[l * 16 + k];
    FMAC_DPP(acc, bc, xv[0], 0);   FMAC_DPP(acc, bc, xv[1], 1);   FMAC_DPP(acc, bc, xv[2], 2);   FMAC_DPP(acc, bc, xv[3], 3);
    FMAC_DPP(acc, bc, xv[4], 4);   FMAC_DPP(acc, bc, xv[5], 5);   FMAC_DPP(acc, bc, xv[6], 6);   FMAC_DPP(acc, bc, xv[7], 7);
    FMAC_DPP(acc, bc, xv[8], 8);   FMAC_DPP(acc, bc, xv[9], 9);   FMAC_DPP(acc, bc, xv[10], 10); FMAC_DPP(acc, bc, xv[11], 11);
    FMAC_DPP(acc, bc, xv[12], 12); FMAC_DPP(acc, bc, xv[13], 13); FMAC_DPP(acc, bc, xv[14], 14); FMAC_DPP(acc, bc, xv[15], 15);
    out[l] = acc;
}

template <typename K> static float run(K kernel, int cus, int wgs_per_cu) {
    const int blocks = cus * wgs_per_cu;
    double* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // semantics first
    double ha[64], hx[64 * 16], ho[64], *da, *dx, *dout;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + 0.37 * l; for (int k = 0; k < 16; ++k) hx[l * 16 + k] = 0.01 * (k + 1) + 1e-4 * l; }
    hipMalloc(&da, sizeof(ha)); hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, dout, da, dx);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l) {
        double ref = 0;
        for (int k = 0; k < 16; ++k) ref = fma(ha[16 * (l / 16) + k], hx[l * 16 + k], ref);
        worst = fmax(worst, fabs(ref - ho[l]));
    }
    printf("semantics (row_newbcast:k = lane 16 (l / 16) + k of src0): max deviation %.3g\n", worst);
    for (int w : {1, 2, 4}) {
        const float a8 = run(rate_kernel<0, 8>, cus, w), d8 = run(rate_kernel<1, 8>, cus, w);
        const float a1 = run(rate_kernel<0, 1>, cus, w), d1 = run(rate_kernel<1, 1>, cus, w);
        const float a2 = run(rate_kernel<0, 2>, cus, w), d2 = run(rate_kernel<1, 2>, cus, w);
        const double n = (double)N_ITER * 96;
        const double clk = 2.4e9;  // nominal; the ratio is what matters
        printf("%d wave(s) per SIMD: 8 chains: v_fma_f64 %.3f ms (%.2f nominal cycles per instruction), v_fmac_f64_dpp %.3f ms (%.2f); one dependent chain: %.3f ms (%.2f) / %.3f ms (%.2f); two interleaved chains: %.3f ms (%.2f) / %.3f ms (%.2f)\n", w,
               a8, a8 * 1e-3 * clk / n / w, d8, d8 * 1e-3 * clk / n / w, a1, a1 * 1e-3 * clk / n / w, d1, d1 * 1e-3 * clk / n / w, a2, a2 * 1e-3 * clk / n / w, d2, d2 * 1e-3 * clk / n / w);
    }
    return 0;
}
