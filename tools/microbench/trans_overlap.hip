// Developer tool: does the quarter-rate fp64 transcendental (v_rsq_f64) overlap with full-rate fp64 VALU work of the same SIMD?
// Kernels (4 waves per SIMD, all CUs): R rsq per iteration alone, F fma per iteration alone, both interleaved in one wave, and
// the mix with the rsq replaced by cvt + v_rsq_f32 + cvt.  If mix == rsq + fma there is no overlap (the issue slot is held for the
// whole quarter-rate pass); if mix == max(.) the pipes are independent.
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/trans_overlap.hip -o /tmp/trans_overlap && /tmp/trans_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_ITER 2048

template <int NR, int NF, int MODE>  // MODE 0: rsq_f64, 1: cvt + rsq_f32 + cvt, 2: v_sqrt_f64
__global__ __launch_bounds__(256) void mix_kernel(double* out, double seed) {
    double r[4], f[24];
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = seed + c + threadIdx.x;
#pragma unroll
    for (int c = 0; c < 24; ++c) f[c] = seed + 0.5 * c + threadIdx.x;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            if (MODE == 0) asm volatile("v_rsq_f64 %0, %0" : "+v"(r[c % 4]));
            else if (MODE == 2) asm volatile("v_sqrt_f64 %0, %0" : "+v"(r[c % 4]));
            else {
                float t;
                asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(r[c % 4]));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(t));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[c % 4]) : "v"(t));
            }
        }
#pragma unroll
        for (int c = 0; c < NF; ++c) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[c % 24]) : "v"(seed));
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += r[c];
#pragma unroll
    for (int c = 0; c < 24; ++c) s += f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K> static float run(K kernel, int cus) {
    const int blocks = cus * 4;
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
    const float rsq = run(mix_kernel<4, 0, 0>, cus), fma = run(mix_kernel<0, 96, 0>, cus), mix = run(mix_kernel<4, 96, 0>, cus);
    const float rsq32 = run(mix_kernel<4, 0, 1>, cus), mix32 = run(mix_kernel<4, 96, 1>, cus);
    const float sq = run(mix_kernel<4, 0, 2>, cus), mixsq = run(mix_kernel<4, 96, 2>, cus);
    printf("per iteration of one wave: 4 v_rsq_f64 %.3f ms | 96 v_fma_f64 %.3f ms | both %.3f ms (sum %.3f, max %.3f)\n", rsq, fma, mix, rsq + fma,
           rsq > fma ? rsq : fma);
    printf("4 x (cvt + v_rsq_f32 + cvt) %.3f ms | with 96 fma %.3f ms\n", rsq32, mix32);
    printf("4 v_sqrt_f64 %.3f ms | with 96 fma %.3f ms\n", sq, mixsq);
    printf("=> one v_rsq_f64 costs %.2f fma issue slots alone, %.2f next to fma work; the f32-seed path %.2f\n", rsq / 4 / (fma / 96),
           (mix - fma) / 4 / (fma / 96), (mix32 - fma) / 4 / (fma / 96));
    return 0;
}
