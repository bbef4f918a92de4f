// Developer tool: do v_mfma_f64_16x16x4_f64 and fp64 VALU instructions overlap on gfx950, or do they share the DP datapath?
// Four kernels over the same iteration count: MFMA only, fp64 FMA only, both in one wave (independent streams), and
// MFMA waves next to VALU waves on every SIMD (512-thread workgroup: waves 0-3 matrix, 4-7 vector).  If the mixed
// kernels take max(a, b) the pipes are separate; if they take a + b the Gram part of the pair kernels gains nothing on MFMA.
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64_overlap.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_ITER 2048
typedef double double4_t __attribute__((ext_vector_type(4)));

#define MFMA_STEP(ACC, A, B) ACC = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, ACC, 0, 0, 0)
#define FMA_STEP(X, S) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(X) : "v"(S))

// MODE 0: 4 MFMA per iteration; 1: VPI fp64 FMAs per iteration; 2: both in the same wave; 3: waves 0-3 MFMA, waves 4-7 FMA
template <int MODE, int VPI>
__global__ __launch_bounds__(512) void k_mix(double* out, double seed) {
    const int wave = threadIdx.x >> 6;
    double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{seed, seed, seed, seed};
    double x[8];
    for (int c = 0; c < 8; ++c) x[c] = seed + c + threadIdx.x;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    if (MODE == 3) {
        if (do_mfma) {
            for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA_STEP(acc[i], a, b);
            }
        } else {
            for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
                for (int v = 0; v < VPI; ++v) FMA_STEP(x[v & 7], seed);
            }
        }
    } else {
        for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (do_mfma) MFMA_STEP(acc[i], a, b);
                if (do_valu) {
#pragma unroll
                    for (int v = 0; v < VPI / 4; ++v) FMA_STEP(x[(i * (VPI / 4) + v) & 7], seed);
                }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int c = 0; c < 8; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run(const char* name, K kernel, int threads, int blocks, double ghz, double mfma_per_simd, double valu_per_simd) {
    double* out;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-44s %8.3f ms  %9.0f cycles", name, ms, cycles);
    if (mfma_per_simd > 0) printf("  %6.1f cycles/MFMA", cycles / mfma_per_simd);
    if (valu_per_simd > 0) printf("  %6.2f cycles/FMA", cycles / valu_per_simd);
    printf("\n");
    hipFree(out);
    return cycles;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, %.2f GHz nominal; one workgroup per CU\n", prop.name, cus, ghz);
    const double M = 4.0 * N_ITER;  // MFMAs per wave
    // 256 threads: one wave per SIMD
    run("MFMA f64 only, 1 wave/SIMD", k_mix<0, 0>, 256, cus, ghz, M, 0);
    run("MFMA f64 only, 2 waves/SIMD", k_mix<0, 0>, 512, cus, ghz, 2 * M, 0);
    run("FMA f64 only (64/iter), 1 wave/SIMD", k_mix<1, 64>, 256, cus, ghz, 0, 64.0 * N_ITER);
    run("FMA f64 only (64/iter), 2 waves/SIMD", k_mix<1, 64>, 512, cus, ghz, 0, 2 * 64.0 * N_ITER);
    run("same wave: 4 MFMA + 32 FMA /iter, 1 wave/SIMD", k_mix<2, 32>, 256, cus, ghz, M, 32.0 * N_ITER);
    run("same wave: 4 MFMA + 64 FMA /iter, 1 wave/SIMD", k_mix<2, 64>, 256, cus, ghz, M, 64.0 * N_ITER);
    run("same wave: 4 MFMA + 64 FMA /iter, 2 waves/SIMD", k_mix<2, 64>, 512, cus, ghz, 2 * M, 2 * 64.0 * N_ITER);
    run("split waves: 4 MFMA | 32 FMA /iter", k_mix<3, 32>, 512, cus, ghz, M, 32.0 * N_ITER);
    run("split waves: 4 MFMA | 64 FMA /iter", k_mix<3, 64>, 512, cus, ghz, M, 64.0 * N_ITER);
    return 0;
}
