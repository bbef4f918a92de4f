// Developer tool: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the pair kernels use.
// Each kernel runs N_ITER iterations of 8 independent chains of one instruction; 4 waves per SIMD keep the pipe full.
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096
#define CHAINS 8

#define DEF_KERNEL(NAME, TYPE, INIT, BODY)                                              \
    __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) {                 \
        TYPE x[CHAINS];                                                                 \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) x[c] = INIT;                 \
        for (int it = 0; it < N_ITER; ++it) {                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { BODY; }                \
        }                                                                               \
        TYPE s = 0;                                                                     \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) s += x[c];                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                 \
    }

DEF_KERNEL(k_fma_f64, double, seed + c + threadIdx.x, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_add_f64, double, seed + c + threadIdx.x, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_mul_f64, double, seed + c + threadIdx.x, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_rndne_f64, double, seed + c + threadIdx.x, asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c])))
DEF_KERNEL(k_fract_f64, double, seed + c + threadIdx.x, asm volatile("v_fract_f64 %0, %0" : "+v"(x[c])))
DEF_KERNEL(k_ldexp_f64, double, seed + c + threadIdx.x, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[c])))
DEF_KERNEL(k_max_f64, double, seed + c + threadIdx.x, asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_rsq_f64, double, seed + c + threadIdx.x, asm volatile("v_rsq_f64 %0, %0" : "+v"(x[c])))
DEF_KERNEL(k_fma_f32, float, seed + c + threadIdx.x, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_exp_f32, float, seed + c + threadIdx.x, asm volatile("v_exp_f32 %0, %0" : "+v"(x[c])))
DEF_KERNEL(k_pk_fma_f32, double, seed + c + threadIdx.x, asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_lshl_add_u32, int, (int)seed + c + threadIdx.x, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_add_u32, int, (int)seed + c + threadIdx.x, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(seed)))
DEF_KERNEL(k_sdwa_shl, int, (int)seed + c + threadIdx.x,
           asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(x[c]) : "v"(seed)))

// cvt needs a type change: chain double -> int -> double through two instructions
__global__ __launch_bounds__(256) void k_cvt_pair(double* out, double seed) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + c + threadIdx.x;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            int t;
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(x[c]));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[c]) : "v"(t));
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, typename K>
static void run(const char* name, K kernel, T seed, double instr_per_iter_per_chain, double clock_ghz, int cus) {
    const int blocks = cus * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    T* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x N_ITER x CHAINS x instr wave-instructions
    const double wave_instr = 4.0 * N_ITER * CHAINS * instr_per_iter_per_chain;
    const double cycles = ms * 1e-3 * clock_ghz * 1e9;
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz)\n", name, ms, cycles / wave_instr, clock_ghz);
    hipFree(out);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
    run<double>("v_fma_f64", k_fma_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_add_f64", k_add_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_mul_f64", k_mul_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_max_f64", k_max_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_rndne_f64", k_rndne_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_fract_f64", k_fract_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_ldexp_f64", k_ldexp_f64, 1.0000001, 1, ghz, cus);
    run<double>("v_rsq_f64", k_rsq_f64, 1.0000001, 1, ghz, cus);
    run<double>("cvt i32<->f64 x2", k_cvt_pair, 1.0000001, 2, ghz, cus);
    run<float>("v_fma_f32", k_fma_f32, 1.0000001f, 1, ghz, cus);
    run<float>("v_exp_f32", k_exp_f32, 1.0000001f, 1, ghz, cus);
    run<double>("v_pk_fma_f32", k_pk_fma_f32, 1.0000001, 1, ghz, cus);
    run<int>("v_lshl_add_u32", k_lshl_add_u32, 3, 1, ghz, cus);
    run<int>("v_add_u32", k_add_u32, 3, 1, ghz, cus);
    run<int>("v_lshlrev_sdwa", k_sdwa_shl, 3, 1, ghz, cus);
    return 0;
}
