// mfma_valu_overlap_probe.hip -- does a SIMD issue vector work while v_mfma_f32_32x32x16_f16 runs?  Per iteration a wave issues NM matrix
// instructions (three independent accumulator tiles, chained in k-steps as dtw_mfma_wide_kernel does) and NV vector instructions
// (two DEPENDENT chains of v_min3_f32 + v_add_f32, as the recurrence of the DTW kernels), either in bursts (all matrix work, then all
// vector work) or interleaved (one matrix instruction every NV / NM vector ones).  Prints SIMD cycles per iteration at 1, 2, 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_valu_overlap_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int MODE>   // MODE 0: burst, 1: interleaved, 2: vector only, 3: matrix only
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
    v16f acc[3];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = seed;
    float x = seed + threadIdx.x, y = seed * 2.f, p0 = seed, p1 = seed * 3.f, q0 = seed * 5.f, q1 = seed * 7.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 3], 0, 0, 0);
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p0), "v"(p1));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(y) : "v"(q0), "v"(q1));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(p0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(q0));
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < NV / 4 / NM; ++v) {
                    asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p0), "v"(p1));
                    asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(y) : "v"(q0), "v"(q1));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(p0));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(q0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = x + y;
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NM, int NV, int MODE>
static double run(int waves_per_simd) {
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const int cus = pr.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 4000;
    float *out; hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double ghz = 2.4;   // nominal: cycles below are "nominal cycles" as in profiles/valu_rate_table.json
    return ms * 1e-3 * ghz * 1e9 / iters / waves_per_simd;   // SIMD cycles per wave-iteration
}

int main() {
    printf("SIMD cycles (at the nominal 2.4 GHz) per wave-iteration of 9 matrix + 160 vector instructions (two dependent min3/add chains)\n");
    for (int w = 1; w <= 3; ++w) {
        printf("waves/SIMD %d: vector only %.0f  matrix only %.0f  burst %.0f  interleaved %.0f\n", w, run<9, 160, 2>(w), run<9, 160, 3>(w),
               run<9, 160, 0>(w), run<9, 160, 1>(w));
    }
    printf("12 matrix + 160 vector:\n");
    for (int w = 1; w <= 3; ++w)
        printf("waves/SIMD %d: matrix only %.0f  burst %.0f  interleaved %.0f\n", w, run<12, 160, 3>(w), run<12, 160, 0>(w), run<12, 160, 1>(w));
    printf("3 matrix + 108 vector (dtw_mfma_kernel's column):\n");
    for (int w = 1; w <= 3; ++w)
        printf("waves/SIMD %d: vector only %.0f  matrix only %.0f  burst %.0f  interleaved %.0f\n", w, run<3, 108, 2>(w), run<3, 108, 3>(w),
               run<3, 108, 0>(w), run<3, 108, 1>(w));
    return 0;
}
