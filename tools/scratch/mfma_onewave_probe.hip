// mfma_onewave_probe.hip -- can ONE wave per SIMD feed both pipes?  Per iteration 120 vector instructions (v_min3_f32 / v_add_f32 pairs) in
// NC independent dependency chains (2 = what a tile's two template pairs give, 4 = two columns in flight, 6) with 0 or 6 chained
// v_mfma_f32_32x32x16_bf16 spread between them; 1, 2, 3 waves per SIMD.  Nominal (2.4 GHz) SIMD cycles per wave-iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_onewave_probe.hip -o /tmp/mop && /tmp/mop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NC, int NM>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
    v16f acc[3];
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + i + (unsigned)seed; b[i] = 0x3f003f00u + threadIdx.x; }
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = seed;
    float x[NC], p0 = seed, p1 = seed * 3.f;
    for (int i = 0; i < NC; ++i) x[i] = seed + threadIdx.x + i;
    constexpr int ROUNDS = 60 / NC;   // 60 min3 + 60 add = 120 vector instructions
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (NM && r % (ROUNDS / NM) == 0 && r / (ROUNDS / NM) < NM) {
                const int m = r / (ROUNDS / NM);
                if (m < 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[m]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m - 3]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(p0), "v"(p1));
#pragma unroll
            for (int i = 0; i < NC; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(p0));
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0.f;
    for (int i = 0; i < NC; ++i) s += x[i];
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NC, int NM>
static double run(int waves_per_simd) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * 4 * waves_per_simd, iters = 4000;
    float *out; (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NC, NM>), dim3(blocks), dim3(64), 0, 0, out, 200, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NC, NM>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms * 1e-3 * 2.4e9 / iters / waves_per_simd;
}

int main() {
    printf("nominal SIMD cycles per wave-iteration (120 vector instructions; none / six matrix instructions)\n");
    for (int w = 1; w <= 3; ++w)
        printf("waves/SIMD %d: 2 chains %.0f / %.0f | 4 chains %.0f / %.0f | 6 chains %.0f / %.0f | 10 chains %.0f / %.0f\n", w, run<2, 0>(w), run<2, 6>(w),
               run<4, 0>(w), run<4, 6>(w), run<6, 0>(w), run<6, 6>(w), run<10, 0>(w), run<10, 6>(w));
    return 0;
}
