// mfma_shape_probe.hip -- what does a column's matrix work cost the vector work beside it, by instruction shape?  Per iteration a wave issues
// NV vector instructions (two dependent chains of v_min3_f32 + v_add_f32) with the matrix instructions spread evenly between them:
//   shape 0: 6 x v_mfma_f32_32x32x16_bf16, three tiles x (C = 0, then C = the first's result)   [dtw_mfma_kernel<..., P3>'s column]
//   shape 1: 12 x v_mfma_f32_16x16x32_bf16, C = 0                                               [the same 31 k-slots in one instruction per tile]
//   shape 2: 6 x v_mfma_f32_32x32x16_bf16, all C = 0
//   shape 3: 3 x v_mfma_f32_32x32x16_bf16, C = 0                                                [the two-part form's column]
//   shape 4: none
//   shape 5: 12 x v_mfma_f32_32x32x16_bf16, two tiles x six chained k-steps, first C = 1.0   [dtw_mfma_wide3_kernel's column]
//   shape 6: 18 x v_mfma_f32_16x16x32_bf16, six tiles x three chained k-steps                [the same products on 12 row slots instead of 16]
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_shape_probe.hip -o /tmp/msp && /tmp/msp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define VEC4()                                                                  \
    do {                                                                        \
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p0), "v"(p1)); \
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(y) : "v"(q0), "v"(q1)); \
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(p0));               \
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(q0));               \
    } while (0)

template <int SHAPE, int NV>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
    v16f acc[3];
    v4f acc4[12];
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + i + (unsigned)seed; b[i] = 0x3f003f00u + threadIdx.x; }
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = seed;
    for (int g = 0; g < 12; ++g) for (int i = 0; i < 4; ++i) acc4[g][i] = seed;
    float x = seed + threadIdx.x, y = seed * 2.f, p0 = seed, p1 = seed * 3.f, q0 = seed * 5.f, q1 = seed * 7.f;
    constexpr int NM = SHAPE == 0 ? 6 : SHAPE == 1 ? 12 : SHAPE == 2 ? 6 : SHAPE == 3 ? 3 : SHAPE == 5 ? 12 : SHAPE == 6 ? 18 : 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SHAPE == 0) {
                if (m < 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[m]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m - 3]) : "v"(a), "v"(b));
            } else if (SHAPE == 1) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc4[m]) : "v"(a), "v"(b));
            } else if (SHAPE == 5) {
                if (m % 6 == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 1.0" : "=&v"(acc[m / 6]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m / 6]) : "v"(a), "v"(b));
            } else if (SHAPE == 6) {
                if (m % 3 == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 1.0" : "=&v"(acc4[m / 3]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[m / 3]) : "v"(a), "v"(b));
            } else if (SHAPE == 2 || SHAPE == 3) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[m % 3]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int v = 0; v < NV / 4 / NM; ++v) VEC4();
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = x + y;
    for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
    for (int g = 0; g < 12; ++g) for (int i = 0; i < 4; ++i) s += acc4[g][i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int SHAPE, int NV>
static double run(int waves_per_simd) {
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const int cus = pr.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 4000;
    float *out; hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NV>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NV>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e-3 * 2.4e9 / iters / waves_per_simd;   // nominal SIMD cycles per wave-iteration
}

int main() {
    printf("nominal SIMD cycles per wave-iteration of 120 vector instructions + the column's matrix instructions\n");
    for (int w = 1; w <= 3; ++w)
        printf("waves/SIMD %d: none %.0f | 3x32x32x16 %.0f | 6x32x32x16 chained %.0f | 6x32x32x16 C=0 %.0f | 12x16x16x32 C=0 %.0f\n", w, run<4, 120>(w),
               run<3, 120>(w), run<0, 120>(w), run<2, 120>(w), run<1, 120>(w));
    printf("132 vector instructions (dtw_mfma_wide3_kernel's column):\n");
    for (int w = 1; w <= 3; ++w)
        printf("waves/SIMD %d: none %.0f | 12x32x32x16 (2 chains of 6) %.0f | 18x16x16x32 (6 chains of 3) %.0f\n", w, run<4, 144>(w), run<5, 144>(w), run<6, 144>(w));
    return 0;
}
