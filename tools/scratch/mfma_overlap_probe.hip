// mfma_overlap_probe.hip -- how much plain VALU work hides under v_mfma_f32_32x32x16_f16 on one SIMD: a loop of one MFMA (fresh
// accumulator, C = 0) followed by N independent VALU instructions of one kind, at 1 / 2 / 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_overlap_probe.hip -o /tmp/mop && /tmp/mop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int N, int OP, int NM>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x + i;
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(seed * 0.5f + i); }
    v16f acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    const float c0 = seed * 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (NM >= 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc[r & 1]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i % 16]) : "v"(c0));
                if (OP == 1) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i % 16]) : "v"(c0));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + acc[0][i] + acc[1][i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int N, int OP, int NM>
static void run(float *out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 3; ++wps) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL((k<N, OP, NM>), dim3(1024 * wps), dim3(64), 0, 0, out, iters, 1.0f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double ns = best * 1e6 / ((double)wps * iters * 4);
        printf("%d MFMA + %2d x %s, %d wave(s)/SIMD: %.1f ns per group and SIMD = %.1f cycles at 2.4 GHz\n", NM, N, OP ? "v_min3_f32" : "v_add_f32", wps, ns, ns * 2.4);
    }
}
int main() {
    float *out; hipMalloc(&out, 4096 * 64 * 4);
    run<0, 0, 1>(out); run<4, 0, 1>(out); run<8, 0, 1>(out); run<12, 0, 1>(out); run<16, 0, 1>(out); run<24, 0, 1>(out); run<16, 0, 0>(out);
    run<4, 1, 1>(out); run<8, 1, 1>(out); run<12, 1, 1>(out); run<8, 1, 0>(out);
    return 0;
}
