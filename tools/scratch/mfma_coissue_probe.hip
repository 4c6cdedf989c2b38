// mfma_coissue_probe.hip -- do the vector and the matrix pipe of one SIMD run at the same time when DIFFERENT waves feed them?  Workgroups of
// eight waves, one per CU (256 workgroups): waves 0..3 issue only v_mfma_f32_32x32x16_bf16 (independent accumulators), waves 4..7 only vector
// instructions (eight independent v_min3 / v_add chains: issue-bound, not latency-bound); a wave lands on SIMD (wave id mod 4), so every SIMD
// holds one wave of each kind.  Time of: the matrix waves alone, the vector waves alone, both.  both ~ max(...) = the pipes overlap;
// both ~ sum = they take turns.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_coissue_probe.hip -o /tmp/mcp && /tmp/mcp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// variant 0: as described; 1: the vector waves at s_setprio 2; 2: the roles swapped (waves 0..3 = the OLDER ones do the vector work);
// 3: the matrix waves lower their own priority (s_setprio 0) and the vector waves run at 1 -- what a kernel whose waves do both could do
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed, int mode, int nv, int variant) {   // mode 1: matrix waves, 2: vector waves, 3: both
    int wave = threadIdx.x >> 6;
    if (variant == 2) wave ^= 4;
    float s = 0.f;
    if (wave >= 4 && (variant == 1)) __builtin_amdgcn_s_setprio(2);
    if (wave >= 4 && (variant == 3)) __builtin_amdgcn_s_setprio(1);
    if (wave < 4) {
        if (!(mode & 1)) return;
        v16f acc[4];
        u32x4 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + i + (unsigned)seed; b[i] = 0x3f003f00u + threadIdx.x; }
        for (int g = 0; g < 4; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15");
        for (int g = 0; g < 4; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
    } else {
        if (!(mode & 2)) return;
        float x[8], p0 = seed, p1 = seed * 3.f;
        for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x + i;
        for (int it = 0; it < iters; ++it) {
            for (int r = 0; r < nv; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(p0), "v"(p1));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(p0));
            }
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static double run(int mode, int nv, int variant = 0) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount, iters = 4000;
    float *out; (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 200, 1.f, mode, nv, variant);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, iters, 1.f, mode, nv, variant);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms * 1e3 / iters * 1e3;   // ns per iteration
}

int main() {
    printf("ns per iteration; one iteration = 16 matrix instructions (512 pipe cycles) on the matrix wave, nv x 16 vector instructions on the vector wave of the same SIMD\n");
    for (int nv : {4, 8, 12, 16}) {
        const double m = run(1, nv), v = run(2, nv), b = run(3, nv);
        printf("nv %2d (%3d vector instructions): matrix alone %.0f  vector alone %.0f  both %.0f   (max %.0f, sum %.0f) | both, vector waves at s_setprio 2: %.0f | roles swapped (vector waves older): %.0f | vector waves at prio 1: %.0f\n",
               nv, nv * 16, m, v, b, m > v ? m : v, m + v, run(3, nv, 1), run(3, nv, 2), run(3, nv, 3));
    }
    return 0;
}
