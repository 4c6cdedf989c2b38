// pk_mix_probe.hip -- do packed f32 operations (half rate) leave room for plain ones?  Per iteration NP v_pk_fma_f32 and NS plain v_fma_f32 /
// v_add_f32, all independent chains, interleaved; 4 waves per SIMD.  If the time of a mix is max(NP x 4.7, NS x 2.4) rather than the sum,
// mfcc_kernel (281 packed + 164 plain instructions per tile) should trade packed for plain until the two sides balance.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/pk_mix_probe.hip -o /tmp/pmp && /tmp/pmp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NP, int NS, int PLAIN>   // PLAIN 0: v_fma_f32, 1: v_add_f32
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    v2f p[8], c = {seed, seed * 0.5f}, d = {0.25f, 0.125f};
    float s[8], e = seed * 0.75f, f = 0.5f;
    for (int i = 0; i < 8; ++i) { p[i] = (v2f){seed + i, seed - i}; s[i] = seed + threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(r * NP + i) & 7]) : "v"(c), "v"(d));
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (PLAIN == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(r * NS + i) & 7]) : "v"(e), "v"(f));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[(r * NS + i) & 7]) : "v"(e));
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += p[i].x + p[i].y + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NP, int NS, int PLAIN>
static double run() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * 4, iters = 4000;
    float *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, NS, PLAIN>), dim3(blocks), dim3(256), 0, 0, out, 200, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, NS, PLAIN>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms * 1e-3 * 2.4e9 / iters / 8.0 / 4.0;   // nominal SIMD cycles per (NP packed + NS plain) group of one wave
}

int main() {
    printf("nominal SIMD cycles per group of one wave (4 waves per SIMD)\n");
    printf("plain = v_fma_f32: 4 packed %.1f | 4 plain %.1f | 4 + 4 %.1f | 4 + 8 %.1f | 8 plain %.1f | 2 + 8 %.1f\n", run<4, 0, 0>(), run<0, 4, 0>(), run<4, 4, 0>(), run<4, 8, 0>(),
           run<0, 8, 0>(), run<2, 8, 0>());
    printf("plain = v_add_f32: 4 packed %.1f | 4 plain %.1f | 4 + 4 %.1f | 4 + 8 %.1f | 8 plain %.1f | 2 + 8 %.1f\n", run<4, 0, 1>(), run<0, 4, 1>(), run<4, 4, 1>(), run<4, 8, 1>(),
           run<0, 8, 1>(), run<2, 8, 1>());
    return 0;
}
