#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    int l = threadIdx.x;
    // A_i = 10*(l%4) + 100*(l/4) + 1 ; B_j = (l%4)+1 + 0.001*(l/4)
    float a = 1.f + 10.f * (l % 4) + 100.f * (l / 4);
    float b = 1.f + (l % 4);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
template <int N> __global__ void thr(float* out, int iters) {
    int l = threadIdx.x;
    float a = l * 0.001f, b = 1.f + l;
    f32x4 c[N];
    for (int n = 0; n < N; ++n) c[n] = (f32x4){0.f, 0.f, 0.f, (float)n};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[n], 0, 0, 0);
    }
    float s = 0; for (int n = 0; n < N; ++n) s += c[n][0] + c[n][3];
    out[blockIdx.x * 64 + l] = s;
}
// mixed: N mfma chains + M independent valu fma per iteration
template <int N, int M> __global__ void mix(float* out, int iters) {
    int l = threadIdx.x;
    float a = l * 0.001f, b = 1.f + l;
    f32x4 c[N > 0 ? N : 1]; float v[M];
    for (int n = 0; n < N; ++n) c[n] = (f32x4){0.f, 0.f, 0.f, (float)n};
    for (int m = 0; m < M; ++m) v[m] = (float)m;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) v[m] = fmaf(v[m], a, b);
    }
    float s = 0; for (int n = 0; n < N; ++n) s += c[n][0] + c[n][3]; for (int m = 0; m < M; ++m) s += v[m];
    out[blockIdx.x * 64 + l] = s;
}
int main() {
    float* d; hipMalloc(&d, 1 << 22); float h[256];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 63}) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 20000; float ms;
    auto t = [&](const char* name, auto launch, double inst) { launch(); hipDeviceSynchronize(); hipEventRecord(a,0); launch(); hipEventRecord(b,0); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b);
        printf("%s: %.3f ms -> %.2f cycles/iter/wave-slot @2.4GHz\n", name, ms, ms*1e-3*2.4e9/iters/inst); };
    // 1024 blocks of 64 threads = 1 wave per SIMD; also 4 waves per SIMD
    t("mfma x8 chains, 1 wave/SIMD (per mfma)", [&]{ hipLaunchKernelGGL(thr<8>, dim3(1024), dim3(64), 0, 0, d, iters); }, 8);
    t("mfma x8 chains, 4 wave/SIMD (per mfma per wave)", [&]{ hipLaunchKernelGGL(thr<8>, dim3(4096), dim3(64), 0, 0, d, iters); }, 8*4);
    t("valu only x16, 4 wave/SIMD (per fma per wave)", [&]{ hipLaunchKernelGGL((mix<0,16>), dim3(4096), dim3(64), 0, 0, d, iters); }, 16*4);
    t("mix 8 mfma + 16 fma, 4 wave/SIMD (per iter per wave)", [&]{ hipLaunchKernelGGL((mix<8,16>), dim3(4096), dim3(64), 0, 0, d, iters); }, 4);
    t("mix 8 mfma + 32 fma, 4 wave/SIMD (per iter per wave)", [&]{ hipLaunchKernelGGL((mix<8,32>), dim3(4096), dim3(64), 0, 0, d, iters); }, 4);
    t("mix 0 mfma + 32 fma, 4 wave/SIMD (per iter per wave)", [&]{ hipLaunchKernelGGL((mix<0,32>), dim3(4096), dim3(64), 0, 0, d, iters); }, 4);
    return 0;
}
