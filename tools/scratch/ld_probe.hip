// scratch: streaming-read patterns over x [B][3120] f32
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// pattern A: MFMA layout: lane (li = l&15 row, lk = l>>4) reads float4 at row*in + 16*kb + 4*lk, KU blocks in flight
template <int KU> __global__ void patA(const float* __restrict__ x, size_t B, int in, float* out) {
    int l = threadIdx.x & 63, wave = threadIdx.x >> 6, li = l & 15, lk = l >> 4;
    size_t row = ((size_t)blockIdx.x * 4 + wave) * 16 + li;
    float s = 0.f;
    for (int kb = 0; kb < in / 16; kb += KU) {
        float4 a[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { int k0 = 16 * (kb + u) + 4 * lk; a[u] = (k0 + 3 < in) ? *reinterpret_cast<const float4*>(x + row * in + k0) : make_float4(0,0,0,0); }
#pragma unroll
        for (int u = 0; u < KU; ++u) s += a[u].x + a[u].y + a[u].z + a[u].w;
    }
    if (s == 1.2345f) out[row] = s;
}
// pattern B: one row per 16-lane group, but each lane reads CONTIGUOUS 16 B across the 4 lk lanes x KU: row chunk of 64 floats per instruction per row
// pattern C: whole wave reads 1 KB contiguous of one row per instruction (row-major streaming), 16 rows in turn
template <int KU> __global__ void patC(const float* __restrict__ x, size_t B, int in, float* out) {
    int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t row0 = ((size_t)blockIdx.x * 4 + wave) * 16;
    float s = 0.f;
    int nchunk = in / 256;  // 256 floats = 1 KB per wave instruction
    for (int r = 0; r < 16; ++r) {
        const float* p = x + (row0 + r) * in;
        for (int c = 0; c < nchunk; c += KU) {
            float4 a[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) { int k0 = 256 * (c + u) + 4 * l; a[u] = (k0 + 3 < in) ? *reinterpret_cast<const float4*>(p + k0) : make_float4(0,0,0,0); }
#pragma unroll
            for (int u = 0; u < KU; ++u) s += a[u].x + a[u].y + a[u].z + a[u].w;
        }
    }
    if (s == 1.2345f) out[row0] = s;
}
int main() {
    size_t B = 65536; int in = 3120;
    float *x, *out; CK(hipMalloc(&x, B * in * 4)); CK(hipMalloc(&out, B * 4)); CK(hipMemset(x, 0x3c, B * in * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        for (int it = 0; it < 3; ++it) { hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            if (it == 2) printf("%s %.3f ms %.2f TB/s\n", name, ms, B * in * 4.0 / ms / 1e9); }
    };
    run("patA KU=1", [&] { hipLaunchKernelGGL(patA<1>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    run("patA KU=4", [&] { hipLaunchKernelGGL(patA<4>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    run("patA KU=8", [&] { hipLaunchKernelGGL(patA<8>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    run("patA KU=16", [&] { hipLaunchKernelGGL(patA<16>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    run("patC KU=4", [&] { hipLaunchKernelGGL(patC<4>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    run("patC KU=12", [&] { hipLaunchKernelGGL(patC<12>, dim3(B / 64), dim3(256), 0, 0, x, B, in, out); });
    return 0;
}
