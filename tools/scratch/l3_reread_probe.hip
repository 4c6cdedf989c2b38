// Probe for the front-end fusion question (round 4): if a workgroup reads its rows' samples LEAD tiles ahead (first touch, what a
// fused RMS wave would do) and again when it filters them, does the second read come out of the 256 MiB Infinity Cache or out of HBM?
// Layout as rp_frontend_batch at C3 size: S rows of N i16 samples in, f32 out; a workgroup owns 64 rows and walks them in 64-sample
// tiles (128 B in, 256 B out per row and tile).  Variants: 0 = read tile k, write tile k; 1 = also read tile k + LEAD; 2 = as 1
// with non-temporal stores; 3 = as 1 with non-temporal look-ahead loads.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l3probe tools/scratch/l3_reread_probe.hip ; run: /tmp/l3probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(128) void walk(const short *__restrict__ in, float *__restrict__ out, size_t N, int lead, int *sink) {
    const int t = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const size_t n_tiles = N / 64;
    int acc = 0;
    // tile = 64 rows x 128 B = 512 pieces of 16 B: piece p -> row p / 8, part p % 8
    for (size_t k = 0; k < n_tiles; ++k) {
        v4i cur[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 128 + t;
            cur[i] = *reinterpret_cast<const v4i *>(in + (row0 + p / 8) * N + k * 64 + (p % 8) * 8);
        }
        if (MODE >= 1 && k + lead < n_tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = i * 128 + t;
                const v4i *q = reinterpret_cast<const v4i *>(in + (row0 + p / 8) * N + (k + lead) * 64 + (p % 8) * 8);
                const v4i a = MODE == 3 ? __builtin_nontemporal_load(q) : *q;
                acc += a.x ^ a.y ^ a.z ^ a.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 128 + t;
            const v4i c = cur[i];
            float *o = out + (row0 + p / 8) * N + k * 64 + (p % 8) * 8;
            const v4f lo = {(float)(short)(c.x & 0xffff), (float)(c.x >> 16), (float)(short)(c.y & 0xffff), (float)(c.y >> 16)};
            const v4f hi = {(float)(short)(c.z & 0xffff), (float)(c.z >> 16), (float)(short)(c.w & 0xffff), (float)(c.w >> 16)};
            if (MODE == 2) {
                __builtin_nontemporal_store(lo, reinterpret_cast<v4f *>(o));
                __builtin_nontemporal_store(hi, reinterpret_cast<v4f *>(o + 4));
            } else {
                *reinterpret_cast<v4f *>(o) = lo;
                *reinterpret_cast<v4f *>(o + 4) = hi;
            }
        }
    }
    if (acc == 0x12345678) *sink = acc;
}
int main(int argc, char **argv) {
    const size_t S = 65536, N = 64000;
    const int lead = argc > 1 ? atoi(argv[1]) : 9;
    short *in; float *out; int *sink;
    hipMalloc(&in, S * N * 2); hipMalloc(&out, S * N * 4); hipMalloc(&sink, 4);
    hipMemset(in, 1, S * N * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(walk<0>, dim3(S / 64), dim3(128), 0, 0, in, out, N, lead, sink);
            if (mode == 1) hipLaunchKernelGGL(walk<1>, dim3(S / 64), dim3(128), 0, 0, in, out, N, lead, sink);
            if (mode == 2) hipLaunchKernelGGL(walk<2>, dim3(S / 64), dim3(128), 0, 0, in, out, N, lead, sink);
            if (mode == 3) hipLaunchKernelGGL(walk<3>, dim3(S / 64), dim3(128), 0, 0, in, out, N, lead, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("mode %d lead %d: %.3f ms, %.2f TB/s of the 25.2 GB a single read + write moves\n", mode, lead, best, S * N * 6.0 / best / 1e9);
    }
    return 0;
}
