// Probe (round 4): how fast can a workgroup-owns-R-rows walk move i16 in -> f32 out as a function of the row segment it touches per
// step?  R rows x TS samples per step (TS * 2 bytes in, TS * 4 bytes out per row), S = 65536 rows of N = 64000 samples, one workgroup
// of 128 threads per R rows.  No LDS, no arithmetic but the conversion: the ceiling of the access pattern itself.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/scratch/rowprobe tools/scratch/row_tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int R, int TS>
__global__ __launch_bounds__(128) void walk(const short *__restrict__ in, float *__restrict__ out, size_t N) {
    const int t = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * R;
    constexpr int PPR = TS / 8;            // 16-byte input pieces per row and step
    constexpr int PIECES = R * PPR;        // per step
    constexpr int PER = PIECES / 128;      // per thread
    static_assert(PIECES % 128 == 0, "shape");
    for (size_t k = 0; k + TS <= N; k += TS) {
        v4i cur[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = i * 128 + t;
            cur[i] = *reinterpret_cast<const v4i *>(in + (row0 + p / PPR) * N + k + (p % PPR) * 8);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = i * 128 + t;
            const v4i c = cur[i];
            float *o = out + (row0 + p / PPR) * N + k + (p % PPR) * 8;
            const v4f lo = {(float)(short)(c.x & 0xffff), (float)(c.x >> 16), (float)(short)(c.y & 0xffff), (float)(c.y >> 16)};
            const v4f hi = {(float)(short)(c.z & 0xffff), (float)(c.z >> 16), (float)(short)(c.w & 0xffff), (float)(c.w >> 16)};
            *reinterpret_cast<v4f *>(o) = lo;
            *reinterpret_cast<v4f *>(o + 4) = hi;
        }
    }
}
template <int R, int TS> void run(const short *in, float *out, size_t S, size_t N) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((walk<R, TS>), dim3(S / R), dim3(128), 0, 0, in, out, N);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%3d rows x %4d samples per step (%4d B in, %4d B out per row; %d workgroups): %.3f ms, %.2f TB/s\n", R, TS, TS * 2, TS * 4, (int)(S / R), best,
           S * N * 6.0 / best / 1e9);
}
int main() {
    const size_t S = 65536, N = 64000;
    short *in; float *out;
    (void)hipMalloc(&in, S * N * 2); (void)hipMalloc(&out, S * N * 4);
    (void)hipMemset(in, 1, S * N * 2);
    run<64, 64>(in, out, S, N);
    run<64, 128>(in, out, S, N);
    run<32, 128>(in, out, S, N);
    run<32, 256>(in, out, S, N);
    run<16, 256>(in, out, S, N);
    run<16, 512>(in, out, S, N);
    run<8, 512>(in, out, S, N);
    run<8, 1024>(in, out, S, N);
    run<4, 1024>(in, out, S, N);
    run<64, 64>(in, out, S, N);
    return 0;
}
