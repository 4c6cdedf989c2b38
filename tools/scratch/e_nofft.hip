// scratch experiment: time mfcc_kernel variants (not part of the product)
#include "k_nofft.hip"
#include "../../rustpotter_amd/csrc/rp_host.h"
#include <cstdio>
#include <vector>
using namespace rp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    size_t S = argc > 1 ? atol(argv[1]) : 8192, N = 64000; int K = 5;
    HostTables h = build_tables(K);
    MfccTablesDev d; d.K1 = h.K1;
    CK(hipMalloc(&d.hamming, 480 * 4)); CK(hipMemcpy(d.hamming, h.hamming.data(), 480 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.tw240, 240 * 8)); CK(hipMemcpy(d.tw240, h.tw240.data(), 240 * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.tw480, 240 * 8)); CK(hipMemcpy(d.tw480, h.tw480.data(), 240 * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.fb, h.fb.size() * 4)); CK(hipMemcpy(d.fb, h.fb.data(), h.fb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.dct, h.dct.size() * 4)); CK(hipMemcpy(d.dct, h.dct.data(), h.dct.size() * 4, hipMemcpyHostToDevice));
    float *pcm, *out; size_t nf = 3 * (N / 480) - 3;
    CK(hipMalloc(&pcm, S * N * 4)); CK(hipMalloc(&out, S * nf * K * 4));
    CK(launch_synth(0, 0x5EED000000000001ULL, 0, S, N, N, pcm));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a, 0);
        CK(launch_mfcc(0, d, pcm, S, N, N, 0, nf, nf, out));
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("S=%zu mfcc %.3f ms  (%.2f Gframes/s)\n", S, ms, S * nf / ms / 1e6);
    }
    return 0;
}
