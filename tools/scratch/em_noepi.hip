#include "km_noepi.hip"
#include "../../rustpotter_amd/csrc/rp_host.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace rp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    size_t B = 65536; int dims[4] = {3120, 32, 16, 2};
    std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 0.02f);
    std::vector<std::vector<float>> W(3), Bv(3);
    const float *wp[3], *bp[3];
    for (int l = 0; l < 3; ++l) { W[l].resize((size_t)dims[l] * dims[l + 1]); Bv[l].resize(dims[l + 1]); for (auto &v : W[l]) v = nd(g); for (auto &v : Bv[l]) v = nd(g); wp[l] = W[l].data(); bp[l] = Bv[l].data(); }
    Ctx *ctx = Ctx::create(0, 0); if (!ctx) return 1;
    Model *m = Model::create(ctx, 3, dims, wp, bp); if (!m) { printf("model fail\n"); return 1; }
    float *x, *out; CK(hipMalloc(&x, B * dims[0] * 4)); CK(hipMalloc(&out, B * 2 * 4)); CK(hipMemset(x, 0x3c, B * dims[0] * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int prec = 0; prec < 2; ++prec)
        for (int it = 0; it < 3; ++it) { hipEventRecord(a, 0); CK(launch_mlp_mfma(0, m->dev, x, B, prec, out)); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            if (it == 2) printf("prec %d: %.3f ms %.2f TB/s\n", prec, ms, B * dims[0] * 4.0 / ms / 1e9); }
    return 0;
}
