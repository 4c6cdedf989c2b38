// dtw_cost_probe.hip -- issue cost of one template-pair ROW of the DTW register kernel in two formulations:
//   A (shipped):  10 band cells x 5 v_pk_fma_f32 (coefficient pair x broadcast window component), then per cell 2 v_min3 + 1 v_pk_add
//   B (DPP sharing of the template-row . frame products): 5 v_pk_fma for G (this lane's frame), 5 for a.mu, then per cell
//      two v_mov_b32_dpp wave_shl:1 (the pair's G of the next frame arrives from the next lane), v_pk_add (a.mu - G),
//      v_pk_fma (x inv, + 1), and the same 2 v_min3 + 1 v_pk_add.
// Both run the same serial min chain; registers are kept live so that nothing is folded.  Two waves per SIMD like the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/dtw_cost_probe.hip -o tools/scratch/dtw_cost_probe && tools/scratch/dtw_cost_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(64, 2) void probe(const float *__restrict__ in, float *__restrict__ out, int rows, long long *cycles) {
    const int lane = threadIdx.x;
    v2f y[10][5], a[5], P[11], mu[5];
    float inv[10];
    for (int q = 0; q < 10; ++q) { inv[q] = in[lane + q]; for (int k = 0; k < 5; ++k) y[q][k] = (v2f){in[lane + q + k], in[lane + 2 * q + k]}; }
    for (int k = 0; k < 5; ++k) { a[k] = (v2f){in[k], in[k + 5]}; mu[k] = (v2f){in[lane + k], in[lane + k]}; }
    for (int q = 0; q <= 10; ++q) P[q] = (v2f){in[q], in[q + 1]};
    const long long t0 = clock64();
    for (int r = 0; r < rows; ++r) {
        v2f d[10];
        if (VARIANT == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) d[q] = (v2f){1.f, 1.f};
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int q = 0; q < 10; ++q) d[q] = __builtin_elementwise_fma(-a[k], y[q][k].xx, d[q]);
        } else {
            v2f g = (v2f){0.f, 0.f}, am = (v2f){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 5; ++k) { g = __builtin_elementwise_fma(a[k], y[0][k].xx, g); am = __builtin_elementwise_fma(a[k], mu[k], am); }
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                if (q > 0) {  // the products of the next frame: one lane to the right
                    g.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g.x), 0x130, 0xf, 0xf, false));  // wave_shl:1
                    g.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g.y), 0x130, 0xf, 0xf, false));
                }
                const v2f t = am - g;
                d[q] = __builtin_elementwise_fma(t, (v2f){inv[q], inv[q]}, (v2f){1.f, 1.f});
            }
        }
        v2f left = (v2f){__builtin_inff(), __builtin_inff()};
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            v2f m;
            m.x = fminf(fminf(P[q + 1].x, left.x), P[q].x);
            m.y = fminf(fminf(P[q + 1].y, left.y), P[q].y);
            const v2f v = d[q] + m;
            P[q] = v;
            left = v;
        }
        a[r % 5] += (v2f){1e-9f, 1e-9f};  // a row-dependent coefficient: nothing is loop invariant
    }
    const long long t1 = clock64();
    float acc = 0.f;
    for (int q = 0; q < 10; ++q) acc += P[q].x + P[q].y;
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    float *in, *out; long long *cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&cyc, 8);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (float)(i % 97) + 0.1f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int rows = 60000;
    for (int rep = 0; rep < 6; ++rep) for (int v = 0; v < 2; ++v) {  // interleaved: the clock settles over the first launches
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(2048), dim3(64), 0, 0, in, out, rows, cyc);
        else hipLaunchKernelGGL(probe<1>, dim3(2048), dim3(64), 0, 0, in, out, rows, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // 2048 waves = 2 per SIMD on 1024 SIMDs, all resident: wall time / rows = time of one pair-row for two waves sharing a SIMD
        printf("variant %c: %.3f ms, %.1f ns per pair-row per wave-pair, wave 0: %.1f clock64 ticks per row\n", v ? 'B' : 'A', ms, ms * 1e6 / rows, (double)c / rows);
    }
    return 0;
}
