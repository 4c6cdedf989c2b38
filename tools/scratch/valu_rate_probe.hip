// valu_rate_probe.hip -- issue cost (cycles per wave-instruction and SIMD) of the VALU operations the DTW kernels are made of,
// at 1 / 2 / 4 waves per SIMD: 16 independent chains per lane, so neither latency nor dependent-issue stalls count.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/valu_rate_probe.hip -o /tmp/vrp && /tmp/vrp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define NCH 16
template <int OP>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
    float a[NCH], b[NCH];
    v2f p[NCH / 2];
    for (int i = 0; i < NCH; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f + i; }
    for (int i = 0; i < NCH / 2; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
    const float c0 = seed * 1.0001f, c1 = seed * 0.9999f;
    const v2f pc = (v2f){c0, c1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c0));
                if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));
                if (OP == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 7) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 8 && i < NCH / 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (OP == 9 && i < NCH / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pc));
                if (OP == 10 && i < NCH / 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (OP == 11) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
                if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 14) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c0));
                if (OP == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 16) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 17) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 18) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 20) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 21) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 23) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 25) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 26) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
                if (OP == 27) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (OP == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (OP == 29) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 30) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 31) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 33) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 34) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 35 && i < NCH / 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(b[i]), "v"(c0) : "vcc");
                if (OP == 36 && i < NCH / 2) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(p[i]) : "v"(p[(i + 1) % (NCH / 2)]));
                if (OP == 37) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 38) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 39 && i < NCH / 2) asm volatile("v_mov_b64 %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) % (NCH / 2)]));
                if (OP == 40) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (OP == 41) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 42) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                // round 5: the mixed-precision multiply-adds (an f16 source or an f16 result in one instruction) and the round-to-nearest pack
                if (OP == 43) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 44) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 45) asm volatile("v_fma_mixhi_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
                if (OP == 46) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 47) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (OP == 48) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += a[i];
    for (int i = 0; i < NCH / 2; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP>
static void run(const char *name, int per_iter, float *out) {
    const int iters = 4000;
    for (int wps = 2; wps <= 4; wps *= 2) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(k<OP>, dim3(1024 * wps), dim3(64), 0, 0, out, iters, 1.0f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        // per SIMD: wps waves x iters x per_iter instructions
        const double ns_per = best * 1e6 / ((double)wps * iters * per_iter);
        printf("%-20s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction and SIMD = %.2f cycles at 2.4 GHz\n", name, wps, best, ns_per, ns_per * 2.4);
    }
}
int main() {
    float *out; hipMalloc(&out, 4096 * 64 * 4);
    run<0>("v_add_f32", 64, out); run<1>("v_min3_f32", 64, out); run<2>("v_fma_f32", 64, out); run<3>("v_and_b32", 64, out);
    run<4>("v_perm_b32", 64, out); run<5>("v_cndmask_b32", 64, out); run<6>("v_cvt_pkrtz_f16_f32", 64, out); run<7>("v_min_f32", 64, out);
    run<8>("v_pk_add_f32", 32, out); run<9>("v_pk_fma_f32", 32, out); run<10>("v_pk_mul_f32", 32, out); run<11>("v_rsq_f32", 64, out);
    run<15>("v_min_u32", 64, out); run<16>("v_min3_u32", 64, out); run<17>("v_min_i32", 64, out); run<18>("v_min3_i32", 64, out);
    run<19>("v_max_f32", 64, out); run<20>("v_med3_f32", 64, out); run<21>("v_sub_f32", 64, out); run<22>("v_mul_f32", 64, out);
    run<23>("v_add3_u32", 64, out); run<24>("v_and_or_b32", 64, out); run<25>("v_bfi_b32", 64, out); run<26>("v_cvt_f16_f32", 64, out);
    run<27>("v_lshlrev_b32", 64, out); run<28>("v_cmp+v_cndmask (2)", 64, out); run<29>("v_max_u32", 64, out); run<30>("v_min_u16", 64, out);
    run<31>("v_pk_min_u16", 64, out); run<32>("v_add_u32", 64, out); run<33>("v_fmac_f32", 64, out); run<34>("v_sub_u32", 64, out);
    run<35>("v_mad_u64_u32", 32, out); run<36>("v_lshl_add_u64", 32, out); run<37>("v_add_f32_dpp", 64, out); run<38>("v_mov_b32_dpp", 64, out);
    run<39>("v_mov_b64", 32, out); run<40>("v_log_f32", 64, out); run<41>("v_mul_lo_u32", 64, out); run<42>("v_mad_u32_u24", 64, out);
    run<43>("v_fma_mix_f32", 64, out); run<44>("v_fma_mixlo_f16", 64, out); run<45>("v_fma_mixhi_f16", 64, out); run<46>("v_cvt_pk_f16_f32", 64, out);
    run<47>("v_pk_mul_f16", 64, out); run<48>("v_max3_f32", 64, out);
    return 0;
}
