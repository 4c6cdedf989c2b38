// vgpr_bank_probe.hip -- does the issue rate of a three-source vector instruction depend on WHICH registers it reads?  v_min3_f32 / v_fma_f32 /
// v_add_f32 with hand-picked physical registers, four independent destination chains each, 4 waves per SIMD (issue-bound).
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/vgpr_bank_probe.hip -o /tmp/vbp && /tmp/vbp
#include <hip/hip_runtime.h>
#include <stdio.h>

// one iteration = 64 instructions; D0..D3 destinations (also first source), S1 / S2 the other sources
#define BODY4(OP, D0, D1, D2, D3, S1, S2)          \
    OP " " D0 ", " D0 ", " S1 ", " S2 "\n"         \
    OP " " D1 ", " D1 ", " S1 ", " S2 "\n"         \
    OP " " D2 ", " D2 ", " S1 ", " S2 "\n"         \
    OP " " D3 ", " D3 ", " S1 ", " S2 "\n"
#define BODY16(OP, D0, D1, D2, D3, S1, S2) BODY4(OP, D0, D1, D2, D3, S1, S2) BODY4(OP, D0, D1, D2, D3, S1, S2) BODY4(OP, D0, D1, D2, D3, S1, S2) BODY4(OP, D0, D1, D2, D3, S1, S2)
#define BODY64(OP, D0, D1, D2, D3, S1, S2) BODY16(OP, D0, D1, D2, D3, S1, S2) BODY16(OP, D0, D1, D2, D3, S1, S2) BODY16(OP, D0, D1, D2, D3, S1, S2) BODY16(OP, D0, D1, D2, D3, S1, S2)

#define KERNEL(NAME, OP, D0, D1, D2, D3, S1, S2)                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed) {                  \
        float r = seed;                                                                                \
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"  \
                     "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"  \
                     "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"  \
                     "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n"  \
                     :: "v"(r) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"); \
        for (int it = 0; it < iters; ++it)                                                             \
            asm volatile(BODY64(OP, D0, D1, D2, D3, S1, S2)                                             \
                         ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"); \
        asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v28\n v_add_f32 %0, %0, v32" : "=v"(r) :: "v20"); \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                       \
    }

// destinations v20..v23 = banks 0..3 (if bank = index mod 4)
KERNEL(min3_spread, "v_min3_f32", "v20", "v21", "v22", "v23", "v25", "v30")   // sources in banks 1, 2: dest bank varies 0..3 -> some same-bank
KERNEL(min3_d0, "v_min3_f32", "v20", "v24", "v28", "v32", "v25", "v30")       // dests all bank 0, sources banks 1, 2: three different banks
KERNEL(min3_same2, "v_min3_f32", "v20", "v24", "v28", "v32", "v25", "v29")    // sources both bank 1
KERNEL(min3_same3, "v_min3_f32", "v20", "v24", "v28", "v32", "v36", "v40")    // all three bank 0 (v36, v40 unwritten: values irrelevant)
KERNEL(fma_d0, "v_fma_f32", "v20", "v24", "v28", "v32", "v25", "v30")
KERNEL(fma_same3, "v_fma_f32", "v20", "v24", "v28", "v32", "v36", "v40")

template <typename F>
static double run(F kern, const char *name) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * 4, iters = 4000;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    float *out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 200, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    const double cyc = ms * 1e-3 * 2.4e9 / iters / 64.0 / 4.0;   // nominal SIMD cycles per wave-instruction
    printf("%-44s %.2f nominal cycles per instruction\n", name, cyc);
    return cyc;
}

int main() {
    run(min3_d0, "v_min3_f32, sources in three different banks");
    run(min3_spread, "v_min3_f32, destination bank 0..3 in turn");
    run(min3_same2, "v_min3_f32, two sources in one bank");
    run(min3_same3, "v_min3_f32, all three in one bank");
    run(fma_d0, "v_fma_f32, three different banks");
    run(fma_same3, "v_fma_f32, all three in one bank");
    return 0;
}
