// probe: VOP3P packed-f32 modifiers (op_sel / op_sel_hi / neg_lo / neg_hi) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f add_rot(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f sub_rot(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f odd_part(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f cmul_asm(v2f w, v2f b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(w), "v"(b), "v"(r));
    return r;
}
__global__ void k(const v2f *a, const v2f *b, v2f *o) {
    int i = threadIdx.x;
    o[i * 5 + 0] = add_rot(a[i], b[i]);
    o[i * 5 + 1] = sub_rot(a[i], b[i]);
    o[i * 5 + 2] = add_conj(a[i], b[i]);
    o[i * 5 + 3] = odd_part(a[i], b[i]);
    o[i * 5 + 4] = cmul_asm(a[i], b[i]);
}
int main() {
    v2f ha[64], hb[64], ho[320], *da, *db, *dout;
    for (int i = 0; i < 64; ++i) { ha[i] = (v2f){1.f + i, 0.5f - i}; hb[i] = (v2f){0.25f * i + 3.f, -7.f + 0.125f * i}; }
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        v2f a = ha[i], b = hb[i];
        v2f e[5] = {{a.x + b.y, a.y - b.x}, {a.x - b.y, a.y + b.x}, {a.x + b.x, a.y - b.y}, {a.y + b.y, b.x - a.x},
                    {__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y)}};
        for (int j = 0; j < 5; ++j)
            if (ho[i * 5 + j].x != e[j].x || ho[i * 5 + j].y != e[j].y) { if (bad < 10) printf("mismatch op %d lane %d: got (%g,%g) want (%g,%g)\n", j, i, ho[i*5+j].x, ho[i*5+j].y, e[j].x, e[j].y); ++bad; }
    }
    printf("pk probe: %d mismatches\n", bad);
    return bad != 0;
}
