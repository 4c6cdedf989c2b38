// rp_device.h -- device-side helpers shared by the gfx950 kernels (rp_*.hip): packed-f32 complex arithmetic, the
// small DFTs the FFT-240 is built from, wave-scope LDS ordering, sample decoding.  Everything is compiled with
// -ffp-contract=off: every fused multiply-add is an explicit fmaf() / __builtin_elementwise_fma (the reference never
// contracts, and pre-emphasis and the DCT must round exactly like it so that digital silence still normalises to
// exactly zero, SURVEY.md §7 "Silence").
#pragma once
#include "rp_kernels.h"

#include <float.h>
#include <math.h>

namespace rp {

#define RP_INF __builtin_inff()

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ complex helpers
// Complex numbers are 2-lane ext vectors so that add/sub/mul map onto the packed f32 VALU ops
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: 2 results per 4-cycle slot against 3 cycles for
// one plain op on gfx950).
typedef float v2f __attribute__((ext_vector_type(2)));
// The half swaps and one-sided sign flips of complex arithmetic are operand modifiers of the packed instructions
// (op_sel / op_sel_hi pick the 32-bit half each result lane reads, neg_lo / neg_hi negate a source for one lane); the
// compiler only folds whole-vector negations, so these are spelled out -- same operations, same bits, no v_xor / v_mov
// (tools/scratch/pk_probe.hip checks the semantics on the device).
// a * b = (a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x) as r = a.xx * b; r = fma((-a.y, a.y), b.yx, r)
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(r));
    return r;
}
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }  // a * (-i)
// a + (-i) b = (a.x + b.y, a.y - b.x) and a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_mi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b) = (a.x + b.x, a.y - b.y) and -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f mi_sub_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the real parts / the imaginary parts of e + t and e - t side by side: (e.x + t.x, e.x - t.x) and (e.y + t.y, e.y - t.y)
__device__ __forceinline__ v2f re_sum_diff(v2f e, v2f t) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(e), "v"(t));
    return r;
}
__device__ __forceinline__ v2f im_sum_diff(v2f e, v2f t) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(e), "v"(t));
    return r;
}

// forward 4-point DFT, in place, natural order
__device__ __forceinline__ void dft4(v2f &x0, v2f &x1, v2f &x2, v2f &x3) {
    v2f t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, d = x1 - x3;
    x0 = t0 + t2; x1 = add_mi(t1, d); x2 = t0 - t2; x3 = sub_mi(t1, d);
}

// forward 16-point DFT in registers.  Input v[n]; output X[c + 4d] is left at v[4c + d].
__device__ __forceinline__ void fft16(v2f (&v)[16]) {
    // W16^e = exp(-2*pi*i*e/16) for e = b*c, b,c in 0..3
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);  // -> v[4c+b]
    // twiddles v[4c+b] *= W16^{bc}
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], (v2f){C1, -S1});   // e=1
    v[4 * 1 + 2] = cmul(v[4 * 1 + 2], (v2f){R2, -R2});   // e=2
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], (v2f){S1, -C1});   // e=3
    v[4 * 2 + 1] = cmul(v[4 * 2 + 1], (v2f){R2, -R2});   // e=2
    v[4 * 2 + 2] = mul_mi(v[4 * 2 + 2]);                 // e=4
    v[4 * 2 + 3] = cmul(v[4 * 2 + 3], (v2f){-R2, -R2});  // e=6
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], (v2f){S1, -C1});   // e=3
    v[4 * 3 + 2] = cmul(v[4 * 3 + 2], (v2f){-R2, -R2});  // e=6
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], (v2f){-C1, S1});   // e=9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

__device__ __forceinline__ void dft3(v2f &x0, v2f &x1, v2f &x2) {
    constexpr float C = 0.86602540378443865f;
    v2f s = x1 + x2, d = x1 - x2;
    v2f m = __builtin_elementwise_fma((v2f){-0.5f, -0.5f}, s, x0);
    v2f r = (v2f){C, -C} * d.yx;  // -i * C * d
    x0 = x0 + s;
    x1 = m + r;
    x2 = m - r;
}

__device__ __forceinline__ void dft5(v2f &x0, v2f &x1, v2f &x2, v2f &x3, v2f &x4) {
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    v2f a1 = x1 + x4, a2 = x2 + x3, b1 = x1 - x4, b2 = x2 - x3;
    v2f m1 = __builtin_elementwise_fma((v2f){c2, c2}, a2, __builtin_elementwise_fma((v2f){c1, c1}, a1, x0));
    v2f m2 = __builtin_elementwise_fma((v2f){c1, c1}, a2, __builtin_elementwise_fma((v2f){c2, c2}, a1, x0));
    v2f n1 = __builtin_elementwise_fma((v2f){s2, s2}, b2, (v2f){s1, s1} * b1);
    v2f n2 = __builtin_elementwise_fma((v2f){-s1, -s1}, b2, (v2f){s2, s2} * b1);
    x0 = x0 + (a1 + a2);
    x1 = add_mi(m1, n1);  // m1 + (-i) n1
    x4 = sub_mi(m1, n1);
    x2 = add_mi(m2, n2);
    x3 = sub_mi(m2, n2);
}

// forward 15-point DFT (Good-Thomas 3x5, no twiddles): z[k] = sum_n u[n] W15^{nk}
__device__ __forceinline__ void dft15(const v2f (&u)[15], v2f (&z)[15]) {
    v2f y[3][5];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        v2f a0 = u[(3 * n2) % 15], a1 = u[(5 + 3 * n2) % 15], a2 = u[(10 + 3 * n2) % 15];
        dft3(a0, a1, a2);
        y[0][n2] = a0; y[1][n2] = a1; y[2][n2] = a2;
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        dft5(y[k1][0], y[k1][1], y[k1][2], y[k1][3], y[k1][4]);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) z[(10 * k1 + 6 * k2) % 15] = y[k1][k2];
    }
}

// orders this wave's LDS traffic (lanes exchange data through LDS without a workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 16 lanes of a DPP row; every lane ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    return v;
}

// Input sample types of the reference's `Sample` trait (src/audio/audio_types.rs:98-137): integers are
// converted as `v as f32 / T::MAX as f32` (an IEEE division, not a multiply by the reciprocal).
// v / d for the small integers v of an i8 / i16 sample, correctly rounded like the IEEE division the reference does
// (`v as f32 / T::MAX as f32`, src/audio/encoder.rs), in three instructions instead of the ~12 of a full f32 divide:
// q0 = v * RN(1/d), the exact remainder e = v - q0 * d by fma, q = fma(e, RN(1/d), q0) (Markstein).  Equal to the division for
// every i8 / i16 value (checked exhaustively: tests/test_gpu_parity.py::test_sample_decode_is_exact_for_every_value).
template <int D> __device__ __forceinline__ float div_small_int(float v) {
    constexpr float d = (float)D, r = 1.0f / d;
    const float q0 = v * r;
    return fmaf(fmaf(-q0, d, v), r, q0);
}

template <class T> struct SampleIn;  // Raw4 / ldraw / cvt4: four samples fetched now, converted when they are used
template <> struct SampleIn<float> {
    using Raw4 = float4;
    static __device__ __forceinline__ float cvt(float v) { return v; }
    static __device__ __forceinline__ Raw4 ldraw(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 s) { return s; }
    static __device__ __forceinline__ float4 load4(const float *p) { return ldraw(p); }
};
template <> struct SampleIn<int16_t> {
    using Raw4 = short4;
    static __device__ __forceinline__ float cvt(int16_t v) { return div_small_int<32767>((float)v); }
    static __device__ __forceinline__ Raw4 ldraw(const int16_t *p) { return *reinterpret_cast<const short4 *>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 s) { return make_float4(cvt(s.x), cvt(s.y), cvt(s.z), cvt(s.w)); }
    static __device__ __forceinline__ float4 load4(const int16_t *p) { return cvt4(ldraw(p)); }
};
template <> struct SampleIn<int8_t> {
    using Raw4 = char4;
    static __device__ __forceinline__ float cvt(int8_t v) { return div_small_int<127>((float)v); }
    static __device__ __forceinline__ Raw4 ldraw(const int8_t *p) { return *reinterpret_cast<const char4 *>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 s) {
        return make_float4(cvt((int8_t)s.x), cvt((int8_t)s.y), cvt((int8_t)s.z), cvt((int8_t)s.w));
    }
    static __device__ __forceinline__ float4 load4(const int8_t *p) { return cvt4(ldraw(p)); }
};
template <> struct SampleIn<int32_t> {
    using Raw4 = int4;
    static __device__ __forceinline__ float cvt(int32_t v) { return (float)v / 2147483648.f; }  // i32::MAX as f32 == 2^31
    static __device__ __forceinline__ Raw4 ldraw(const int32_t *p) { return *reinterpret_cast<const int4 *>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 s) { return make_float4(cvt(s.x), cvt(s.y), cvt(s.z), cvt(s.w)); }
    static __device__ __forceinline__ float4 load4(const int32_t *p) { return cvt4(ldraw(p)); }
};

// The SECOND part of an f16 two-way split on the window side, x1 = f16(x - x0) (x0 = x truncated to 11 significant bits): truncated
// like x0 (v_cvt_pkrtz_f16_f32).  Rounding it to nearest (v_cvt_pk_f16_f32, RP_SPLIT_RTN builds) was measured in round 4: the instruction
// issues at less than half pkrtz's rate (dtw_mfma_kernel 11.70 -> 12.17 ms at C3) for a third less error.  The truncation makes every
// product (x0 + x1) a fall short by 2^-22.4 of itself on average; the TEMPLATE side, split on the host, takes that out (kDtwSplitGain,
// rp_ctx.cpp) and rounds both of its parts to nearest, which also makes the dropped x1 a1 term zero-mean.
__device__ __forceinline__ unsigned pk_f16_second(float lo, float hi) {
#ifndef RP_SPLIT_RTN
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
#else
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    typedef float f2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2_){lo, hi}, h2_));
#endif
}

// A row of the wakeword-model forward with a feature beyond the f16 range: listed for the f32 pass (rp_kernels.h kMlpF16x2; redo[0] = rows
// listed, entries from redo[2] on, room for `cap` = every row of the call).  A row is listed at most once per call and the words are zero
// between calls (the second pass, or the launcher's mlp_redo_abort on an error in between, puts them back), so the index stays below cap;
// the clamp keeps a count left behind by anything else from writing past the buffer.
__device__ __forceinline__ void mlp_redo_append(uint32_t *redo, uint32_t row, size_t cap) {
    const uint32_t i = atomicAdd(redo, 1u);
    if ((size_t)i < cap) redo[2 + i] = row;
}

// A (window, chunk or template) pair whose frames left the norm range of the scale-invariant cosine (rp_kernels.h, DtwWork):
// appended to the call's list for dtw_ref_kernel.  spec = chunk index, kFixSpecTemplate | template index.
__device__ __forceinline__ void dtw_fix_append(uint32_t *fix, size_t row, uint32_t spec) {
    const uint32_t i = atomicAdd(fix, 1u);
    if (i < kDtwFixCap) reinterpret_cast<unsigned long long *>(fix + 2)[i] = ((unsigned long long)row << 24) | spec;
}

}  // namespace rp
