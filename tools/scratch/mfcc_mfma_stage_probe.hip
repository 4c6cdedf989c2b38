// mfcc_mfma_stage_probe.hip -- round 4, the "last structural attempt" on mfcc_kernel with its kill criterion:
// could the FFT16 stage (or the DFT15 stage) of the four-step FFT-240 run on the matrix cores as a product with the DFT matrix,
// operands as f16 two-way splits?  The matrix pipe only takes f16 / bf16 operands at a useful rate (the f32 matrix instructions run
// at the vector pipe's flop rate), so every input value has to be split on the VECTOR pipe first.  This file holds the two
// alternatives for ONE lane's share of a 4-frame tile (16 complex values per lane) as straight-line kernels; tools/isa_mix.py
// --whole prices their VALU issue cycles with the committed rate table (no GPU needed):
//   stage_valu: what mfcc_kernel does today -- window multiply + fft16() in registers (rp_device.h)
//   stage_mfma: window multiply + split of the 16 complex values into (x0, x1) f16 pairs + the matrix instructions that replace
//               fft16 (M = 32 rows of the real DFT-16 matrix, K = 32 = 16 n1 x re/im -> two 32x32x16 instructions per product,
//               x0 W0 + x1 W0 + x0 W1 -> 6 per 32 columns; a 4-frame tile is 60 columns -> 12 instructions per wave), results
//               back in the C/D layout.  The operand exchange that would bring (frame, n2) columns into B-operand lanes is NOT
//               included: this is a lower bound of the matrix form.
// Result (profiles/r04_mfcc_mfma_probe.txt): the split alone costs as many VALU issue cycles as the butterflies it would replace.
#include "../../rustpotter_amd/csrc/rp_device.h"

namespace rp {
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef float v16fp __attribute__((ext_vector_type(16)));
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stage_valu(const v2f *__restrict__ in, const v2f *__restrict__ ham, v2f *__restrict__ out) {
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i] * ham[i];
    fft16(v);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = v[i];
}

// the second stage as it is today: DFT15 in registers (for comparison: its split would cost the same as the first stage's)
__global__ __launch_bounds__(256) void stage_valu15(const v2f *__restrict__ in, v2f *__restrict__ out) {
    v2f u[15], z[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) u[i] = in[threadIdx.x * 15 + i];
    dft15(u, z);
#pragma unroll
    for (int i = 0; i < 15; ++i) out[threadIdx.x * 15 + i] = z[i];
}

__global__ __launch_bounds__(256) void stage_mfma(const v2f *__restrict__ in, const v2f *__restrict__ ham, const u32x4p *__restrict__ wmat,
                                                  v16fp *__restrict__ out) {
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i] * ham[i];
    // (re, im) of n1 -> k slots 2 n1, 2 n1 + 1: x0 = rtz_f16 (mask trick), x1 = rtn_f16(x - x0); 8 slots per operand register quad
    unsigned h0[16], h1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float p = v[i].x, q = v[i].y;
        h0[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p, q));
        h1[i] = pk_f16_second(p - __uint_as_float(__float_as_uint(p) & 0xffffe000u), q - __uint_as_float(__float_as_uint(q) & 0xffffe000u));
    }
    // this lane's k half of the B operand (a 32x32x16 instruction takes 8 k slots per lane): two k steps x (x0, x1)
    v16fp acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const f16x8p b0 = __builtin_bit_cast(f16x8p, (u32x4p){h0[8 * ks], h0[8 * ks + 1], h0[8 * ks + 2], h0[8 * ks + 3]});
        const f16x8p b1 = __builtin_bit_cast(f16x8p, (u32x4p){h1[8 * ks], h1[8 * ks + 1], h1[8 * ks + 2], h1[8 * ks + 3]});
        const f16x8p w0 = __builtin_bit_cast(f16x8p, wmat[(2 * ks) * 64 + (threadIdx.x & 63)]);
        const f16x8p w1 = __builtin_bit_cast(f16x8p, wmat[(2 * ks + 1) * 64 + (threadIdx.x & 63)]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b0, acc, 0, 0, 0);
    }
    // (the other half of the 16 values of this lane would feed a second column block: the split above already covers all 16)
    out[threadIdx.x] = acc;
    // keep the second half of the split alive
    unsigned keep = 0;
#pragma unroll
    for (int i = 4; i < 8; ++i) keep ^= h0[i] ^ h1[i] ^ h0[8 + i] ^ h1[8 + i];
    if (keep == 0x12345u) out[0][0] = 1.f;
}
}  // namespace rp
