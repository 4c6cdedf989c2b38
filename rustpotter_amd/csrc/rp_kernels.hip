// rp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the rustpotter MFCC + DTW
// scoring path.  Written for 64-wide wavefronts; compiled with -ffp-contract=off so
// that every fused multiply-add below is an explicit fmaf(): the reference never
// contracts (Rust), and two places (pre-emphasis, DCT) must round exactly like it so
// that digital silence still normalises to exactly zero (SURVEY.md §7 "Silence").
//
// Kernels (DESIGN.md §3 has the data layout and the roofline of each):
//   mfcc_kernel        src/mfcc/extractor.rs:60-163   16 lanes per 30 ms frame
//   dtw_band_kernel    src/mfcc/dtw.rs:56-105 + comparator.rs + normalizer.rs,
//                      one lane per (window, template), band + ring in registers
//   dtw_generic_kernel same, any K / band / m!=n, band state in LDS (fallback)
//   aggregate_kernel   src/wakewords/comp/wakeword_comp.rs:38-49,108-139
//   scan_kernel        src/detector.rs:290-302,377-454 (no VAD)
//   mlp_layer_kernel   src/wakewords/nn/wakeword_nn.rs:305-389
//   synth_kernel       BASELINE.md §2 input generator
#include "rp_kernels.h"

#include <float.h>
#include <math.h>

namespace rp {

#define RP_INF __builtin_inff()

// ------------------------------------------------------------------ complex helpers
// Complex numbers are 2-lane ext vectors so that add/sub/mul map onto the packed f32 VALU ops
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: 2 results per 4-cycle slot against 3 cycles for
// one plain op on gfx950).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
    v2f r = a.xx * b;
    return __builtin_elementwise_fma((v2f){-a.y, a.y}, b.yx, r);
}
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }  // a * (-i)

// forward 4-point DFT, in place, natural order
__device__ __forceinline__ void dft4(v2f &x0, v2f &x1, v2f &x2, v2f &x3) {
    v2f t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, t3 = mul_mi(x1 - x3);
    x0 = t0 + t2; x1 = t1 + t3; x2 = t0 - t2; x3 = t1 - t3;
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
    v2f r1 = mul_mi(n1), r2 = mul_mi(n2);  // -i * n
    x0 = x0 + (a1 + a2);
    x1 = m1 + r1;
    x4 = m1 - r1;
    x2 = m2 + r2;
    x3 = m2 - r2;
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

// ------------------------------------------------------------------------- MFCC
// One wave = 4 consecutive frames of one stream, 16 lanes per frame; waves are independent
// (no workgroup barrier after the one-time table staging) and walk the (stream, frame-tile)
// space grid-stride, so the constant tables are staged into LDS once per workgroup, not once
// per tile.  Frame j covers samples [(j+1)*160, (j+4)*160) of the stream (the frame made of
// the first three shifts is never emitted, src/mfcc/extractor.rs:69-79).  The real 480-point
// DFT is one 240-point complex FFT (16 x 15 four-step: FFT16 per lane over n1, twiddle,
// transpose through LDS, DFT15 per lane over n2) plus the even/odd untangling step; only bins
// 0..239 are formed (src/mfcc/extractor.rs:28,111-113).
constexpr int kMfccFramesPerWave = 4;
constexpr int kMfccWaves = 4;
constexpr int kMfccThreads = 64 * kMfccWaves;
constexpr int kMfccStage = (kMfccFramesPerWave + 2) * kShift;  // 960 samples per wave tile
constexpr int kMfccWaveScratch = kMfccFramesPerWave * 240;      // float2 per wave (aliases the samples)

__host__ __device__ inline size_t mfcc_lds_bytes(int K1) {
    size_t f = 480 + (size_t)K1 * kBins + (size_t)K1 * K1 + (size_t)kMfccWaves * kMfccFramesPerWave * K1;
    size_t c = (size_t)kMfccWaves * kMfccWaveScratch + 240 + 240;
    return c * sizeof(float2) + f * sizeof(float);
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
template <class T> struct SampleIn;
template <> struct SampleIn<float> {
    static __device__ __forceinline__ float cvt(float v) { return v; }
    static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
};
template <> struct SampleIn<int16_t> {
    static __device__ __forceinline__ float cvt(int16_t v) { return (float)v / 32767.f; }
    static __device__ __forceinline__ float4 load4(const int16_t *p) {
        const short4 s = *reinterpret_cast<const short4 *>(p);
        return make_float4(cvt(s.x), cvt(s.y), cvt(s.z), cvt(s.w));
    }
};
template <> struct SampleIn<int8_t> {
    static __device__ __forceinline__ float cvt(int8_t v) { return (float)v / 127.f; }
    static __device__ __forceinline__ float4 load4(const int8_t *p) {
        const char4 s = *reinterpret_cast<const char4 *>(p);
        return make_float4(cvt((int8_t)s.x), cvt((int8_t)s.y), cvt((int8_t)s.z), cvt((int8_t)s.w));
    }
};
template <> struct SampleIn<int32_t> {
    static __device__ __forceinline__ float cvt(int32_t v) { return (float)v / 2147483648.f; }  // i32::MAX as f32 == 2^31
    static __device__ __forceinline__ float4 load4(const int32_t *p) {
        const int4 s = *reinterpret_cast<const int4 *>(p);
        return make_float4(cvt(s.x), cvt(s.y), cvt(s.z), cvt(s.w));
    }
};

// K1T: compile-time K+1 (6 and 17 are instantiated), 0 = runtime value.  TIN: input sample type.
template <bool VEC4, int K1T, class TIN>
__global__ __launch_bounds__(kMfccThreads, 3) void mfcc_kernel(
    const TIN *__restrict__ pcm, size_t n_samples, size_t pcm_stride, unsigned tiles_per_stream, size_t total_tiles,
    size_t first_frame, size_t n_frames, size_t out_frame_pitch, int K1rt, const float *__restrict__ g_ham,
    const float2 *__restrict__ g_tw240, const float2 *__restrict__ g_tw480, const float *__restrict__ g_fb,
    const float *__restrict__ g_dct, float *__restrict__ out) {
    const int K1 = K1T > 0 ? K1T : K1rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f *scr_all = reinterpret_cast<v2f *>(smem);               // [waves][4][240]
    v2f *tw240 = scr_all + kMfccWaves * kMfccWaveScratch;       // [240]
    v2f *tw480 = tw240 + 240;                                   // [240]
    float *ham = reinterpret_cast<float *>(tw480 + 240);        // [480]
    float *fb = ham + 480;                                      // [K1][240]
    float *dct = fb + K1 * kBins;                               // [K1][K1]
    float *lgb_all = dct + K1 * K1;                             // [waves][4][K1]

    const int tid = threadIdx.x;
    for (int i = tid; i < 480; i += kMfccThreads) ham[i] = g_ham[i];
    for (int i = tid; i < 240; i += kMfccThreads) {
        tw240[i] = (v2f){g_tw240[i].x, g_tw240[i].y};
        tw480[i] = (v2f){g_tw480[i].x, g_tw480[i].y};
    }
    for (int i = tid; i < K1 * kBins; i += kMfccThreads) fb[i] = g_fb[i];
    for (int i = tid; i < K1 * K1; i += kMfccThreads) dct[i] = g_dct[i];
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int grp = lane >> 4, l = lane & 15;
    v2f *scr = scr_all + wave * kMfccWaveScratch;
    float *ypre = reinterpret_cast<float *>(scr);  // [960] pre-emphasised samples, dead before scr is written
    v2f *my = scr + grp * 240;
    float *lgb = lgb_all + (wave * kMfccFramesPerWave + grp) * K1;
    const int K = K1 - 1;
    // per-lane base pointers: every LDS access below is base[compile-time offset]
    const int l15 = l < 15 ? l : 0;
    const v2f *ysrc = reinterpret_cast<const v2f *>(ypre + grp * kShift) + l15;  // z[15*n1 + n2] -> +15*n1
    const v2f *hsrc = reinterpret_cast<const v2f *>(ham) + l15;
    v2f *t1dst = my + l;              // [k1*15 + l]
    const v2f *t1src = my + l * 15;   // [l*15 + n2]
    v2f *zdst = my + l;               // [l + 16*k2]
    const v2f *zmir = my + 240 - l;  // Z[240-k] = zmir[-16*k2]; k == 0 pairs with itself (handled below)
    const v2f *w480 = tw480 + l;
    const float *fbk = fb + l;                       // bins k = l + 16*k2
    const float *fbm = fb + 240 - l;                 // bins 240-k = fbm[-16*k2] (k == 0: weight unused, power forced to 0)
    // twiddles W240^{n2*k1}, fixed per lane across tiles, k1 = c + 4d kept at register 4c+d
    v2f twl[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) twl[4 * c + d] = tw240[l15 * (c + 4 * d)];

    const size_t wave_stride = (size_t)gridDim.x * kMfccWaves;
    for (size_t wt = (size_t)blockIdx.x * kMfccWaves + wave; wt < total_tiles; wt += wave_stride) {
        const size_t s = wt / tiles_per_stream;
        const size_t j0 = first_frame + (wt - s * tiles_per_stream) * kMfccFramesPerWave;
        const TIN *x = pcm + s * pcm_stride;
        // pre_emphasis, src/mfcc/extractor.rs:87-97: previous sample is 0 at the start of EVERY shift.
        // All loads are issued unconditionally (clamped index) before the first use, so the wave
        // pays one memory round trip per tile instead of one per load.
        const size_t base = (j0 + 1) * kShift;
        const size_t last = n_samples - 1;
        if (VEC4) {
            float4 cur[4];
            float prv[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + lane;            // float4 index inside the 960-sample tile
                size_t g = base + 4 * (size_t)(q < 240 ? q : 239);
                g = g + 3 <= last ? g : (last - 3) & ~(size_t)3;
                cur[it] = SampleIn<TIN>::load4(x + g);
                prv[it] = SampleIn<TIN>::cvt(x[g - 1]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + lane;
                const float p0 = (q % (kShift / 4) == 0) ? 0.f : prv[it];
                float4 y;
                y.x = cur[it].x - 0.97f * p0;  // separate multiply and subtract, like the reference
                y.y = cur[it].y - 0.97f * cur[it].x;
                y.z = cur[it].z - 0.97f * cur[it].y;
                y.w = cur[it].w - 0.97f * cur[it].z;
                if (q < 240) *reinterpret_cast<float4 *>(ypre + 4 * q) = y;
            }
        } else {
            float cur[kMfccStage / 64], prv[kMfccStage / 64];
#pragma unroll
            for (int it = 0; it < kMfccStage / 64; ++it) {
                size_t g = base + it * 64 + lane;
                g = g <= last ? g : last;
                cur[it] = SampleIn<TIN>::cvt(x[g]);
                prv[it] = SampleIn<TIN>::cvt(x[g - 1]);
            }
#pragma unroll
            for (int it = 0; it < kMfccStage / 64; ++it) {
                const int i = it * 64 + lane;
                ypre[i] = cur[it] - 0.97f * ((i % kShift == 0) ? 0.f : prv[it]);
            }
        }
        wave_lds_sync();
        // ---- step 1: lane n2=l (<15): FFT16 over n1 of z[15*n1 + n2], z[n] = (y[2n], y[2n+1]) * hamming
        v2f v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = ysrc[15 * n1] * hsrc[15 * n1];
        wave_lds_sync();  // every lane has its samples in registers: the scratch may now overwrite them
        fft16(v);
        if (l < 15) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) t1dst[(c + 4 * d) * 15] = cmul(v[4 * c + d], twl[4 * c + d]);  // W240^{n2*k1}
        }
        wave_lds_sync();
        // ---- step 3: lane k1=l: DFT15 over n2 -> Z[k1 + 16*k2]
        v2f z[15];
        {
            v2f u[15];
#pragma unroll
            for (int n2 = 0; n2 < 15; ++n2) u[n2] = t1src[n2];
            dft15(u, z);
        }
        wave_lds_sync();
#pragma unroll
        for (int k2 = 0; k2 < 15; ++k2) zdst[16 * k2] = z[k2];
        wave_lds_sync();
        // ---- untangle the two interleaved real sequences, bins k = l + 16*k2 <= 120 together with their
        // mirrors 240-k (X[240-k] = conj(E - W480^k O) shares E, O and the twiddle product with X[k]).
        // Everything is kept at twice its value; the factor 4 on the powers is removed before the log.
        float Pk[8], Pm[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const int k = l + 16 * k2;
            const v2f a = z[k2];
            const v2f b = (k2 == 0 && l == 0) ? a : zmir[-16 * k2];
            const v2f e = (v2f){a.x + b.x, a.y - b.y};           // 2E = a + conj(b)
            const v2f o = (v2f){a.y + b.y, b.x - a.x};           // 2O = -i (a - conj(b))
            const v2f t = cmul(w480[16 * k2], o);
            const v2f xp = e + t, xm = e - t;
            const float pk = fmaf(xp.x, xp.x, xp.y * xp.y);      // 4 |X[k]|^2
            const float pm = fmaf(xm.x, xm.x, xm.y * xm.y);      // 4 |X[240-k]|^2
            Pk[k2] = k <= 120 ? pk : 0.f;                        // k > 120 is formed by the mirror lane
            Pm[k2] = (k >= 1 && k < 120) ? pm : 0.f;             // bin 240 is not used; 120 is its own mirror
        }
        // ---- mel filterbank (dense rows, 8 filters per pass) + ln, src/mfcc/extractor.rs:121-145
        for (int i0 = 0; i0 < K1; i0 += 8) {
            float acc[8];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                acc[ii] = 0.f;
                if (i0 + ii < K1) {
                    const float *rk = fbk + (i0 + ii) * kBins, *rm = fbm + (i0 + ii) * kBins;
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) {
                        acc[ii] = fmaf(Pk[k2], rk[16 * k2], acc[ii]);
                        acc[ii] = fmaf(Pm[k2], rm[-16 * k2], acc[ii]);
                    }
                }
            }
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const float tot = row16_sum(acc[ii]);
                if (l == ii && i0 + ii < K1) lgb[i0 + ii] = logf(0.25f * tot + FLT_MIN);
            }
        }
        wave_lds_sync();
        // ---- DCT-II x2, coefficient 0 dropped, src/mfcc/extractor.rs:84,146-163.  Sequential
        // multiply-then-add in the reference's order (NOT fused): see the file header.
        const size_t j = j0 + grp;
        if (j < first_frame + n_frames) {
            float *dst = out + (s * out_frame_pitch + (j - first_frame)) * (size_t)K;
            for (int c = 1 + l; c <= K; c += 16) {
                float sum = 0.f;
                for (int n = 0; n < K1; ++n) sum += lgb[n] * dct[c * K1 + n];
                dst[c - 1] = 2.f * sum;
            }
        }
        wave_lds_sync();  // lgb / scratch are reused by the next tile
    }
}

template <class TIN>
static hipError_t launch_mfcc_t(hipStream_t st, const MfccTablesDev &tb, const TIN *pcm, size_t S, size_t n_samples,
                                size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc) {
    if (S == 0 || n_frames == 0) return hipSuccess;
    const size_t tiles = (n_frames + kMfccFramesPerWave - 1) / kMfccFramesPerWave;
    const size_t total = tiles * S;
    if (tiles > 0xffffffffULL) return hipErrorInvalidValue;
    const size_t lds = mfcc_lds_bytes(tb.K1);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // 4-sample vector loads need rows aligned to 4 samples (and at least one full vector before the last sample)
    const bool vec4 = (reinterpret_cast<uintptr_t>(pcm) % (4 * sizeof(TIN)) == 0) && (pcm_stride % 4 == 0) && n_samples >= 8;
    // persistent grid: 3 workgroups of 4 waves per CU x 2 rounds, fewer for small problems
    size_t blocks = (total + kMfccWaves - 1) / kMfccWaves;
    if (blocks > 1536) blocks = 1536;
#define RP_MFCC_LAUNCH(V, KT)                                                                                              \
    do {                                                                                                                   \
        static bool attr_done = false;                                                                                     \
        if (!attr_done) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mfcc_kernel<V, KT, TIN>),                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
            if (e != hipSuccess) return e;                                                                                 \
            attr_done = true;                                                                                              \
        }                                                                                                                  \
        hipLaunchKernelGGL((mfcc_kernel<V, KT, TIN>), dim3((unsigned)blocks), dim3(kMfccThreads), lds, st, pcm, n_samples, \
                           pcm_stride, (unsigned)tiles, total, first_frame, n_frames, out_frame_pitch, tb.K1, tb.hamming,  \
                           tb.tw240, tb.tw480, tb.fb, tb.dct, mfcc);                                                       \
    } while (0)
    if (vec4 && tb.K1 == 6) RP_MFCC_LAUNCH(true, 6);
    else if (vec4 && tb.K1 == 17) RP_MFCC_LAUNCH(true, 17);
    else if (vec4) RP_MFCC_LAUNCH(true, 0);
    else RP_MFCC_LAUNCH(false, 0);
#undef RP_MFCC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_mfcc(hipStream_t st, const MfccTablesDev &tb, const float *pcm, size_t S, size_t n_samples,
                       size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc) {
    return launch_mfcc_t<float>(st, tb, pcm, S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc);
}

// fmt: 0 i8, 1 i16, 2 i32, 3 f32 (rp_sample_format); samples in host byte order
hipError_t launch_mfcc_fmt(hipStream_t st, const MfccTablesDev &tb, const void *pcm, int fmt, size_t S, size_t n_samples,
                           size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc) {
    switch (fmt) {
    case 0: return launch_mfcc_t<int8_t>(st, tb, static_cast<const int8_t *>(pcm), S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc);
    case 1: return launch_mfcc_t<int16_t>(st, tb, static_cast<const int16_t *>(pcm), S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc);
    case 2: return launch_mfcc_t<int32_t>(st, tb, static_cast<const int32_t *>(pcm), S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc);
    case 3: return launch_mfcc_t<float>(st, tb, static_cast<const float *>(pcm), S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc);
    }
    return hipErrorInvalidValue;
}

// -------------------------------------------------------------------------- DTW
// Scoring of one (window, template) pair, reference semantics:
//   window = frames [w, w+L) of the stream (cut to the template length L keeping the
//   OLDEST frames, wakeword_comp.rs:22-27), column-mean normalised (normalizer.rs);
//   D[r][c] = (1 - cos(a[r-1], b[c-1])) + min(D[r-1][c], D[r][c-1], D[r-1][c-1]) on the band
//   c in [r-W, r+W-1]; result D[m-1][n] (dtw.rs:101); score = 1/(1+exp((cost/(m+n)-ref)/ref)).
// Template rows are pre-scaled to unit length on the host and window frames are scaled
// to unit length here, so a cell costs K fused multiply-adds instead of three dot
// products, a sqrt and a divide (comparator.rs:28-48); zero vectors stay zero, which
// reproduces the reference's "magnitude == 0 -> similarity 0".
constexpr int kDtwWin = 64;   // windows per wave

// One wave = 64 consecutive windows of one stream x one chunk of TC same-length templates.
// Per lane: the window's column means, a ring of the 2W unit-length window frames inside the
// band (shared by all TC templates), and TC bands of 2W+1 running costs held as register pairs of
// two templates: one v_pk_fma_f32 forms the cosine costs of a band cell for both templates (the
// coefficient pair comes from scalar registers, the window component is broadcast by op_sel),
// one v_pk_add_f32 adds the two v_min3_f32 results.  Rows are unrolled 2W at a time so
// every ring slot and band index is a compile-time register.  Only the first 2W rows can touch
// columns c < 1 and need the +inf guard; columns c > n are never read back by an in-range cell
// (they only feed cells further right / below-right), so they are left unguarded.
// GX: lanes read their window's frames straight from global memory instead of an LDS stage: used when a
// stream contributes only a few windows per launch (streaming batches), so that the 64 lanes of a wave
// can belong to many different streams.
template <int K, int W, int TC, bool GX>
__global__ __launch_bounds__(kDtwWin) void dtw_band_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, unsigned n_chunks,
    int chunk_base, size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks,
    const float *__restrict__ dup, int T, float score_ref, float *__restrict__ scores, float *__restrict__ avg,
    int flat, size_t n_streams) {
    constexpr int B = 2 * W;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;  // odd pitch: conflict-free lane-strided LDS reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [64 + 2 (L + W)][KP]: up to two stream segments

    const unsigned tile = blockIdx.x % tiles;
    const unsigned ci = (blockIdx.x / tiles) % n_chunks;
    const int lane = threadIdx.x;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    // Lane -> (stream, window).  flat != 0: the 64 lanes are consecutive entries of the flattened
    // (stream, window) space, so a wave may straddle two streams (needs n_win >= 64) and no lane is
    // wasted on a ragged last tile; flat == 0: tiles never cross a stream.
    size_t s;
    int w;
    bool valid;
    const float *xl;
    if (GX) {
        static_assert(!GX || KP == K, "global-memory frames have pitch K");
        const size_t f = (size_t)tile * kDtwWin + lane;
        valid = f < n_streams * n_win;
        s = valid ? f / n_win : 0;
        w = valid ? (int)(f - s * n_win) : 0;
        xl = mfcc + (s * frame_pitch + first_win + (size_t)w) * K;
    } else {
        size_t sA, sB = 0;
        int wA, nA, nB = 0;
        if (flat) {
            const size_t f0 = (size_t)tile * kDtwWin;  // here `tiles` counts flattened tiles and there is no stream index
            sA = f0 / n_win;
            wA = (int)(f0 - sA * n_win);
            nA = (int)n_win - wA < kDtwWin ? (int)n_win - wA : kDtwWin;
            if (nA < kDtwWin && sA + 1 < n_streams) { sB = sA + 1; nB = kDtwWin - nA; }
        } else {
            sA = blockIdx.x / ((size_t)tiles * n_chunks);
            wA = (int)tile * kDtwWin;
            nA = (int)n_win - wA < kDtwWin ? (int)n_win - wA : kDtwWin;
        }
        const int segA = nA + L + W;  // frames staged for the first stream segment
        {
            const float *src = mfcc + sA * frame_pitch * K;
            const size_t g0 = first_win + wA;
            for (int i = lane; i < segA * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = g0 + f;
                xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        if (nB > 0) {
            const float *src = mfcc + sB * frame_pitch * K;
            const int segB = nB + L + W;
            for (int i = lane; i < segB * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = first_win + f;
                xs[(segA + f) * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        __syncthreads();
        const bool inA = lane < nA;
        valid = inA || (lane - nA < nB);
        s = inA ? sA : sB;
        w = inA ? wA + lane : lane - nA;
        xl = xs + (inA ? lane : (valid ? segA + lane - nA : 0)) * KP;
    }
    // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums
    float mu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = 0.f;
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] += xl[i * KP + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;

    v2f ring[B / 2][K];  // ring[j][k] = { y_slot(2j)[k], y_slot(2j+1)[k] }
#pragma unroll
    for (int j = 0; j < B / 2; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = (v2f){0.f, 0.f};

#define RP_LOAD_COL(c, slot)                                                              \
    do {                                                                                  \
        float y_[K], bb_ = 0.f;                                                           \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            y_[k] = xl[((c)-1) * KP + k] - mu[k];                                         \
            bb_ = fmaf(y_[k], y_[k], bb_);                                                \
        }                                                                                 \
        const float inv_ = bb_ > 0.f ? rsqrtf(bb_) : 0.f;                                 \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            if (((slot)&1) == 0) ring[(slot) / 2][k].x = y_[k] * inv_;                    \
            else ring[(slot) / 2][k].y = y_[k] * inv_;                                    \
        }                                                                                 \
    } while (0)

#pragma unroll
    for (int c = 1; c < W; ++c) RP_LOAD_COL(c, c % B);

    // P[tp][q] = D[r-1][(r-1-W)+q] of templates (2tp, 2tp+1); row 0 has D[0][0] = 0 at q = W
    v2f P[TC / 2][B + 1];
#pragma unroll
    for (int t = 0; t < TC / 2; ++t) {
#pragma unroll
        for (int q = 0; q <= B; ++q) P[t][q] = (v2f){RP_INF, RP_INF};
        P[t][W] = (v2f){0.f, 0.f};
    }

    const float *rows = dup + ch->rows_off;
#define RP_ROWS(GUARD)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < B; ++u) {                                                    \
        const int r = r0 + u;                                                                          \
        if (r < L) { /* rows 1..m-1 only: row m is never read (dtw.rs:101) */                          \
            RP_LOAD_COL(r + W - 1, (u + W) % B);                                                       \
            _Pragma("unroll") for (int t = 0; t < TC / 2; ++t) {                                       \
                /* coefficients of the template pair, interleaved (t0,t1) per k: one scalar pair */    \
                const v2f *arow = reinterpret_cast<const v2f *>(rows + ((size_t)(r - 1) * (TC / 2) + t) * K * 2); \
                v2f a2[K];                                                                             \
                _Pragma("unroll") for (int k = 0; k < K; ++k) a2[k] = arow[k];                         \
                /* costs of the 2W band cells first (independent FMA chains), then the serial min chain */ \
                v2f d[B];                                                                              \
                _Pragma("unroll") for (int q = 0; q < B; ++q) d[q] = (v2f){1.f, 1.f};                  \
                _Pragma("unroll") for (int k = 0; k < K; ++k) {                                        \
                    _Pragma("unroll") for (int q = 0; q < B; ++q) {                                    \
                        const int slot = (1 + u + q + B - W) % B;                                      \
                        const v2f yy = (slot & 1) ? ring[slot / 2][k].yy : ring[slot / 2][k].xx;       \
                        d[q] = __builtin_elementwise_fma(-a2[k], yy, d[q]);                            \
                    }                                                                                  \
                }                                                                                      \
                v2f left = (v2f){RP_INF, RP_INF};                                                      \
                _Pragma("unroll") for (int q = 0; q < B; ++q) {                                        \
                    v2f m;                                                                             \
                    m.x = fminf(fminf(P[t][q + 1].x, left.x), P[t][q].x);                              \
                    m.y = fminf(fminf(P[t][q + 1].y, left.y), P[t][q].y);                              \
                    v2f v = d[q] + m;                                                                  \
                    if (GUARD) v = (r - W + q >= 1) ? v : (v2f){RP_INF, RP_INF};                       \
                    P[t][q] = v;                                                                       \
                    left = v;                                                                          \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }

    {
        const int r0 = 1;
        RP_ROWS(true)
    }
    for (int r0 = 1 + B; r0 < L; r0 += B) { RP_ROWS(false) }
#undef RP_ROWS
#undef RP_LOAD_COL

    if (valid) {
        const size_t row = s * out_win_pitch + (size_t)w;
        const float denom = (float)(L + L);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (t < ch->count) {
                const float cost = (t & 1) ? P[t / 2][W + 1].y : P[t / 2][W + 1].x;  // D[m-1][n] for m == n
                const float nc = cost / denom;
                const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
                const int tid = ch->tid[t];
                if (tid < T) scores[row * T + tid] = sc;
                else avg[row] = sc;
            }
        }
    }
}

// Variant for wide frames (K = 16): the ring alone is 160 registers, so the band costs are formed one
// template at a time with two band cells per v_pk_fma_f32 (coefficient duplicated into a scalar pair)
// instead of holding the costs of a template pair for all 2W cells.
template <int K, int W, int TC>
__global__ __launch_bounds__(kDtwWin) void dtw_band_wide_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, unsigned n_chunks,
    int chunk_base, size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks,
    const float *__restrict__ dup, int T, float score_ref, float *__restrict__ scores, float *__restrict__ avg) {
    constexpr int B = 2 * W;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;  // odd pitch: conflict-free lane-strided LDS reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [64 + L + W][KP]

    const unsigned tile = blockIdx.x % tiles;
    const unsigned ci = (blockIdx.x / tiles) % n_chunks;
    const size_t s = blockIdx.x / ((size_t)tiles * n_chunks);
    const int lane = threadIdx.x;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    const size_t w0 = first_win + (size_t)tile * kDtwWin;

    const int n_stage = kDtwWin + L + W;
    const float *src = mfcc + s * frame_pitch * K;
    for (int i = lane; i < n_stage * K; i += kDtwWin) {
        int f = i / K, k = i - f * K;
        size_t g = w0 + f;
        xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
    }
    __syncthreads();

    const float *xl = xs + lane * KP;
    // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums
    float mu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = 0.f;
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] += xl[i * KP + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;

    v2f ring[B / 2][K];  // ring[j][k] = { y_slot(2j)[k], y_slot(2j+1)[k] }
#pragma unroll
    for (int j = 0; j < B / 2; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = (v2f){0.f, 0.f};

#define RP_LOAD_COL(c, slot)                                                              \
    do {                                                                                  \
        float y_[K], bb_ = 0.f;                                                           \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            y_[k] = xl[((c)-1) * KP + k] - mu[k];                                         \
            bb_ = fmaf(y_[k], y_[k], bb_);                                                \
        }                                                                                 \
        const float inv_ = bb_ > 0.f ? rsqrtf(bb_) : 0.f;                                 \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            if (((slot)&1) == 0) ring[(slot) / 2][k].x = y_[k] * inv_;                    \
            else ring[(slot) / 2][k].y = y_[k] * inv_;                                    \
        }                                                                                 \
    } while (0)

#pragma unroll
    for (int c = 1; c < W; ++c) RP_LOAD_COL(c, c % B);

    // P[t][q] = D[r-1][(r-1-W)+q] of template t; row 0 has D[0][0] = 0 at q = W
    float P[TC][B + 1];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
#pragma unroll
        for (int q = 0; q <= B; ++q) P[t][q] = RP_INF;
        P[t][W] = 0.f;
    }

    const float *rows = dup + ch->rows_off;
#define RP_ROWS(GUARD)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < B; ++u) {                                                    \
        const int r = r0 + u;                                                                          \
        if (r < L) { /* rows 1..m-1 only: row m is never read (dtw.rs:101) */                          \
            RP_LOAD_COL(r + W - 1, (u + W) % B);                                                       \
            _Pragma("unroll") for (int t = 0; t < TC; ++t) {                                           \
                const float *arow = rows + ((size_t)(r - 1) * (TC / 2) + t / 2) * K * 2 + (t & 1);     \
                v2f dd[B / 2];                                                                         \
                _Pragma("unroll") for (int j = 0; j < B / 2; ++j) dd[j] = (v2f){1.f, 1.f};             \
                _Pragma("unroll") for (int k = 0; k < K; ++k) {                                        \
                    const v2f a2 = (v2f){arow[2 * k], arow[2 * k]};                                   \
                    _Pragma("unroll") for (int j = 0; j < B / 2; ++j)                                  \
                        dd[j] = __builtin_elementwise_fma(-a2, ring[j][k], dd[j]);                     \
                }                                                                                      \
                float left = RP_INF;                                                                   \
                _Pragma("unroll") for (int q = 0; q < B; ++q) {                                        \
                    const int slot = (1 + u + q + B - W) % B;                                          \
                    const float d = (slot & 1) ? dd[slot / 2].y : dd[slot / 2].x;                      \
                    float v = d + fminf(fminf(P[t][q + 1], left), P[t][q]);                            \
                    if (GUARD) v = (r - W + q >= 1) ? v : RP_INF;                                      \
                    P[t][q] = v;                                                                       \
                    left = v;                                                                          \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }

    {
        const int r0 = 1;
        RP_ROWS(true)
    }
    for (int r0 = 1 + B; r0 < L; r0 += B) { RP_ROWS(false) }
#undef RP_ROWS
#undef RP_LOAD_COL

    if (tile * (size_t)kDtwWin + lane < n_win) {
        const size_t row = s * out_win_pitch + (size_t)tile * kDtwWin + lane;
        const float denom = (float)(L + L);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (t < ch->count) {
                const float cost = P[t][W + 1];  // D[m-1][n] for m == n
                const float nc = cost / denom;
                const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
                const int tid = ch->tid[t];
                if (tid < T) scores[row * T + tid] = sc;
                else avg[row] = sc;
            }
        }
    }
}

// Fallback for any K / band size / m != n: one wave = 64 windows x one template, band
// and column means in LDS (lane-minor, conflict-free), costs evaluated per cell.
__global__ __launch_bounds__(64) void dtw_generic_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, size_t first_win,
    size_t n_win, size_t out_win_pitch, const int *__restrict__ lens, const float *__restrict__ unit, int Lpad, int K,
    int T, int Ttot, int max_len, int band, float score_ref, float *__restrict__ scores, float *__restrict__ avg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KP = K | 1;
    const unsigned tile = blockIdx.x % tiles;
    const int t = (blockIdx.x / tiles) % Ttot;
    const size_t s = blockIdx.x / ((size_t)tiles * Ttot);
    const int lane = threadIdx.x;
    const size_t w0 = first_win + (size_t)tile * 64;
    const int m = lens[t];
    const int n = m < max_len ? m : max_len;  // window cut to the template length
    const int diff = m > n ? m - n : n - m;
    const int W = band > diff ? band : diff;
    const int B = 2 * W;

    const int n_stage = 64 + max_len - 1;
    float *xs = reinterpret_cast<float *>(smem);      // [n_stage][KP]
    float *mus = xs + (size_t)n_stage * KP;           // [K][64]
    float *Pb = mus + (size_t)K * 64;                 // [B+1][64]
    const float *src = mfcc + s * frame_pitch * K;
    for (int i = lane; i < n_stage * K; i += 64) {
        int f = i / K, k = i - f * K;
        size_t g = w0 + f;
        xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
    }
    __syncthreads();
    const float *xl = xs + lane * KP;
    for (int k = 0; k < K; ++k) {
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += xl[i * KP + k];
        mus[k * 64 + lane] = sum / (float)n;
    }
    for (int q = 0; q <= B; ++q) Pb[q * 64 + lane] = RP_INF;
    Pb[W * 64 + lane] = 0.f;
    const float *trow = unit + (size_t)t * Lpad * K;
    for (int r = 1; r < m; ++r) {
        float left = RP_INF;
        for (int q = 0; q < B; ++q) {
            const int c = r - W + q;
            float v = RP_INF;
            if (c >= 1 && c <= n) {
                float dot = 0.f, bb = 0.f;
                for (int k = 0; k < K; ++k) {
                    float y = xl[(c - 1) * KP + k] - mus[k * 64 + lane];
                    dot = fmaf(trow[(r - 1) * K + k], y, dot);
                    bb = fmaf(y, y, bb);
                }
                float cosv = bb > 0.f ? dot * rsqrtf(bb) : 0.f;
                v = (1.f - cosv) + fminf(fminf(Pb[(q + 1) * 64 + lane], left), Pb[q * 64 + lane]);
            }
            Pb[q * 64 + lane] = v;
            left = v;
        }
    }
    if (tile * (size_t)64 + lane < n_win) {
        const int qs = n - (m - 1 - W);  // column n of row m-1
        float cost = (qs >= 0 && qs < B) ? Pb[qs * 64 + lane] : RP_INF;
        float nc = cost / (float)(m + n);
        float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
        size_t row = s * out_win_pitch + (size_t)tile * 64 + lane;
        if (t < T) scores[row * T + t] = sc;
        else avg[row] = sc;
    }
}

template <int K, int W, int TC>
static hipError_t launch_dtw_class(hipStream_t st, const TemplatesDev &t, int cls, int n_chunks, const float *mfcc, size_t S,
                                   size_t frame_pitch, size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch,
                                   float score_ref, float *scores, float *avg, bool few_windows) {
    if (n_chunks <= 0) return hipSuccess;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    if (few_windows && KP == K && W == 5) {
        // streams contribute fewer than 64 windows each: lanes of a wave span many streams and read their frames
        // from global memory (the caller guarantees W*K floats of slack after the last stream's frames)
        const size_t ft = (S * n_win + kDtwWin - 1) / kDtwWin;
        const size_t blocks = ft * (size_t)n_chunks;
        if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
        if (W == 5)
            hipLaunchKernelGGL((dtw_band_kernel<K, 5, TC, (KP == K)>), dim3((unsigned)blocks), dim3(kDtwWin), 0, st, mfcc, frame_pitch,
                               frame_pitch, (unsigned)ft, (unsigned)n_chunks, t.class_first[cls], first_win, n_win, out_win_pitch,
                               t.chunks, t.dup, t.T, score_ref, scores, avg, 1, S);
        return hipGetLastError();
    }
    // flattened (stream, window) tiling when every stream has at least one full tile of windows
    const int flat = (n_win >= (size_t)kDtwWin && S > 1) ? 1 : 0;
    const size_t ft = flat ? (S * n_win + kDtwWin - 1) / kDtwWin : tiles;
    const size_t blocks = flat ? ft * (size_t)n_chunks : tiles * (size_t)n_chunks * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)(kDtwWin + 2 * (t.max_len + W)) * KP * sizeof(float);
    hipLaunchKernelGGL((dtw_band_kernel<K, W, TC, false>), dim3((unsigned)blocks), dim3(kDtwWin), lds, st, mfcc, frame_pitch,
                       frame_pitch, (unsigned)ft, (unsigned)n_chunks, t.class_first[cls], first_win, n_win, out_win_pitch,
                       t.chunks, t.dup, t.T, score_ref, scores, avg, flat, S);
    return hipGetLastError();
}

template <int K, int W, int TC>
static hipError_t launch_dtw_wide(hipStream_t st, const TemplatesDev &t, int cls, int n_chunks, const float *mfcc, size_t S,
                                  size_t frame_pitch, size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch,
                                  float score_ref, float *scores, float *avg) {
    if (n_chunks <= 0) return hipSuccess;
    const size_t blocks = tiles * (size_t)n_chunks * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    const size_t lds = (size_t)(kDtwWin + t.max_len + W) * KP * sizeof(float);
    hipLaunchKernelGGL((dtw_band_wide_kernel<K, W, TC>), dim3((unsigned)blocks), dim3(kDtwWin), lds, st, mfcc, frame_pitch,
                       frame_pitch, (unsigned)tiles, (unsigned)n_chunks, t.class_first[cls], first_win, n_win, out_win_pitch,
                       t.chunks, t.dup, t.T, score_ref, scores, avg);
    return hipGetLastError();
}

// Largest template tile the register kernels are built for at this (mfcc_size, band) (0 = only the generic
// kernel applies).  Built: mfcc_size 5 with band 3..6, mfcc_size 16 with band 5.
int dtw_register_tile(int K, int band) {
    if (K == 5 && band >= 3 && band <= 6) return 8;
    if (K == 16 && band == 5) return 2;
    return 0;
}

template <int W>
static hipError_t launch_dtw_k5(hipStream_t st, const TemplatesDev &t, int n2, const float *mfcc, size_t S, size_t frame_pitch,
                                size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref,
                                float *scores, float *avg, bool few) {
    hipError_t e;
    if ((e = launch_dtw_class<5, W, 2>(st, t, 0, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few)) != hipSuccess) return e;
    if ((e = launch_dtw_class<5, W, 4>(st, t, 1, t.class_count[1], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few)) != hipSuccess) return e;
    return launch_dtw_class<5, W, 8>(st, t, 2, t.class_count[2], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few);
}

hipError_t launch_dtw(hipStream_t st, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                      size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, int with_avg,
                      float *scores, float *avg, bool padded_rows) {
    if (S == 0 || n_win == 0) return hipSuccess;
    // many streams with few windows each (streaming batches): cross-stream waves reading frames from global memory;
    // needs `padded_rows` (slack after the last stream's frames for the never-used out-of-band columns)
    const bool few = padded_rows && S > 1 && n_win < (size_t)kDtwWin;
    const bool do_avg = with_avg && t.has_avg;
    const int Ttot = t.T + (do_avg ? 1 : 0);
    const size_t tiles = (n_win + kDtwWin - 1) / kDtwWin;
    // the register kernels assume m == n (no template longer than the window)
    if (dtw_register_tile(t.K, band) > 0 && t.max_diff == 0 && t.chunks) {
        const int n2 = t.class_count[0] - ((t.has_avg && !do_avg) ? 1 : 0);
        if (t.K == 5) {
            switch (band) {
            case 3: return launch_dtw_k5<3>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few);
            case 4: return launch_dtw_k5<4>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few);
            case 5: return launch_dtw_k5<5>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few);
            default: return launch_dtw_k5<6>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few);
            }
        }
        return launch_dtw_wide<16, 5, 2>(st, t, 0, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg);
    }
    const size_t blocks = tiles * (size_t)Ttot * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const int KP = t.K | 1;
    // the band is widened to |m-n| inside the kernel; size for the worst case over templates
    const int Wmax = band > t.max_diff ? band : t.max_diff;
    const size_t lds = ((size_t)(64 + t.max_len - 1) * KP + (size_t)t.K * 64 + (size_t)(2 * Wmax + 1) * 64) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_generic_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(dtw_generic_kernel, dim3((unsigned)blocks), dim3(64), lds, st, mfcc, frame_pitch, frame_pitch,
                       (unsigned)tiles, first_win, n_win, out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, Ttot,
                       t.max_len, band, score_ref, scores, avg);
    return hipGetLastError();
}

// -------------------------------------------------------------------- aggregate
// src/wakewords/comp/wakeword_comp.rs:38-49 (get_percentile) and :108-139
__device__ inline float percentile_sorted(const float *v, int n, float percentile) {
    float index = percentile / 100.0f * (float)(n - 1);
    float fl = floorf(index);
    if (fl == index) return v[(int)index];
    int i = (int)fl;
    float d = index - fl;
    return v[i] * (1.0f - d) + v[i + 1] * d;
}

constexpr int kAggMaxT = 256;

__global__ __launch_bounds__(64) void aggregate_kernel(const float *__restrict__ scores, size_t n_rows, int T, int mode,
                                                       float *__restrict__ agg) {
    size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= n_rows) return;
    const float *v = scores + row * T;
    if (mode == 1) {  // Max
        float m = v[0];
        for (int i = 1; i < T; ++i) m = fmaxf(m, v[i]);
        agg[row] = m;
        return;
    }
    if (mode == 0) {  // Average: sequential sum in template order
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += v[i];
        agg[row] = s / (float)T;
        return;
    }
    float tmp[kAggMaxT];
    for (int i = 0; i < T; ++i) {  // insertion sort ascending (total_cmp order for non-NaN scores)
        float x = v[i];
        int j = i - 1;
        while (j >= 0 && tmp[j] > x) { tmp[j + 1] = tmp[j]; --j; }
        tmp[j + 1] = x;
    }
    float p = 50.f;
    switch (mode) {
    case 3: p = 25.f; break;
    case 5: p = 75.f; break;
    case 6: p = 80.f; break;
    case 7: p = 90.f; break;
    case 8: p = 95.f; break;
    default: p = 50.f; break;  // Median, P50
    }
    agg[row] = percentile_sorted(tmp, T, p);
}

hipError_t launch_aggregate(hipStream_t st, const float *scores, size_t n_rows, int T, int mode, float *agg) {
    if (n_rows == 0) return hipSuccess;
    if (T < 1 || T > kAggMaxT) return hipErrorInvalidValue;
    size_t blocks = (n_rows + 63) / 64;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(aggregate_kernel, dim3((unsigned)blocks), dim3(64), 0, st, scores, n_rows, T, mode, agg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- scan
// The partial-detection / countdown state machine of src/detector.rs:377-454 with
// reset() of :290-302, one lane per stream, over precomputed window scores.  Frame f
// is emitted while chunk c = f/3 + 1 is processed; after an emit the extractor and the
// window are cleared, the rest of that chunk's frames are dropped (find_map, :372-375),
// chunk c+1 only refills the extractor, so the next frame seen is 3*(f/3) + 6.
// mean(|mfcc|) of every frame, summed in coefficient order like VadDetector::is_voice (src/mfcc/vad.rs:12)
__global__ __launch_bounds__(256) void vad_value_kernel(const float *__restrict__ mfcc, size_t n, int K, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *v = mfcc + i * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += fabsf(v[k]);
    out[i] = s / (float)K;
}

hipError_t launch_vad_value(hipStream_t st, const float *mfcc, size_t n_frames_total, int K, float *out) {
    if (n_frames_total == 0) return hipSuccess;
    const size_t blocks = (n_frames_total + 255) / 256;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vad_value_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mfcc, n_frames_total, K, out);
    return hipGetLastError();
}

// the same over rows of `pitch` frames: out [S][n] from mfcc [S][pitch][K]
__global__ __launch_bounds__(256) void vad_value_rows_kernel(const float *__restrict__ mfcc, size_t S, size_t n, size_t pitch, int K,
                                                             float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * n) return;
    const size_t s = i / n, f = i - s * n;
    const float *v = mfcc + (s * pitch + f) * K;
    float a = 0.f;
    for (int k = 0; k < K; ++k) a += fabsf(v[k]);
    out[i] = a / (float)K;
}
hipError_t launch_vad_value_rows(hipStream_t st, const float *mfcc, size_t S, size_t n, size_t pitch, int K, float *out) {
    if (S * n == 0) return hipSuccess;
    const size_t blocks = (S * n + 255) / 256;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vad_value_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mfcc, S, n, pitch, K, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void scan_kernel(const float *__restrict__ agg, const float *__restrict__ avg,
                                                  const float *__restrict__ vad_value, float vad_mode_value, size_t S,
                                                  size_t n_frames, ScanConfig cfg, BatchDetection *__restrict__ det,
                                                  int32_t *__restrict__ n_det, int max_det) {
    __shared__ float vwin[50][64];  // VadDetector::window, one column per stream (lane)
    __shared__ unsigned long long candidates;  // bit r: stream r of this block has a window that can fire
    const int lane = threadIdx.x;
    const long max_len = cfg.max_len;
    const long n_win = (long)n_frames - max_len + 1;
    // A detection needs at least one window whose aggregate passes the thresholds (run_detection :411-429;
    // the VAD only gates).  The wave first sweeps the block's 64 score rows with coalesced loads; streams
    // without such a window (all of them on non-matching audio) are done, the others run the state machine.
    {
        if (lane == 0) candidates = 0;
        __syncthreads();
        const size_t s0 = (size_t)blockIdx.x * 64;
        const size_t rows = S - s0 < 64 ? S - s0 : 64;
        const size_t total = n_win > 0 ? rows * (size_t)n_win : 0;  // the block's score rows are one contiguous range
        const float *a0 = agg + s0 * (size_t)(n_win > 0 ? n_win : 0);
        const float *v0 = cfg.avg_enabled ? avg + s0 * (size_t)(n_win > 0 ? n_win : 0) : nullptr;
        unsigned long long mask = 0;
        for (size_t e = lane; e < total; e += 64) {
            bool pass = a0[e] > cfg.threshold;
            if (pass && v0) pass = !(v0[e] < cfg.avg_threshold);
            if (pass) mask |= 1ull << (e / (size_t)n_win);
        }
        if (mask) atomicOr(&candidates, mask);
        __syncthreads();
    }
    size_t s = (size_t)blockIdx.x * 64 + lane;
    if (s >= S) return;
    if (!((candidates >> lane) & 1ull)) { n_det[s] = 0; return; }
    const float *a = agg + s * (size_t)(n_win > 0 ? n_win : 0);
    const float *v = avg ? avg + s * (size_t)(n_win > 0 ? n_win : 0) : nullptr;
    const float *vv = vad_value ? vad_value + s * n_frames : nullptr;
    // VadDetector state (src/mfcc/vad.rs:3-50)
    int vad_index = 0, voice_countdown = 0;
    if (vv)
        for (int i = 0; i < 50; ++i) vwin[i][lane] = __builtin_nanf("");
    long win_start = 0, resume = 0;
    bool has_partial = false;
    float p_score = 0.f, p_avg = 0.f;
    int p_counter = 0, p_window = 0, countdown = 0, nd = 0;
    for (long f = 0; f < (long)n_frames; ++f) {
        if (f < resume) continue;
        // process_new_mfccs :379-383: the VAD only sees a frame while no partial detection exists
        bool should_run = true;
        if (vv && !has_partial) {
            vwin[vad_index][lane] = vv[f];
            vad_index = vad_index >= 49 ? 0 : vad_index + 1;
            float mn = RP_INF;
            for (int i = 0; i < 50; ++i) { float w = vwin[i][lane]; if (w == w && w < mn) mn = w; }
            mn = fmaxf(mn, 0.01f);
            const float th = mn * vad_mode_value;
            int n_high = 0;
            for (int i = 0; i < 50; ++i) n_high += vwin[i][lane] > th ? 1 : 0;
            if (n_high > 10) voice_countdown = 500;
            if (voice_countdown > 0) { voice_countdown -= 1; should_run = true; } else should_run = false;
        }
        if (f - win_start + 1 < max_len) continue;
        if (!should_run) continue;
        const long w = f - max_len + 1;
        if (countdown != 0) countdown -= 1;
        if (has_partial) {
            bool done = countdown == 0 ? true : (cfg.eager && p_counter >= cfg.min_scores);
            if (done) {
                has_partial = false;  // take()
                if (p_counter >= cfg.min_scores) {
                    if (nd < max_det) {
                        BatchDetection d;
                        d.stream = (int32_t)s; d.frame = (int32_t)f; d.window = p_window; d.counter = p_counter;
                        d.avg_score = p_avg; d.score = p_score;
                        det[s * (size_t)max_det + nd] = d;
                    }
                    ++nd;
                    win_start = resume = 3 * (f / 3) + 6;  // reset()
                    if (vv) {  // vad.reset()
                        for (int i = 0; i < 50; ++i) vwin[i][lane] = __builtin_nanf("");
                        vad_index = 0; voice_countdown = 0;
                    }
                    continue;
                }
            }
        }
        float sc = a[w];
        float av = 0.f;
        bool pass = true;
        if (cfg.avg_enabled) { av = v[w]; pass = !(av < cfg.avg_threshold); }
        if (pass && sc > cfg.threshold) {
            int counter = has_partial ? p_counter + 1 : 1;
            if (!has_partial || p_score < sc) { p_score = sc; p_avg = av; p_window = (int)w; has_partial = true; }
            p_counter = counter;
            countdown = (int)(max_len / 2);
        }
    }
    n_det[s] = nd;
}

hipError_t launch_scan(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                       size_t S, size_t n_frames, const ScanConfig &cfg, BatchDetection *det, int32_t *n_det, int max_det) {
    if (S == 0) return hipSuccess;
    size_t blocks = (S + 63) / 64;
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)blocks), dim3(64), 0, st, agg, avg, vad_value, vad_mode_value, S, n_frames,
                       cfg, det, n_det, max_det);
    return hipGetLastError();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------- resampler
// rubato FftFixedInOut (src/audio/encoder.rs:72-83) as a fixed linear map per output frame, see rp_resampler.cpp:
//   out[s][c*fo + j] = sum_{n < 2*fi} xs[s][c*fi + n] * g2t[j][n]
// (xs holds one history frame in front of the stream).  A [units x 2*fi] x [2*fi x fo] product in f32 on the
// matrix cores: a wave owns 16 consecutive (stream, frame) units and all fo output columns (NT tiles of 16),
// the workgroup stages the matrix in k-groups of 16 through LDS, double buffered.
template <class TIN>
__global__ __launch_bounds__(256) void resample_stage_kernel(const TIN *__restrict__ pcm, int channels, size_t S, size_t n_new, int fi,
                                                             size_t pcm_stride, const float *__restrict__ prev, float *__restrict__ xs) {
    const size_t pitch = (size_t)fi + n_new, total = S * pitch;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t s = i / pitch, k = i - s * pitch;
        float v;
        if (k < (size_t)fi) v = prev ? prev[s * fi + k] : 0.f;
        else v = SampleIn<TIN>::cvt(pcm[s * pcm_stride + (k - fi) * channels]);  // reencode_to_mono: chunk[0] of every frame
        xs[i] = v;
    }
}

hipError_t launch_resample_stage(hipStream_t st, const void *pcm, int fmt, int channels, size_t S, size_t n_chunks, int fi,
                                 size_t pcm_stride, const float *prev, float *xs) {
    if (S == 0) return hipSuccess;
    const size_t n_new = n_chunks * (size_t)fi;
    size_t blocks = (S * (fi + n_new) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    switch (fmt) {
    case 0: hipLaunchKernelGGL(resample_stage_kernel<int8_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int8_t *>(pcm), channels, S, n_new, fi, pcm_stride, prev, xs); break;
    case 1: hipLaunchKernelGGL(resample_stage_kernel<int16_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int16_t *>(pcm), channels, S, n_new, fi, pcm_stride, prev, xs); break;
    case 2: hipLaunchKernelGGL(resample_stage_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int32_t *>(pcm), channels, S, n_new, fi, pcm_stride, prev, xs); break;
    case 3: hipLaunchKernelGGL(resample_stage_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float *>(pcm), channels, S, n_new, fi, pcm_stride, prev, xs); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

constexpr int kRsWaves = 4, kRsKG = 16, kRsPitch = kRsKG + 4;

template <int NT>
__global__ __launch_bounds__(64 * kRsWaves) void resample_mfma_kernel(const float *__restrict__ xs, size_t xs_pitch, size_t n_units,
                                                                      unsigned n_chunks, int fi, int kpad,
                                                                      const float *__restrict__ g2t, float *__restrict__ out,
                                                                      size_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FO = 16 * NT;
    constexpr int BUF = FO * kRsPitch;                 // floats per staged k-group
    constexpr int NV = (FO * (kRsKG / 4) + 64 * kRsWaves - 1) / (64 * kRsWaves);
    float *wbuf = reinterpret_cast<float *>(smem);     // [2][FO][kRsPitch]
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
    const size_t u0 = ((size_t)blockIdx.x * kRsWaves + wave) * 16;
    size_t u = u0 + li;
    if (u >= n_units) u = n_units - 1;                 // rows past the end recompute the last unit; dropped below
    const size_t su = u / n_chunks, cu = u - su * n_chunks;
    const float *xr = xs + su * xs_pitch + cu * (size_t)fi;
    const int k_real = 2 * fi;
    const bool vec = (fi & 3) == 0 && (xs_pitch & 3) == 0;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 wreg[NV];
    auto wload = [&](int g) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kRsWaves;
            const int o = i / (kRsKG / 4), c = i - o * (kRsKG / 4);
            if (o < FO) wreg[v] = *reinterpret_cast<const float4 *>(g2t + (size_t)o * kpad + g * kRsKG + 4 * c);
        }
    };
    auto wstore = [&](float *dst) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kRsWaves;
            const int o = i / (kRsKG / 4), c = i - o * (kRsKG / 4);
            if (o < FO) *reinterpret_cast<float4 *>(dst + o * kRsPitch + 4 * c) = wreg[v];
        }
    };
    const int ngrp = kpad / kRsKG;
    wload(0);
    wstore(wbuf);
    __syncthreads();
    for (int g = 0; g < ngrp; ++g) {
        const float *cur = wbuf + (g & 1) * BUF;
        if (g + 1 < ngrp) wload(g + 1);
        const int k0 = g * kRsKG + 4 * lk;
        float4 a;
        if (vec) a = k0 + 3 < k_real ? *reinterpret_cast<const float4 *>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        else {
            a.x = k0 + 0 < k_real ? xr[k0 + 0] : 0.f; a.y = k0 + 1 < k_real ? xr[k0 + 1] : 0.f;
            a.z = k0 + 2 < k_real ? xr[k0 + 2] : 0.f; a.w = k0 + 3 < k_real ? xr[k0 + 3] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float4 b = *reinterpret_cast<const float4 *>(cur + (16 * n + li) * kRsPitch + 4 * lk);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[n], 0, 0, 0);
        }
        if (g + 1 < ngrp) wstore(wbuf + ((g + 1) & 1) * BUF);
        __syncthreads();
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t ue = u0 + 4 * lk + e;
        if (ue >= n_units) continue;
        const size_t s = ue / n_chunks, c = ue - s * n_chunks;
        float *dst = out + s * out_stride + c * (size_t)FO + li;
#pragma unroll
        for (int n = 0; n < NT; ++n) dst[16 * n] = acc[n][e];
    }
}

template <int NT>
static hipError_t launch_resample_t(hipStream_t st, const ResamplerDev &rs, const float *xs, size_t S, size_t n_chunks, float *out,
                                    size_t out_stride) {
    const size_t units = S * n_chunks;
    const size_t blocks = (units + 16 * kRsWaves - 1) / (16 * kRsWaves);
    if (blocks > 0x7fffffffULL || n_chunks > 0xffffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * 16 * NT * kRsPitch * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_mfma_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(resample_mfma_kernel<NT>, dim3((unsigned)blocks), dim3(64 * kRsWaves), lds, st, xs, (1 + n_chunks) * (size_t)rs.fi,
                       units, (unsigned)n_chunks, rs.fi, rs.kpad, rs.g2t, out, out_stride);
    return hipGetLastError();
}

// The 48 kHz kernel decodes and reads its input where it lies (no staging copy)
bool resample_reads_in_place(const ResamplerDev &rs, const void *pcm, int fmt, size_t pcm_stride, const float *out, size_t out_stride) {
    const size_t eb = fmt == 0 ? 1 : fmt == 1 ? 2 : 4;
    return rs.fft48 && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && ((pcm_stride * eb) & 15) == 0 && (out_stride & 1) == 0 &&
           (reinterpret_cast<uintptr_t>(out) & 7) == 0;
}
hipError_t launch_resample_in_place(hipStream_t st, const ResamplerDev &rs, const void *pcm, int fmt, int channels, size_t pcm_stride,
                                    const float *prev, float *prev_out, size_t S, size_t n_chunks, float *out, size_t out_stride) {
    return launch_resample48(st, rs.fft48, pcm, fmt, channels, pcm_stride, 0, prev, prev_out, S, n_chunks, out, out_stride);
}

hipError_t launch_resample(hipStream_t st, const ResamplerDev &rs, const float *xs, size_t S, size_t n_chunks, float *out,
                           size_t out_stride) {
    if (S == 0 || n_chunks == 0) return hipSuccess;
    if (rs.fft48 && (out_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0)
        return launch_resample48(st, rs.fft48, xs, 3, 1, (1 + n_chunks) * (size_t)rs.fi, 1, nullptr, nullptr, S, n_chunks, out, out_stride);
    if (rs.fo == 480) return launch_resample_t<30>(st, rs, xs, S, n_chunks, out, out_stride);
    if (rs.fo == 640) return launch_resample_t<40>(st, rs, xs, S, n_chunks, out, out_stride);
    return hipErrorInvalidValue;
}

// ---- 48 kHz -> 16 kHz on the FFT-240 machinery of the MFCC kernel -------------------------------------------
// The same unit as resample_mfma_kernel, evaluated the way rubato structures it (transform, filter, truncate,
// inverse transform, overlap-add) but pruned to what is non-zero / kept: the 2 880-point transform of the
// zero-padded 1 440-sample frame is split n = 6m + d into six 240-sample real sequences u_d; only bins q < 480
// are needed, X[q] = sum_d W2880^{dq} U_d[q] with U_d the 480-point transform of the zero-padded u_d:
//   even q = 2t:   U_d[2t]   = DFT240(u_d)[t]                     (two real sequences per complex FFT-240)
//   odd  q = 2t+1: U_d[2t+1] = DFT240(u_d[m] W480^m)[t]           (also pairs up: V[239-t] = conj(V[t]))
// = six FFT-240.  The 960-point real inverse is one complex 480-point inverse (E/O packing) = two FFT-240 and a
// radix-2 step.  One wave per run of consecutive frames of one stream (the overlap half stays in LDS); the
// wave's four 16-lane groups run four FFT-240 at a time (rounds: 4 + 2 forward, 2 inverse).
constexpr int kR48Waves = 4;
constexpr int kR48WaveLds = 6 * 240 * 8 + 480 * 8 + 480 * 4;  // Z buffers | spectrum | overlap half

// v[n1] = z[15*n1 + n2] of lane n2 = l (lane 15 idles); leaves Z[l + 16*k2] in z[k2].  `my` = 240 v2f of scratch.
__device__ __forceinline__ void fft240_lanes(v2f (&v)[16], v2f *my, int l, const v2f (&twl)[16], v2f (&z)[15]) {
    fft16(v);
    if (l < 15) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) my[(c + 4 * d) * 15 + l] = cmul(v[4 * c + d], twl[4 * c + d]);
    }
    wave_lds_sync();
    v2f u[15];
#pragma unroll
    for (int n2 = 0; n2 < 15; ++n2) u[n2] = my[l * 15 + n2];
    dft15(u, z);
    wave_lds_sync();
}

__device__ __forceinline__ v2f cconj(v2f a) { return (v2f){a.x, -a.y}; }
__device__ __forceinline__ v2f mul_pi(v2f a) { return (v2f){-a.y, a.x}; }  // a * (+i)

template <class TIN>
__global__ __launch_bounds__(64 * kR48Waves) void resample48_fft_kernel(
    const TIN *__restrict__ xs, size_t xs_pitch, int channels, int has_hist, const float *__restrict__ prev,
    float *__restrict__ prev_out, size_t n_waves, unsigned n_chunks, unsigned seg_len, unsigned n_seg,
    const v2f *__restrict__ g_tw240, const v2f *__restrict__ g_tw480, const v2f *__restrict__ g_twc,
    const v2f *__restrict__ g_w960c, const v2f *__restrict__ g_hf, float *__restrict__ out, size_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15, l15 = l < 15 ? l : 0;
    const size_t wid = (size_t)blockIdx.x * kR48Waves + wave;
    if (wid >= n_waves) return;  // waves never meet at a workgroup barrier
    const size_t s = wid / n_seg;
    const unsigned seg = (unsigned)(wid - s * n_seg);
    const long c0 = (long)seg * seg_len;
    const long c1 = c0 + seg_len < (long)n_chunks ? c0 + seg_len : (long)n_chunks;
    v2f *zb = reinterpret_cast<v2f *>(smem + (size_t)wave * kR48WaveLds);  // [6][240]
    v2f *xy = zb + 6 * 240;                                                  // [480]
    v2f *tail = xy + 480;                                                    // [240] = 480 floats
    float *xin = reinterpret_cast<float *>(zb);                              // 1440 samples, dead before zb is written
    v2f twl[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) twl[4 * c + d] = g_tw240[l15 * (c + 4 * d)];
    for (int m = lane; m < 240; m += 64) tail[m] = (v2f){0.f, 0.f};
    float *orow = out + s * out_stride;
    // frame c0-1 only seeds the overlap half; in front of frame 0 that is the history frame xs carries
    // (has_hist), the stream's previous input frame `prev` [S][1440], or silence (the overlap half stays zero).
    // Samples are decoded here (Sample::into_f32, first channel of every interleaved frame).
    const bool vec = sizeof(TIN) == 4 && channels == 1 && (xs_pitch & 3) == 0;
    for (long c = (c0 == 0 && !has_hist && !prev) ? 0 : c0 - 1; c < c1; ++c) {
        wave_lds_sync();
        if (c < 0 && prev) {
            const float *x = prev + s * 1440;
            for (int i = lane; i < 360; i += 64) reinterpret_cast<float4 *>(xin)[i] = reinterpret_cast<const float4 *>(x)[i];
        } else {
            const TIN *x = xs + s * xs_pitch + (size_t)(c + (has_hist ? 1 : 0)) * 1440 * channels;
            if (vec) {
                for (int i = lane; i < 360; i += 64)
                    reinterpret_cast<float4 *>(xin)[i] = SampleIn<TIN>::load4(x + 4 * i);
            } else {
                for (int i = lane; i < 1440; i += 64) xin[i] = SampleIn<TIN>::cvt(x[(size_t)i * channels]);
            }
        }
        wave_lds_sync();
        if (prev_out && c == (long)n_chunks - 1) {  // the last input frame is the next call's history
            float *po = prev_out + s * 1440;
            for (int i = lane; i < 360; i += 64) reinterpret_cast<float4 *>(po)[i] = reinterpret_cast<const float4 *>(xin)[i];
        }
        // ---- forward round 1: groups 0..2 = even bins of the pairs (0,1) (2,3) (4,5), group 3 = odd bins of pair (0,1).
        // z[m] = (x[6m+2p], x[6m+2p+1]) [* W480^m for the odd bins], m = 15*n1 + n2.  The samples occupy zb[0..2];
        // the results go to zb[3..5] and xy[0..239] so that round 2 can still read them.
        v2f va[16], z[15];
        {
            const int p1 = grp < 3 ? grp : 0;
            const v2f *s1 = reinterpret_cast<const v2f *>(xin + 6 * l15 + 2 * p1);
            const bool odd1 = grp == 3;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const v2f a = s1[45 * n1];
                va[n1] = odd1 ? cmul(a, g_tw480[15 * n1 + l15]) : a;
            }
            v2f *my = grp < 3 ? zb + (3 + grp) * 240 : xy;
            fft240_lanes(va, my, l, twl, z);
#pragma unroll
            for (int k2 = 0; k2 < 15; ++k2) my[l + 16 * k2] = z[k2];
        }
        // ---- forward round 2: groups 0,1 = odd bins of pairs (2,3), (4,5) -> zb[0], zb[1]; groups 2,3 repeat them
        // into dead space (zb[2], xy[240..479])
        {
            const int p2 = 1 + (grp & 1);
            const v2f *s2 = reinterpret_cast<const v2f *>(xin + 6 * l15 + 2 * p2);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) va[n1] = cmul(s2[45 * n1], g_tw480[15 * n1 + l15]);
            wave_lds_sync();  // every lane holds its samples: zb[0..2] may be overwritten
            v2f *my = grp < 3 ? zb + grp * 240 : xy + 240;
            fft240_lanes(va, my, l, twl, z);
#pragma unroll
            for (int k2 = 0; k2 < 15; ++k2) my[l + 16 * k2] = z[k2];
        }
        wave_lds_sync();
        // ---- untangle the pairs, apply the radix-6 twiddles, filter: Y[q] = H[q]/2 * sum_d W2880^{dq} 2U_d[q];
        // Y[2t] -> zb[2][t], Y[2t+1] -> xy[240 + t] (both dead)
        const v2f *zodd[3] = {xy, zb, zb + 240};
#pragma unroll 1
        for (int t = lane; t < 240; t += 64) {
            const int tm = t == 0 ? 0 : 240 - t;
            v2f xe = (v2f){0.f, 0.f}, xo = (v2f){0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const v2f ze = zb[(3 + p) * 240 + t], zem = cconj(zb[(3 + p) * 240 + tm]);
                const v2f zo = zodd[p][t], zom = cconj(zodd[p][239 - t]);
                const v2f ea = ze + zem, eb = mul_mi(ze - zem);   // 2 U_a[2t], 2 U_b[2t]
                const v2f oa = zo + zom, ob = mul_mi(zo - zom);   // 2 U_a[2t+1], 2 U_b[2t+1]
                const float4 wb = *reinterpret_cast<const float4 *>(g_twc + (2 * p + 1) * 480 + 2 * t);
                if (p == 0) {  // d = 0: twiddle 1
                    xe += ea; xo += oa;
                } else {
                    const float4 wa = *reinterpret_cast<const float4 *>(g_twc + (2 * p) * 480 + 2 * t);
                    xe += cmul((v2f){wa.x, wa.y}, ea); xo += cmul((v2f){wa.z, wa.w}, oa);
                }
                xe += cmul((v2f){wb.x, wb.y}, eb); xo += cmul((v2f){wb.z, wb.w}, ob);
            }
            const float4 h = *reinterpret_cast<const float4 *>(g_hf + 2 * t);
            zb[2 * 240 + t] = cmul(xe, (v2f){h.x, h.y});
            xy[240 + t] = cmul(xo, (v2f){h.z, h.w});
        }
        wave_lds_sync();
        // ---- E/O packing of the Hermitian spectrum (bin 480 = 0) into the two conjugated inputs of the inverse:
        // A[t] = conj(Z'[2t]) -> zb[3], B[t] = conj(Z'[2t+1]) -> zb[4];  Z'[k] = E[k] + i O[k]
        const v2f *yev = zb + 2 * 240, *yod = xy + 240;
        v2f *ab = zb + 3 * 240;
#pragma unroll 1
        for (int k = lane; k <= 240; k += 64) {
            const v2f yk = (k & 1) ? yod[k >> 1] : yev[k >> 1];
            if (k == 0) {
                ab[0] = (v2f){yk.x, -yk.x};                      // conj(Y0 (1 + i)), Y0 real
            } else if (k == 240) {
                ab[120] = yk + yk;                               // conj(2 conj(Y[240]))
            } else {
                const int k2 = 480 - k;
                const v2f ym = cconj((k2 & 1) ? yod[k2 >> 1] : yev[k2 >> 1]);
                const v2f e = yk + ym, o = cmul(yk - ym, g_w960c[k]);
                const v2f zp = e + mul_pi(o);                    // Z'[k]
                const v2f zq = cconj(e) + mul_pi(cconj(o));      // Z'[480-k]
                ab[(k & 1) * 240 + (k >> 1)] = cconj(zp);
                ab[(k2 & 1) * 240 + (k2 >> 1)] = cconj(zq);
            }
        }
        wave_lds_sync();
        // ---- inverse FFT-240 of the even / odd bins: groups 0,1 -> zb[0], zb[1]; groups 2,3 repeat into zb[5], xy[0..239]
        {
            const v2f *src = ab + (grp & 1) * 240 + l15;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) va[n1] = src[15 * n1];
            v2f *my = grp < 2 ? zb + grp * 240 : (grp == 2 ? zb + 5 * 240 : xy);
            fft240_lanes(va, my, l, twl, z);
#pragma unroll
            for (int k2 = 0; k2 < 15; ++k2) my[l + 16 * k2] = z[k2];
        }
        wave_lds_sync();
        // ---- radix-2 step, overlap-add: z[m] = Ee[m] + W480^{-m} Oo[m], z[m+240] = Ee[m] - ...; y[2m], y[2m+1] = z[m]
#pragma unroll 1
        for (int m = lane; m < 240; m += 64) {
            const v2f ee = cconj(zb[m]), oo = cconj(zb[240 + m]);
            const v2f tq = cmul(cconj(g_tw480[m]), oo);
            const v2f z0 = ee + tq, z1 = ee - tq;
            if (c >= c0) *reinterpret_cast<v2f *>(orow + (size_t)c * 480 + 2 * m) = z0 + tail[m];
            tail[m] = z1;
        }
    }
}

template <class TIN>
static hipError_t launch_resample48_t(hipStream_t st, const float *tables, const TIN *xs, size_t xs_pitch, int channels, int has_hist,
                                      const float *prev, float *prev_out, size_t S, size_t n_chunks, float *out, size_t out_stride) {
    // enough waves to fill the chip: split long streams into runs (each run recomputes one frame for its overlap)
    size_t n_seg = S >= 8192 ? 1 : (8192 + S - 1) / S;
    if (n_seg > n_chunks) n_seg = n_chunks;
    const size_t seg_len = (n_chunks + n_seg - 1) / n_seg;
    n_seg = (n_chunks + seg_len - 1) / seg_len;
    const size_t n_waves = S * n_seg, blocks = (n_waves + kR48Waves - 1) / kR48Waves;
    if (blocks > 0x7fffffffULL || n_chunks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)kR48Waves * kR48WaveLds;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample48_fft_kernel<TIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const v2f *t = reinterpret_cast<const v2f *>(tables);
    hipLaunchKernelGGL(resample48_fft_kernel<TIN>, dim3((unsigned)blocks), dim3(64 * kR48Waves), lds, st, xs, xs_pitch, channels, has_hist,
                       prev, prev_out, n_waves, (unsigned)n_chunks, (unsigned)seg_len, (unsigned)n_seg, t + kR48OffTw240,
                       t + kR48OffTw480, t + kR48OffTwc, t + kR48OffW960c, t + kR48OffHf, out, out_stride);
    return hipGetLastError();
}

// 48 kHz input in any sample format / channel count, read where it lies.  prev [S][1440] f32 (nullptr: the streams
// start from silence) is the input frame in front of frame 0; prev_out (nullptr: not kept) receives the last one.
hipError_t launch_resample48(hipStream_t st, const float *tables, const void *pcm, int fmt, int channels, size_t pcm_stride,
                             int has_hist, const float *prev, float *prev_out, size_t S, size_t n_chunks, float *out,
                             size_t out_stride) {
    if (S == 0 || n_chunks == 0) return hipSuccess;
    switch (fmt) {
    case 0: return launch_resample48_t(st, tables, static_cast<const int8_t *>(pcm), pcm_stride, channels, has_hist, prev, prev_out, S, n_chunks, out, out_stride);
    case 1: return launch_resample48_t(st, tables, static_cast<const int16_t *>(pcm), pcm_stride, channels, has_hist, prev, prev_out, S, n_chunks, out, out_stride);
    case 2: return launch_resample48_t(st, tables, static_cast<const int32_t *>(pcm), pcm_stride, channels, has_hist, prev, prev_out, S, n_chunks, out, out_stride);
    case 3: return launch_resample48_t(st, tables, static_cast<const float *>(pcm), pcm_stride, channels, has_hist, prev, prev_out, S, n_chunks, out, out_stride);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------- streaming batches
// State of one live stream between rp_stream_batch_process calls: the detector's countdown / partial
// detection / window bookkeeping (src/detector.rs:62-79) in absolute frame numbers, and the VadDetector.
struct StreamState {
    long long win_start, resume;
    int has_partial, p_counter, countdown, vad_index, voice_countdown, pad;
    long long p_window;
    float p_score, p_avg;
    float vad_window[50];
};

// hist [S][hist_pitch] = the last 480-sample chunk of the previous call (old_hist row + old_off) | the new chunks decoded
template <class TIN>
__global__ __launch_bounds__(256) void stream_stage_kernel(const TIN *__restrict__ pcm, int channels, size_t S, size_t n_new,
                                                           size_t pcm_stride, const float *__restrict__ old_hist, size_t old_off,
                                                           float *__restrict__ hist, size_t hist_pitch) {
    const size_t row = kFrame + n_new, total = S * row;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t s = i / row, k = i - s * row;
        hist[s * hist_pitch + k] = k < (size_t)kFrame ? old_hist[s * hist_pitch + old_off + k]
                                                      : SampleIn<TIN>::cvt(pcm[s * pcm_stride + (k - kFrame) * channels]);
    }
}
// rows [S][src_pitch] -> [S][dst_pitch]: dst[s][0..count) = src[s][src_off .. src_off+count)
__global__ __launch_bounds__(256) void carry_rows_kernel(const float *src, size_t S, size_t src_pitch, size_t src_off, size_t count,
                                                         float *dst, size_t dst_pitch) {
    const size_t total = S * count;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t s = i / count, k = i - s * count;
        dst[s * dst_pitch + k] = src[s * src_pitch + src_off + k];
    }
}

hipError_t launch_stream_stage(hipStream_t st, const void *pcm, int fmt, int channels, size_t S, size_t n_new, size_t pcm_stride,
                               const float *old_hist, size_t old_off, float *hist, size_t hist_pitch) {
    if (S == 0 || n_new == 0) return hipSuccess;
    size_t blocks = (S * (kFrame + n_new) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    switch (fmt) {
    case 0: hipLaunchKernelGGL(stream_stage_kernel<int8_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int8_t *>(pcm), channels, S, n_new, pcm_stride, old_hist, old_off, hist, hist_pitch); break;
    case 1: hipLaunchKernelGGL(stream_stage_kernel<int16_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int16_t *>(pcm), channels, S, n_new, pcm_stride, old_hist, old_off, hist, hist_pitch); break;
    case 2: hipLaunchKernelGGL(stream_stage_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const int32_t *>(pcm), channels, S, n_new, pcm_stride, old_hist, old_off, hist, hist_pitch); break;
    case 3: hipLaunchKernelGGL(stream_stage_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float *>(pcm), channels, S, n_new, pcm_stride, old_hist, old_off, hist, hist_pitch); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_carry_rows(hipStream_t st, const float *src, size_t S, size_t src_pitch, size_t src_off, size_t count, float *dst,
                             size_t dst_pitch) {
    if (S == 0 || count == 0) return hipSuccess;
    size_t blocks = (S * count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(carry_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, S, src_pitch, src_off, count, dst, dst_pitch);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void stream_state_init_kernel(StreamState *__restrict__ st, size_t S) {
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    StreamState z;
    z.win_start = 0; z.resume = 0; z.has_partial = 0; z.p_counter = 0; z.countdown = 0; z.vad_index = 0; z.voice_countdown = 0; z.pad = 0;
    z.p_window = 0; z.p_score = 0.f; z.p_avg = 0.f;
    for (int i = 0; i < 50; ++i) z.vad_window[i] = __builtin_nanf("");
    st[s] = z;
}
hipError_t launch_stream_state_init(hipStream_t st, void *state, size_t S) {
    if (S == 0) return hipSuccess;
    hipLaunchKernelGGL(stream_state_init_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, static_cast<StreamState *>(state), S);
    return hipGetLastError();
}
size_t stream_state_bytes() { return sizeof(StreamState); }

// Rustpotter::reset (src/detector.rs:290-302) for one stream (or all, stream < 0): the next chunk only refills
// the extractor, so the next frame seen is `resume`.
__global__ __launch_bounds__(64) void stream_state_reset_kernel(StreamState *__restrict__ st, size_t S, long long stream, long long resume) {
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= S || (stream >= 0 && (size_t)stream != s)) return;
    StreamState z = st[s];
    z.win_start = z.resume = resume;
    z.has_partial = 0; z.p_counter = 0; z.countdown = 0; z.vad_index = 0; z.voice_countdown = 0;
    for (int i = 0; i < 50; ++i) z.vad_window[i] = __builtin_nanf("");
    st[s] = z;
}
hipError_t launch_stream_state_reset(hipStream_t st, void *state, size_t S, long long stream, long long resume) {
    if (S == 0) return hipSuccess;
    hipLaunchKernelGGL(stream_state_reset_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, static_cast<StreamState *>(state), S,
                       stream, resume);
    return hipGetLastError();
}

// scan_kernel over the n_new frames of this call with carried state.  Frame i of the call is absolute frame
// f0 + i; the window ending at it is row i of agg / avg (the history prefix is max_len-1 frames long).
__global__ __launch_bounds__(64) void scan_stream_kernel(const float *__restrict__ agg, const float *__restrict__ avg,
                                                         const float *__restrict__ vad_value, float vad_mode_value, size_t S,
                                                         long long f0, int n_new, ScanConfig cfg, StreamState *__restrict__ state,
                                                         BatchDetection *__restrict__ det, int32_t *__restrict__ n_det, int max_det) {
    __shared__ float vwin[50][64];
    const int lane = threadIdx.x;
    const size_t s = (size_t)blockIdx.x * 64 + lane;
    if (s >= S) return;
    StreamState z = state[s];
    const long long max_len = cfg.max_len;
    const float *a = agg + s * (size_t)n_new;
    const float *v = avg ? avg + s * (size_t)n_new : nullptr;
    const float *vv = vad_value ? vad_value + s * (size_t)n_new : nullptr;
    if (vv)
        for (int i = 0; i < 50; ++i) vwin[i][lane] = z.vad_window[i];
    int nd = 0;
    for (int i = 0; i < n_new; ++i) {
        const long long f = f0 + i;
        if (f < 0 || f < z.resume) continue;  // frames the extractor never emits (first chunk, refill after a reset)
        bool should_run = true;
        if (vv && !z.has_partial) {
            vwin[z.vad_index][lane] = vv[i];
            z.vad_index = z.vad_index >= 49 ? 0 : z.vad_index + 1;
            float mn = RP_INF;
            for (int j = 0; j < 50; ++j) { float w = vwin[j][lane]; if (w == w && w < mn) mn = w; }
            mn = fmaxf(mn, 0.01f);
            const float th = mn * vad_mode_value;
            int n_high = 0;
            for (int j = 0; j < 50; ++j) n_high += vwin[j][lane] > th ? 1 : 0;
            if (n_high > 10) z.voice_countdown = 500;
            if (z.voice_countdown > 0) { z.voice_countdown -= 1; should_run = true; } else should_run = false;
        }
        if (f - z.win_start + 1 < max_len) continue;
        if (!should_run) continue;
        if (z.countdown != 0) z.countdown -= 1;
        if (z.has_partial) {
            const bool done = z.countdown == 0 ? true : (cfg.eager && z.p_counter >= cfg.min_scores);
            if (done) {
                z.has_partial = 0;
                if (z.p_counter >= cfg.min_scores) {
                    if (nd < max_det) {
                        BatchDetection d;
                        d.stream = (int32_t)s; d.frame = (int32_t)f; d.window = (int32_t)z.p_window; d.counter = z.p_counter;
                        d.avg_score = z.p_avg; d.score = z.p_score;
                        det[s * (size_t)max_det + nd] = d;
                    }
                    ++nd;
                    z.win_start = z.resume = 3 * (f / 3) + 6;
                    if (vv) { for (int j = 0; j < 50; ++j) vwin[j][lane] = __builtin_nanf(""); z.vad_index = 0; z.voice_countdown = 0; }
                    continue;
                }
            }
        }
        const float sc = a[i];
        float av = 0.f;
        bool pass = true;
        if (cfg.avg_enabled) { av = v[i]; pass = !(av < cfg.avg_threshold); }
        if (pass && sc > cfg.threshold) {
            const int counter = z.has_partial ? z.p_counter + 1 : 1;
            if (!z.has_partial || z.p_score < sc) { z.p_score = sc; z.p_avg = av; z.p_window = f - max_len + 1; z.has_partial = 1; }
            z.p_counter = counter;
            z.countdown = (int)(max_len / 2);
        }
    }
    if (vv)
        for (int i = 0; i < 50; ++i) z.vad_window[i] = vwin[i][lane];
    state[s] = z;
    n_det[s] = nd;
}

hipError_t launch_scan_stream(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                              size_t S, long long f0, int n_new, const ScanConfig &cfg, void *state, BatchDetection *det,
                              int32_t *n_det, int max_det) {
    if (S == 0) return hipSuccess;
    hipLaunchKernelGGL(scan_stream_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, agg, avg, vad_value, vad_mode_value, S, f0,
                       n_new, cfg, static_cast<StreamState *>(state), det, n_det, max_det);
    return hipGetLastError();
}

// --------------------------------------------------------------------- front-end
// Sample decode + GainNormalizerFilter + BandPassFilter for whole streams (src/detector.rs:358-371).
// None of it depends on the detection state (reset() leaves both filters alone, :290-302), so it is
// a pure function of the stream: per-chunk RMS in parallel, the gain recursion per stream over the
// chunk RMS values, then gain + biquad per stream along time (one lane per stream: a lane re-reads
// its own 128-byte lines from L1, HBM traffic stays one read + one write of the PCM).
template <class TIN>
__global__ __launch_bounds__(256) void chunk_rms_kernel(const TIN *__restrict__ pcm, size_t S, size_t n_chunks, size_t pcm_stride,
                                                        int vec4, float *__restrict__ rms) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * n_chunks) return;
    const size_t s = i / n_chunks, c = i - s * n_chunks;
    const TIN *x = pcm + s * pcm_stride + c * kFrame;
    float sum_squared = 0.0f;  // GainNormalizerFilter::get_rms_level, gain_normalizer_filter.rs:49-55 (sequential sum)
    if (vec4) {
        constexpr int NB = 24;
        for (int k0 = 0; k0 < kFrame; k0 += 4 * NB) {
            float4 buf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) buf[b] = SampleIn<TIN>::load4(x + k0 + 4 * b);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float4 v = buf[b];
                sum_squared += v.x * v.x; sum_squared += v.y * v.y; sum_squared += v.z * v.z; sum_squared += v.w * v.w;
            }
        }
    } else {
        for (int k = 0; k < kFrame; ++k) { const float v = SampleIn<TIN>::cvt(x[k]); sum_squared += v * v; }
    }
    rms[i] = sqrtf(sum_squared / (float)kFrame);
}

// GainNormalizerFilter::filter, gain_normalizer_filter.rs:14-41, one lane per stream; the RMS window lives
// in LDS ([window_size][64]) when it fits, else in the global ring [S][window_size]
__global__ __launch_bounds__(64) void gain_kernel(const float *__restrict__ rms, size_t S, size_t n_chunks, float rms_level_ref,
                                                  float min_gain, float max_gain, int window_size, int ring_in_lds,
                                                  float *__restrict__ ring, float *__restrict__ gains) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    float *w = ring_in_lds ? reinterpret_cast<float *>(smem) + threadIdx.x : ring + s * (size_t)window_size;
    const int pitch = ring_in_lds ? 64 : 1;
    const float rms_level_sqrt = sqrtf(rms_level_ref);
    int len = 0, head = 0;  // logical window = w[((head + i) % window_size) * pitch], i < len (oldest first)
    for (size_t c = 0; c < n_chunks; ++c) {
        const float r = rms[s * n_chunks + c];
        float gain = 1.f;
        if (!(rms_level_ref != rms_level_ref) && r != 0.f) {
            if (len < window_size) { w[((head + len) % window_size) * pitch] = r; ++len; }
            else { w[head * pitch] = r; head = (head + 1) % window_size; }  // push + drain(0..1)
            float sum = 0.f;
            for (int i = 0; i < len; ++i) sum += w[((head + i) % window_size) * pitch];
            const float frame_rms_level = sum / (float)len;
            gain = rms_level_sqrt / sqrtf(frame_rms_level);
            gain = roundf(gain * 10.f) / 10.f;
            gain = gain < min_gain ? min_gain : gain;  // f32::clamp
            gain = gain > max_gain ? max_gain : gain;
        }
        gains[s * n_chunks + c] = gain;
    }
}

struct BiquadCoef { float a0, a1, a2, b1, b2; };

// gain (+clamp) and BandPassFilter::filter (band_pass_filter.rs:19-30) along the stream, one lane per stream
template <class TIN>
__global__ __launch_bounds__(64) void apply_filters_kernel(const TIN *__restrict__ pcm, size_t S, size_t n_samples, size_t n_chunks,
                                                           size_t pcm_stride, const float *__restrict__ gains, int band_pass,
                                                           BiquadCoef q, int vec4, float *__restrict__ out, size_t out_stride) {
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    const TIN *x = pcm + s * pcm_stride;
    float *y = out + s * out_stride;
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
    for (size_t c = 0; c < n_chunks; ++c) {
        const float g = gains ? gains[s * n_chunks + c] : 1.f;
        auto one = [&](float v) {
            if (g != 1.f) { v = v * g; v = v < -1.f ? -1.f : v; v = v > 1.f ? 1.f : v; }
            if (band_pass) {
                const float o = q.a0 * v + q.a1 * x1 + q.a2 * x2 - q.b1 * y1 - q.b2 * y2;
                x2 = x1; x1 = v; y2 = y1; y1 = o;
                v = o;
            }
            return v;
        };
        if (vec4) {  // 4 samples per load/store, 24 loads in flight: with one wave per SIMD (S/64 waves in all)
                     // the loads must be issued ahead of the dependent filter chain
            constexpr int NB = 24;
            for (int k0 = 0; k0 < kFrame; k0 += 4 * NB) {
                float4 buf[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) buf[b] = SampleIn<TIN>::load4(x + c * kFrame + k0 + 4 * b);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float4 v = buf[b];
                    v.x = one(v.x); v.y = one(v.y); v.z = one(v.z); v.w = one(v.w);
                    *reinterpret_cast<float4 *>(y + c * kFrame + k0 + 4 * b) = v;
                }
            }
        } else {
            for (int k = 0; k < kFrame; ++k) y[c * kFrame + k] = one(SampleIn<TIN>::cvt(x[c * kFrame + k]));
        }
    }
    for (size_t k = n_chunks * kFrame; k < n_samples; ++k) y[k] = SampleIn<TIN>::cvt(x[k]);  // tail shorter than a chunk: never framed
}

template <class TIN>
static hipError_t launch_frontend_t(hipStream_t st, const TIN *pcm, size_t S, size_t n_samples, size_t pcm_stride, int gain_on,
                                    float rms_level_ref, float min_gain, float max_gain, int window_size, int band_pass,
                                    BiquadCoef q, float *ring, float *rms, float *gains, float *out, size_t out_stride) {
    const size_t n_chunks = n_samples / kFrame;
    if (S == 0) return hipSuccess;
    const int vec4 = (reinterpret_cast<uintptr_t>(pcm) % (4 * sizeof(TIN)) == 0) && (pcm_stride % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (out_stride % 4 == 0);
    if (n_chunks) {
        const size_t n = S * n_chunks;
        if ((n + 255) / 256 > 0x7fffffffULL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(chunk_rms_kernel<TIN>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pcm, S, n_chunks, pcm_stride, vec4, rms);
        if (gain_on) {
            const size_t ring_lds = (size_t)window_size * 64 * sizeof(float);
            const int in_lds = ring_lds <= 48 * 1024;
            hipLaunchKernelGGL(gain_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), in_lds ? ring_lds : 0, st, rms, S, n_chunks,
                               rms_level_ref, min_gain, max_gain, window_size, in_lds, ring, gains);
        }
    }
    hipLaunchKernelGGL(apply_filters_kernel<TIN>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, pcm, S, n_samples, n_chunks,
                       pcm_stride, gain_on ? gains : nullptr, band_pass, q, vec4, out, out_stride);
    return hipGetLastError();
}

hipError_t launch_frontend(hipStream_t st, const void *pcm, int fmt, size_t S, size_t n_samples, size_t pcm_stride, int gain_on,
                           float rms_level_ref, float min_gain, float max_gain, int window_size, int band_pass, float a0,
                           float a1, float a2, float b1, float b2, float *ring, float *rms, float *gains, float *out,
                           size_t out_stride) {
    BiquadCoef q{a0, a1, a2, b1, b2};
#define RP_FE(T) launch_frontend_t<T>(st, static_cast<const T *>(pcm), S, n_samples, pcm_stride, gain_on, rms_level_ref, min_gain, \
                                      max_gain, window_size, band_pass, q, ring, rms, gains, out, out_stride)
    switch (fmt) {
    case 0: return RP_FE(int8_t);
    case 1: return RP_FE(int16_t);
    case 2: return RP_FE(int32_t);
    case 3: return RP_FE(float);
    }
#undef RP_FE
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------ synth
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_stream, size_t S, size_t n_samples,
                                                    size_t pcm_stride, float *__restrict__ pcm) {
    const size_t total = S * n_samples;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t s = idx / n_samples, i = idx - s * n_samples;
        uint64_t h = splitmix64(seed ^ (((first_stream + s) << 32) + (uint64_t)i));
        pcm[s * pcm_stride + i] = (float)(h >> 40) / 16777216.f - 0.5f;
    }
}

hipError_t launch_synth(hipStream_t st, uint64_t seed, uint64_t first_stream, size_t S, size_t n_samples,
                        size_t pcm_stride, float *pcm) {
    if (S == 0 || n_samples == 0) return hipSuccess;
    size_t total = S * n_samples;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, first_stream, S, n_samples,
                       pcm_stride, pcm);
    return hipGetLastError();
}

// -------------------------------------------------------------------------- MLP
// Linear (x.W^T + b, W [out][in]) + optional ReLU, f32, k-ordered accumulation like
// candle's CPU gemm restated in the oracle.  One wave per (row, 64 outputs) tile with
// the input row staged in LDS.  (Round-1 correctness path; DESIGN.md lists the MFMA
// bf16 path for BASELINE config 5 as next.)
__global__ __launch_bounds__(64) void mlp_layer_kernel(const float *__restrict__ x, size_t B, int in, int on,
                                                       const float *__restrict__ Wt, const float *__restrict__ bias,
                                                       int relu, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xr = reinterpret_cast<float *>(smem);
    const size_t b = blockIdx.x;
    const int o = blockIdx.y * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < in; i += 64) xr[i] = x[b * in + i];
    __syncthreads();
    if (o >= on) return;
    const float *w = Wt + (size_t)o * in;
    float s = 0.f;
    for (int i = 0; i < in; ++i) s += xr[i] * w[i];
    s += bias[o];
    if (relu && s < 0.f) s = 0.f;
    out[b * on + o] = s;
}

__global__ __launch_bounds__(64) void normalize_windows_kernel(const float *__restrict__ mfcc, size_t first_win,
                                                                size_t n_win, int L, int K, float *__restrict__ x) {
    const size_t w = blockIdx.x;
    const float *src = mfcc + (first_win + w) * K;
    float *dst = x + w * (size_t)L * K;
    for (int k = threadIdx.x; k < K; k += 64) {
        float sum = 0.f;
        for (int i = 0; i < L; ++i) sum += src[(size_t)i * K + k];
        for (int i = 0; i < L; ++i) dst[(size_t)i * K + k] = src[(size_t)i * K + k] - sum / (float)L;
    }
}

hipError_t launch_normalize_windows(hipStream_t st, const float *mfcc, size_t first_win, size_t n_win, int L, int K,
                                    float *x) {
    if (n_win == 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_windows_kernel, dim3((unsigned)n_win), dim3(64), 0, st, mfcc, first_win, n_win, L, K, x);
    return hipGetLastError();
}

hipError_t launch_mlp(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                      float *const *Bv, float *scratch0, float *scratch1, float *out) {
    if (B == 0) return hipSuccess;
    const float *cur = x;
    float *bufs[2] = {scratch0, scratch1};
    for (int l = 0; l < n_layers; ++l) {
        float *dst = (l + 1 == n_layers) ? out : bufs[l & 1];
        dim3 grid((unsigned)B, (unsigned)((dims[l + 1] + 63) / 64));
        size_t lds = (size_t)dims[l] * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(mlp_layer_kernel, grid, dim3(64), lds, st, cur, B, dims[l], dims[l + 1], W[l], Bv[l],
                           l + 1 < n_layers ? 1 : 0, dst);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        cur = dst;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------ MLP training
// WakewordModelTrain's loop (src/wakewords/nn/wakeword_model_train.rs:204-209): full-batch forward,
// log_softmax + nll (mean over the batch), backward, plain SGD.  The matrices are tiny (tens of recordings x a few
// thousand features): one thread per result element, reductions along the batch / the layer width in a loop.
__global__ __launch_bounds__(64) void train_forward_kernel(const float *__restrict__ x, size_t B, int in, int on,
                                                           const float *__restrict__ W, const float *__restrict__ bias, int relu,
                                                           float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xr = reinterpret_cast<float *>(smem);
    const size_t b = blockIdx.x;
    const int o = blockIdx.y * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < in; i += 64) xr[i] = x[b * in + i];
    __syncthreads();
    if (o >= on) return;
    const float *w = W + (size_t)o * in;
    float s = 0.f;
    for (int i = 0; i < in; ++i) s += xr[i] * w[i];
    s += bias[o];
    if (relu && s < 0.f) s = 0.f;
    out[b * on + o] = s;
}

// per row: log_softmax (x - max - ln(sum exp(x - max))), loss_row = -log_sm[label], dlogits = (softmax - onehot) / B
__global__ __launch_bounds__(64) void train_softmax_grad_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels,
                                                                size_t B, int C, float *__restrict__ dz, float *__restrict__ loss_rows) {
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float *x = logits + b * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float lse = logf(se);
    const int lab = labels[b];
    for (int c = 0; c < C; ++c) {
        const float lsm = (x[c] - mx) - lse;
        if (c == lab) loss_rows[b] = -lsm;
        dz[b * C + c] = (expf(lsm) - (c == lab ? 1.f : 0.f)) / (float)B;
    }
}

// dZprev[b][i] = A_prev[b][i] > 0 ? sum_o dZ[b][o] * W[o][i] : 0     (ReLU backward through the layer's input)
__global__ __launch_bounds__(256) void train_backprop_kernel(const float *__restrict__ dz, const float *__restrict__ W,
                                                             const float *__restrict__ a_prev, size_t B, int in, int on,
                                                             float *__restrict__ dz_prev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= in) return;
    float s = 0.f;
    for (int o = 0; o < on; ++o) s += dz[b * on + o] * W[(size_t)o * in + i];
    dz_prev[b * in + i] = a_prev[b * in + i] > 0.f ? s : 0.f;
}

// SGD step of one layer: W[o][i] -= lr * sum_b dZ[b][o] * A_in[b][i];  bias[o] -= lr * sum_b dZ[b][o]
__global__ __launch_bounds__(256) void train_update_kernel(const float *__restrict__ dz, const float *__restrict__ a_in, size_t B,
                                                           int in, int on, float lr, float *__restrict__ W, float *__restrict__ bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (i > in) return;  // i == in: the bias column
    float g = 0.f;
    if (i < in) {
        for (size_t b = 0; b < B; ++b) g += dz[b * on + o] * a_in[b * in + i];
        W[(size_t)o * in + i] = W[(size_t)o * in + i] - g * lr;
    } else {
        for (size_t b = 0; b < B; ++b) g += dz[b * on + o];
        bias[o] = bias[o] - g * lr;
    }
}

hipError_t launch_train_forward(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                                float *const *Bv, float *const *act) {
    if (B == 0) return hipSuccess;
    const float *cur = x;
    for (int l = 0; l < n_layers; ++l) {
        dim3 grid((unsigned)B, (unsigned)((dims[l + 1] + 63) / 64));
        const size_t lds = (size_t)dims[l] * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(train_forward_kernel, grid, dim3(64), lds, st, cur, B, dims[l], dims[l + 1], W[l], Bv[l],
                           l + 1 < n_layers ? 1 : 0, act[l]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        cur = act[l];
    }
    return hipSuccess;
}

// act[l] = output of layer l (post-ReLU for hidden layers, logits for the last); dz[l] same shapes
hipError_t launch_train_step(hipStream_t st, const float *x, const int32_t *labels, size_t B, int n_layers, const int *dims,
                             float *const *W, float *const *Bv, float *const *act, float *const *dz, float lr, float *loss_rows) {
    if (B == 0) return hipSuccess;
    hipError_t e = launch_train_forward(st, x, B, n_layers, dims, W, Bv, act);
    if (e != hipSuccess) return e;
    const int C = dims[n_layers];
    hipLaunchKernelGGL(train_softmax_grad_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, act[n_layers - 1], labels, B, C,
                       dz[n_layers - 1], loss_rows);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    for (int l = n_layers - 1; l >= 0; --l) {
        const int in = dims[l], on = dims[l + 1];
        const float *a_in = l == 0 ? x : act[l - 1];
        if (l > 0) {  // through W_l as it was in the forward pass, before its own update
            hipLaunchKernelGGL(train_backprop_kernel, dim3((unsigned)((in + 255) / 256), (unsigned)B), dim3(256), 0, st, dz[l], W[l],
                               act[l - 1], B, in, on, dz[l - 1]);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(train_update_kernel, dim3((unsigned)((in + 1 + 255) / 256), (unsigned)on), dim3(256), 0, st, dz[l], a_in, B, in,
                           on, lr, W[l], Bv[l]);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------ MLP on MFMA
// One workgroup = 8 waves = 128 rows; one wave = one 16-row tile x all layer-1 outputs (NT
// 16-column tiles).  The layer-1 weights are walked in k-groups of 128: the group's [16*NT][128]
// slice is staged once per workgroup in LDS and shared by the 8 waves (reading it per wave from L2
// cost more than the HBM stream itself: 0.39 -> 0.18 ms when removed), rows stream from HBM in
// the MFMA A-operand layout (16 rows x 64 B per instruction; 5.5 TB/s measured on its own).
//  f32 variant:  v_mfma_f32_16x16x4_f32, exact f32 (each output is a k-ordered fmaf chain).  A lane
//                loads 16 bytes of its row per 16-k block and feeds component j to MFMA step j; the
//                weight lane does the same, so both sides agree on the (permuted) k order.
//  bf16 variant: v_mfma_f32_16x16x32_bf16, inputs rounded to bf16 (RNE) in registers, f32 accumulate.
// The tail layers (<= 130 x 32 weights) run per row from LDS.  HBM-bound by construction: 4*in bytes
// per row against 2*in*N1 flops (12 480 B/row, ceiling 0.64 G rows/s at 8 TB/s).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kMlpWaves = 8;
constexpr int kMlpRowsPerWave = 16;
constexpr int kMlpKG = 128;  // k-group staged per step

__host__ __device__ constexpr int mlp_wpitch_f32() { return kMlpKG + 4; }   // floats per staged weight row
__host__ __device__ constexpr int mlp_wpitch_bf16() { return kMlpKG + 8; }  // bf16 per staged weight row

template <int NT, int PREC>
__global__ __launch_bounds__(64 * kMlpWaves) void mlp_mfma_kernel(
    const float *__restrict__ x, size_t B, int in, int kpad, const float *__restrict__ w1f,
    const __bf16 *__restrict__ w1h, const float *__restrict__ b1, const float *__restrict__ tail, int tail_floats,
    int n_layers, int d1, int d2, int d3, int d4, int h2w, int wbuf_floats, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int N1P = 16 * NT;
    float *tl = reinterpret_cast<float *>(smem);                       // tail weights
    float *wbuf = tl + ((tail_floats + 3) & ~3);                       // staged weight group; later h1 [waves][16][N1P+1]
    float *h2_all = wbuf + wbuf_floats;                                // [waves][16][h2w]
    for (int i = threadIdx.x; i < tail_floats; i += blockDim.x) tl[i] = tail[i];

    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int li = l & 15, lk = l >> 4;
    const size_t row0 = ((size_t)blockIdx.x * kMlpWaves + wave) * kMlpRowsPerWave;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t r = row0 + li;
    if (r >= B) r = B - 1;  // rows past the end recompute the last row; their results are dropped
    const float *xr = x + r * in;

    // staged weight groups are double buffered when they fit (NT <= 2): the next group's global
    // loads are issued before this group's MFMAs and land in the other buffer, one barrier per group
    constexpr bool DB = NT <= 2;
    constexpr int PF = mlp_wpitch_f32(), PH = mlp_wpitch_bf16();
    constexpr int NV = PREC == kMlpF32 ? (N1P * (kMlpKG / 4) + 64 * kMlpWaves - 1) / (64 * kMlpWaves)
                                       : (N1P * (kMlpKG / 8) + 64 * kMlpWaves - 1) / (64 * kMlpWaves);
    const int half = DB ? wbuf_floats / 2 : 0;
    float4 wreg[NV];  // one 16-byte piece = 4 f32 or 8 bf16
    auto wload = [&](int g) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kMlpWaves;
            if (PREC == kMlpF32) {
                const int o = i / (kMlpKG / 4), c = i - o * (kMlpKG / 4);
                if (o < N1P) wreg[v] = *reinterpret_cast<const float4 *>(w1f + (size_t)o * kpad + g * kMlpKG + 4 * c);
            } else {
                const int o = i / (kMlpKG / 8), c = i - o * (kMlpKG / 8);
                if (o < N1P) wreg[v] = *reinterpret_cast<const float4 *>(w1h + (size_t)o * kpad + g * kMlpKG + 8 * c);
            }
        }
    };
    auto wstore = [&](float *dstbuf) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kMlpWaves;
            if (PREC == kMlpF32) {
                const int o = i / (kMlpKG / 4), c = i - o * (kMlpKG / 4);
                if (o < N1P) *reinterpret_cast<float4 *>(dstbuf + o * PF + 4 * c) = wreg[v];
            } else {
                const int o = i / (kMlpKG / 8), c = i - o * (kMlpKG / 8);
                if (o < N1P) *reinterpret_cast<float4 *>(reinterpret_cast<__bf16 *>(dstbuf) + o * PH + 8 * c) = wreg[v];
            }
        }
    };
    const int ngrp = kpad / kMlpKG;
    wload(0);
    wstore(wbuf);
    __syncthreads();  // group 0 staged (and the tail weights landed)
    for (int g = 0; g < ngrp; ++g) {
        const int kg = g * kMlpKG;
        const float *cur = wbuf + ((DB && (g & 1)) ? half : 0);
        if (g + 1 < ngrp) wload(g + 1);
        if (PREC == kMlpF32) {
            constexpr int KU = kMlpKG / 16;
            float4 a[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k0 = kg + 16 * u + 4 * lk;  // rows are 16-byte aligned: in % 4 == 0 (checked by the launcher)
                a[u] = (k0 + 3 < in) ? *reinterpret_cast<const float4 *>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 b = *reinterpret_cast<const float4 *>(cur + (16 * n + li) * PF + 16 * u + 4 * lk);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b.w, acc[n], 0, 0, 0);
                }
        } else {
            constexpr int KU = kMlpKG / 32;
            const __bf16 *wb = reinterpret_cast<const __bf16 *>(cur);
            float4 lo[KU], hi[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k0 = kg + 32 * u + 8 * lk;
                lo[u] = (k0 + 3 < in) ? *reinterpret_cast<const float4 *>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                hi[u] = (k0 + 7 < in) ? *reinterpret_cast<const float4 *>(xr + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                bf16x8 a;
                a[0] = (__bf16)lo[u].x; a[1] = (__bf16)lo[u].y; a[2] = (__bf16)lo[u].z; a[3] = (__bf16)lo[u].w;
                a[4] = (__bf16)hi[u].x; a[5] = (__bf16)hi[u].y; a[6] = (__bf16)hi[u].z; a[7] = (__bf16)hi[u].w;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wb + (16 * n + li) * PH + 32 * u + 8 * lk);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
                }
            }
        }
        if (g + 1 < ngrp) {
            if (!DB) __syncthreads();  // single buffer: everyone must be done reading before the overwrite
            wstore(wbuf + ((DB && !(g & 1)) ? half : 0));
        }
        __syncthreads();
    }
    // every wave is past the last barrier, i.e. done with the staged weights: the buffer becomes h1
    // ---- layer-1 bias (+ReLU) -> LDS, C/D layout: col = lane&15, row = (lane>>4)*4 + reg
    float *h1 = wbuf + wave * kMlpRowsPerWave * (N1P + 1);
    const bool relu1 = n_layers > 1;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[n][e] + b1[16 * n + li];
            if (relu1 && v < 0.f) v = 0.f;
            h1[(4 * lk + e) * (N1P + 1) + 16 * n + li] = v;
        }
    wave_lds_sync();
    // ---- tail layers: lane = (row l&15, output phase l>>4), outputs strided by 4 over the phases
    {
        const int rr = l & 15, ph = l >> 4;
        const bool row_ok = row0 + rr < B;
        const float *hin = h1 + rr * (N1P + 1);
        float *h2 = h2_all + (wave * kMlpRowsPerWave + rr) * h2w;
        const int dd[5] = {in, d1, d2, d3, d4};
        const float *wp = tl;
        int cur_in = d1;
        float *dst = out + (row0 + rr) * (size_t)dd[n_layers];
        if (n_layers == 1 && row_ok)
            for (int o = ph; o < d1; o += 4) dst[o] = hin[o];
        for (int layer = 1; layer < n_layers; ++layer) {
            const int on = dd[layer + 1];
            const bool last = layer + 1 == n_layers;
            for (int o = ph; o < on; o += 4) {
                const float *wr = wp + (size_t)o * cur_in;
                float s0 = 0.f, s1 = 0.f;
                int i = 0;
                for (; i + 1 < cur_in; i += 2) { s0 = fmaf(hin[i], wr[i], s0); s1 = fmaf(hin[i + 1], wr[i + 1], s1); }
                if (i < cur_in) s0 = fmaf(hin[i], wr[i], s0);
                float sacc = (s0 + s1) + wp[(size_t)on * cur_in + o];
                if (!last && sacc < 0.f) sacc = 0.f;
                if (last) { if (row_ok) dst[o] = sacc; } else h2[o] = sacc;
            }
            wave_lds_sync();  // the hidden layer is complete before anyone reads it
            wp += (size_t)on * cur_in + on;
            cur_in = on;
            hin = h2;  // n_layers <= 3: at most one hidden tail layer
        }
    }
}

template <int NT>
static hipError_t launch_mlp_nt(hipStream_t st, const MlpDev &m, const float *x, size_t B, int precision, float *out) {
    const size_t rows_per_block = (size_t)kMlpWaves * kMlpRowsPerWave;
    const size_t blocks = (B + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    int h2w = 1;
    for (int l2 = 2; l2 < m.n_layers; ++l2) h2w = m.dims[l2] + 1 > h2w ? m.dims[l2] + 1 : h2w;
    h2w |= 1;
    size_t wbuf = (size_t)16 * NT * mlp_wpitch_f32() * (NT <= 2 ? 2 : 1);       // f32 group(s) (the bf16 ones are smaller)
    const size_t h1 = (size_t)kMlpWaves * kMlpRowsPerWave * (16 * NT + 1);
    if (h1 > wbuf) wbuf = h1;
    wbuf = (wbuf + 3) & ~(size_t)3;
    const size_t lds = ((size_t)((m.tail_floats + 3) & ~3) + wbuf + (size_t)kMlpWaves * kMlpRowsPerWave * h2w) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpBf16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpF32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (precision == kMlpBf16)
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpBf16>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1h), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out);
    else
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpF32>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1h), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out);
    return hipGetLastError();
}

hipError_t launch_mlp_mfma(hipStream_t st, const MlpDev &m, const float *x, size_t B, int precision, float *out) {
    if (B == 0) return hipSuccess;
    switch (m.nt) {
    case 1: return launch_mlp_nt<1>(st, m, x, B, precision, out);
    case 2: return launch_mlp_nt<2>(st, m, x, B, precision, out);
    case 5: return launch_mlp_nt<5>(st, m, x, B, precision, out);
    case 9: return launch_mlp_nt<9>(st, m, x, B, precision, out);
    }
    return hipErrorInvalidValue;
}

}  // namespace rp
