// rp_mfcc.hip -- mfcc_kernel: src/mfcc/extractor.rs:60-163 (framing, pre-emphasis, Hamming, DFT-480, mel, ln, DCT),
// 16 lanes per 30 ms frame, and synth_kernel, the benchmark input generator of BASELINE.md §2.
// DESIGN.md §4.1 has the layout and the roofline.
#include "rp_device.h"

#include <stdlib.h>
#include <type_traits>

namespace rp {

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

// floats the mel table takes in LDS: the compact per-lane rows of the sparse kernels (rp_kernels.h, mel_index) or [K1][240]
__host__ __device__ inline size_t mel_lds_floats(int K1, bool sparse) {
    return sparse && K1 == 6 ? (size_t)16 * kMelRowPitch<6> : sparse && K1 == 14 ? (size_t)16 * kMelRowPitch<14>
           : sparse && K1 == 17 ? (size_t)16 * kMelRowPitch<17> : (size_t)K1 * kBins;
}
// The window and the untangling twiddles are kept as one row per lane (lane l only ever needs window values 2(15 n1 + l),
// +1 and twiddles W480^(l + 16 k2)): consecutive entries of a row are read two at a time with 16-byte LDS reads, the row
// pitches (36 / 20 floats) keep the 16 lanes of a frame on distinct bank quads.  24 eight-byte reads per tile become 12
// sixteen-byte ones.
constexpr int kHamRowPitch = 36, kTwRowPitch = 20;
__host__ __device__ inline size_t mfcc_lds_bytes(int K1, bool sparse) {
    size_t f = 16 * kHamRowPitch + 16 * kTwRowPitch + mel_lds_floats(K1, sparse) + (size_t)K1 * K1 + (size_t)kMfccWaves * kMfccFramesPerWave * K1;
    size_t c = (size_t)kMfccWaves * kMfccWaveScratch;  // wave scratch (W240 is only read once per lane, from global)
    return c * sizeof(float2) + f * sizeof(float);
}

// An 8-byte LDS read the backend may NOT fuse with its neighbour into ds_read2_b64: on gfx950 a ds_read2_b64 occupies the
// LDS for 8 cycles against 2 + 2 for the two ds_read_b64 it replaces (MI355X_MICROARCH.md, LDS table), and this kernel
// keeps the CU's LDS busier than its VALUs.  (A volatile access is never merged; LDS volatiles carry no extra waits.)
typedef __attribute__((address_space(3))) const volatile v2f lds_cv2f;
__device__ __forceinline__ v2f lds_read_b64(const v2f *p) { return *(lds_cv2f *)p; }  // explicit LDS address space: a volatile
                                                                                      // generic access would become a flat load
// the same for stores: the backend pairs neighbouring 8-byte stores into ds_write2_b64; kept apart they measured 0.8 % faster
// (7.15-7.21 against 7.22-7.26 ms, three interleaved runs on one box)
typedef __attribute__((address_space(3))) volatile v2f lds_v2f;
__device__ __forceinline__ void lds_write_b64(v2f *p, v2f v) { *(lds_v2f *)p = v; }

template <int K1T> __device__ constexpr bool mel_uses(int f, int k2, bool mirror) {
    if constexpr (K1T == 6 || K1T == 14 || K1T == 17) return mel_touches<K1T>(f, k2, mirror);
    else return true;
}

// RP_ABL_*: ablation builds only (never defined in the product; profiles/HISTORY.md 4.1 has the numbers they gave): NOSTAGE drops the
// staging write of the pre-emphasised samples, NOT1 the transposition between FFT16 and DFT15, NOMIR the mirror exchange
// before the untangle step, NOMEL the mel / log stage -- each leaves the arithmetic in place and produces wrong values.
// K1T: compile-time K+1 (6 and 17 are instantiated), 0 = runtime value.  TIN: input sample type.
// HS (live-stream batches): the stream is [history chunk | new chunks] in two buffers -- samples 0..479 are the
// previous call's last chunk, decoded f32 at hist[s * hist_pitch + i]; sample 480 + i is pcm[s * pcm_stride + i] -- and the
// kernel leaves this call's last chunk (decoded) at hist_out for the next call, so that no staging copy runs in front
// of it.  n_samples counts the history chunk.  Needs VEC4.
#ifdef RP_MFCC_HAM_REGS
#define RP_MFCC_MIN_WGS(HS) 3
#else
#define RP_MFCC_MIN_WGS(HS) ((HS) ? 3 : 4)
#endif
template <bool VEC4, int K1T, class TIN, bool HS = false>
__global__ __launch_bounds__(kMfccThreads, RP_MFCC_MIN_WGS(HS)) void mfcc_kernel(
    const TIN *__restrict__ pcm, size_t n_samples, size_t pcm_stride, unsigned tiles_per_stream, size_t total_tiles,
    size_t first_frame, size_t n_frames, size_t out_frame_pitch, int K1rt, const float *__restrict__ g_ham,
    const float2 *__restrict__ g_tw240, const float2 *__restrict__ g_tw480, const float *__restrict__ g_fb,
    const float *__restrict__ g_dct, float *__restrict__ out, float *__restrict__ out2,
    const float *__restrict__ hist = nullptr, size_t hist_pitch = 0, float *__restrict__ hist_out = nullptr) {
    static_assert(!HS || VEC4, "the history split is implemented for 4-sample loads");
    const int K1 = K1T > 0 ? K1T : K1rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f *scr_all = reinterpret_cast<v2f *>(smem);               // [waves][4][240]
    float *tw480 = reinterpret_cast<float *>(scr_all + kMfccWaves * kMfccWaveScratch);  // [16 lanes][8 k2] float2, row pitch kTwRowPitch floats
    float *ham = tw480 + 16 * kTwRowPitch;                      // [16 lanes][16 n1] float2, row pitch kHamRowPitch floats
    float *fb = ham + 16 * kHamRowPitch;                        // [K1][240], or the compact rows [16][kMelRowPitch] (K1T > 0)
    const int fb_floats = (int)mel_lds_floats(K1, K1T > 0);
    float *dct = fb + fb_floats;                                // [K1][K1]
    float *lgb_all = dct + K1 * K1;                             // [waves][4][K1]

    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * 32; i += kMfccThreads) {   // lane row (i >> 5), entry n1 = (i & 31) >> 1, component i & 1
        const int ll = i >> 5, n1 = (i & 31) >> 1;
        ham[ll * kHamRowPitch + (i & 31)] = ll < 15 ? g_ham[2 * (15 * n1 + ll) + (i & 1)] : 0.f;
    }
    for (int i = tid; i < 16 * 8; i += kMfccThreads) {
        const int ll = i >> 3, k2 = i & 7;
        const float2 w = g_tw480[ll + 16 * k2];   // l + 16 k2 <= 127 < 240
        tw480[ll * kTwRowPitch + 2 * k2] = w.x; tw480[ll * kTwRowPitch + 2 * k2 + 1] = w.y;
    }
    for (int i = tid; i < fb_floats; i += kMfccThreads) fb[i] = g_fb[i];
    for (int i = tid; i < K1 * K1; i += kMfccThreads) dct[i] = g_dct[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave-uniform: tile indices stay scalar
    const int grp = lane >> 4, l = lane & 15;
    v2f *scr = scr_all + wave * kMfccWaveScratch;
    float *ypre = reinterpret_cast<float *>(scr);  // [960] pre-emphasised samples, dead before scr is written
    v2f *my = scr + grp * 240;
    float *lgb = lgb_all + (wave * kMfccFramesPerWave + grp) * K1;
    const int K = K1 - 1;
    // per-lane base pointers: every LDS access below is base[compile-time offset]
    const int l15 = l < 15 ? l : 0;
    const v2f *ysrc = reinterpret_cast<const v2f *>(ypre + grp * kShift) + l15;  // z[15*n1 + n2] -> +15*n1
    const f32x4 *hrow = reinterpret_cast<const f32x4 *>(ham + l * kHamRowPitch);     // (n1, n1 + 1) pairs
    v2f *t1dst = my + l;              // [k1*15 + l]
    const v2f *t1src = my + l * 15;   // [l*15 + n2]
    v2f *zdst = my + l;               // [l + 16*k2]
    const v2f *zmir = my + 240 - l;  // Z[240-k] = zmir[-16*k2]; k == 0 pairs with itself (handled below)
    const f32x4 *wrow = reinterpret_cast<const f32x4 *>(tw480 + l * kTwRowPitch);   // (k2, k2 + 1) pairs
    const float *fbk = fb + l;                       // bins k = l + 16*k2
    const float *fbm = fb + 240 - l;                 // bins 240-k = fbm[-16*k2] (k == 0: weight unused, power forced to 0)
    // twiddles W240^{n2*k1}, fixed per lane across tiles, k1 = c + 4d kept at register 4c+d
    v2f twl[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) { const float2 w = g_tw240[l15 * (c + 4 * d)]; twl[4 * c + d] = (v2f){w.x, w.y}; }

#ifdef RP_MFCC_HAM_REGS
    // A/B: the lane's Hamming row in registers for the whole launch (32 VGPRs: three waves per SIMD instead of four)
    f32x4 hreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hreg[i] = hrow[i];
#endif
    // (stream, tile) of this wave's tile, advanced by the grid stride without a division per tile
    const size_t wave_stride = (size_t)gridDim.x * kMfccWaves;
    const size_t stride_s = wave_stride / tiles_per_stream;
    const unsigned stride_t = (unsigned)(wave_stride - stride_s * tiles_per_stream);
    size_t wt = (size_t)blockIdx.x * kMfccWaves + wave;
    size_t s = wt / tiles_per_stream;
    unsigned tile = (unsigned)(wt - s * tiles_per_stream);
    // The samples of a tile are fetched one tile ahead (raw, decoded when they are used): the HBM round trip runs under
    // the previous tile's FFT instead of in front of this one's.  All loads are unconditional (clamped index).
    using Raw4 = typename SampleIn<TIN>::Raw4;
    constexpr int NV = VEC4 ? 4 : 1, NS = VEC4 ? 4 : kMfccStage / 64;
    Raw4 cur4[NV];
    TIN cur1[NS], prv[NS];
    f32x4 hcur[HS ? 4 : 1];  // HS: the groups of a stream's first tile that lie in the history chunk (f32x4: a float4
                              // struct array would be copied through scratch)
    const size_t last = n_samples - 1;
    // position of 4-sample group `it` of this lane in the stream (clamped to the last whole group)
    auto group_pos = [&](unsigned ftile, int it) {
        const size_t base = (first_frame + (size_t)ftile * kMfccFramesPerWave + 1) * kShift;
        const int q = it * 64 + lane;                    // 4-sample group inside the 960-sample tile
        const size_t g = base + 4 * (size_t)(q < 240 ? q : 239);
        return g + 3 <= last ? g : (last - 3) & ~(size_t)3;
    };
    // byte offsets of this lane's four 4-sample groups inside a tile (the last 16 lanes of the fourth round repeat group 239)
    constexpr unsigned kGroupBytes = 4 * sizeof(TIN);
    const unsigned goff_lo = (unsigned)lane * kGroupBytes;
    const unsigned goff_3 = (unsigned)(192 + lane < 240 ? 192 + lane : 239) * kGroupBytes;
    auto fetch = [&](size_t fs, unsigned ftile) {
        const TIN *x = pcm + fs * pcm_stride;
        const size_t base = (first_frame + (size_t)ftile * kMfccFramesPerWave + 1) * kShift;
        if (VEC4 && !HS) {
            // One wave-uniform 64-bit base per group round and a 32-bit lane offset (per-lane 64-bit positions cost ~80
            // VALU instructions per tile).  Positions are clamped to the last whole group of the stream -- only a ragged
            // last tile reaches it: lim = byte offset of that group from the tile's first sample, applied as a scalar
            // cap on the round's base and one v_min on the lane offset.
            const char *sp = reinterpret_cast<const char *>(x + base);
            const size_t room = n_samples - base;                       // >= 4: a tile starts at least one frame before the end
            const unsigned lim = (unsigned)((room < (size_t)kMfccStage ? room : (size_t)kMfccStage) / 4 - 1) * kGroupBytes;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const unsigned c = (unsigned)it * 64u * kGroupBytes;
                const unsigned cb = c < lim ? c : lim, rest = lim - cb;   // scalar
                const unsigned t = goff_lo < rest ? goff_lo : rest;
                cur4[it] = SampleIn<TIN>::ldraw(reinterpret_cast<const TIN *>(sp + cb + (size_t)t));
            }
            const unsigned t3 = goff_3 < lim ? goff_3 : lim;
            cur4[3] = SampleIn<TIN>::ldraw(reinterpret_cast<const TIN *>(sp + (size_t)t3));
        } else if (VEC4) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const size_t g = group_pos(ftile, it);
                if (HS) {  // new data starts at sample 480; only a stream's first tile reaches below it
                    const size_t gn = g >= kFrame ? g - kFrame : 0;
                    cur4[it] = SampleIn<TIN>::ldraw(x + gn);
                } else {
                    cur4[it] = SampleIn<TIN>::ldraw(x + g);
                }
            }
            if (HS && ftile == 0 && first_frame == 0) {
                const float *h = hist + fs * hist_pitch;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const size_t g = group_pos(ftile, it), gh = g < kFrame ? g : kFrame - 4;
                    hcur[it] = *reinterpret_cast<const f32x4 *>(h + gh);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < kMfccStage / 64; ++it) {
                size_t g = base + it * 64 + lane;
                g = g <= last ? g : last;
                cur1[it] = x[g];
                prv[it] = x[g - 1];
            }
        }
    };
    if (wt < total_tiles) fetch(s, tile);
    while (wt < total_tiles) {
        const size_t j0 = first_frame + (size_t)tile * kMfccFramesPerWave;
        // pre_emphasis, src/mfcc/extractor.rs:87-97: previous sample is 0 at the start of EVERY shift.
        if (VEC4) {
            float carry = 0.f;  // last sample of the previous 64 groups
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + lane;
                float4 c = SampleIn<TIN>::cvt4(cur4[it]);
                if (HS) {
                    const size_t g = group_pos(tile, it);
                    if (tile == 0 && first_frame == 0) {
                        if (g < kFrame) c = make_float4(hcur[it].x, hcur[it].y, hcur[it].z, hcur[it].w);
                    }
                    // this call's last chunk is the next call's history (every sample of it is loaded by the stream's
                    // last tile; groups loaded twice store the same values)
                    if (g >= n_samples - kFrame) *reinterpret_cast<float4 *>(hist_out + s * hist_pitch + (g - (n_samples - kFrame))) = c;
                }
                // the sample in front of a group is the left neighbour's last one (groups are consecutive along the lanes):
                // one DPP wave shift instead of a second global load per group
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c.w), 0x138, 0xf, 0xf, false));  // wave_shr:1
                const float pv = lane == 0 ? carry : left;
                carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w), 63));
                const float p0 = (q % (kShift / 4) == 0) ? 0.f : pv;
                float4 y;
                y.x = c.x - 0.97f * p0;  // separate multiply and subtract, like the reference
                y.y = c.y - 0.97f * c.x;
                y.z = c.z - 0.97f * c.y;
                y.w = c.w - 0.97f * c.z;
#ifndef RP_ABL_NOSTAGE
                if (q < 240) *reinterpret_cast<float4 *>(ypre + 4 * q) = y;
#else
                asm volatile("" :: "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w));
#endif
            }
        } else {
#pragma unroll
            for (int it = 0; it < kMfccStage / 64; ++it) {
                const int i = it * 64 + lane;
                ypre[i] = SampleIn<TIN>::cvt(cur1[it]) - 0.97f * ((i % kShift == 0) ? 0.f : SampleIn<TIN>::cvt(prv[it]));
            }
        }
        const size_t s_now = s;
        wt += wave_stride; s += stride_s; tile += stride_t;
        if (tile >= tiles_per_stream) { tile -= tiles_per_stream; ++s; }
        if (wt < total_tiles) fetch(s, tile);
        wave_lds_sync();
        // ---- step 1: lane n2=l (<15): FFT16 over n1 of z[15*n1 + n2], z[n] = (y[2n], y[2n+1]) * hamming
        v2f v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; n1 += 2) {
#ifdef RP_MFCC_HAM_REGS
            const f32x4 h = hreg[n1 / 2];
#else
            const f32x4 h = hrow[n1 / 2];
#endif
            v[n1] = lds_read_b64(ysrc + 15 * n1) * (v2f){h.x, h.y};
            v[n1 + 1] = lds_read_b64(ysrc + 15 * (n1 + 1)) * (v2f){h.z, h.w};
        }
        wave_lds_sync();  // every lane has its samples in registers: the scratch may now overwrite them
        fft16(v);
#ifndef RP_ABL_NOT1
        if (l < 15) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) lds_write_b64(t1dst + (c + 4 * d) * 15, cmul(v[4 * c + d], twl[4 * c + d]));  // W240^{n2*k1}
        }
#else
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = cmul(v[c], twl[c]);
#endif
        wave_lds_sync();
        // ---- step 3: lane k1=l: DFT15 over n2 -> Z[k1 + 16*k2]
        v2f z[15];
        {
            v2f u[15];
#pragma unroll
#ifndef RP_ABL_NOT1
            for (int n2 = 0; n2 < 15; ++n2) u[n2] = lds_read_b64(t1src + n2);
#else
            for (int n2 = 0; n2 < 15; ++n2) u[n2] = v[n2];
#endif
            dft15(u, z);
        }
        wave_lds_sync();
        // only the mirrors are read back: bin 240-k of lane l, k2 = 0..7, is Z[(16-l) + 16(14-k2)] (lane 0: Z[16(15-k2)]),
        // i.e. registers 7..14 of the partner lane
#if !defined(RP_ABL_NOMIR) && !defined(RP_MFCC_DPP_MIRROR)
#pragma unroll
        for (int k2 = 7; k2 < 15; ++k2) lds_write_b64(zdst + 16 * k2, z[k2]);
#endif
#ifndef RP_MFCC_DPP_MIRROR
        wave_lds_sync();
#endif
        // ---- untangle the two interleaved real sequences, bins k = l + 16*k2 <= 120 together with their
        // mirrors 240-k (X[240-k] = conj(E - W480^k O) shares E, O and the twiddle product with X[k]).
        // Everything is kept at twice its value; the factor 4 on the powers is removed before the log.
        float Pk[8], Pm[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const int k = l + 16 * k2;
            const v2f a = z[k2];
#if defined(RP_MFCC_DPP_MIRROR)
            // A/B: the partner's register by two DPP moves (row_mirror: lane l <- 15 - l, then row_ror:1: lane l <- l - 1, together
            // lane l <- 16 - l) instead of the LDS exchange; lane 0 pairs with its own registers
            auto mir = [](float x) {
                const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true);
                return __int_as_float(__builtin_amdgcn_update_dpp(0, t, 0x121, 0xf, 0xf, true));
            };
            const v2f own = k2 == 0 ? a : z[15 - k2];
            const v2f far = (v2f){mir(z[14 - k2].x), mir(z[14 - k2].y)};
            const v2f b = l == 0 ? own : far;
#elif !defined(RP_ABL_NOMIR)
            const v2f b = (k2 == 0 && l == 0) ? a : lds_read_b64(zmir - 16 * k2);
#else
            const v2f b = z[14 - k2];
#endif
            const v2f e = add_conj(a, b);                        // 2E = a + conj(b)
            const v2f o = mi_sub_conj(a, b);                     // 2O = -i (a - conj(b))
            const f32x4 w2 = wrow[k2 / 2];
            const v2f t = cmul((k2 & 1) ? (v2f){w2.z, w2.w} : (v2f){w2.x, w2.y}, o);
            // X[k] = e + t and X[240-k] = e - t, kept as (re, re) and (im, im): the two powers come out of ONE packed multiply
            // and ONE packed multiply-add -- per half the same fmaf(re, re, im * im) as before
            const v2f re = re_sum_diff(e, t), im = im_sum_diff(e, t);
            const v2f pw = __builtin_elementwise_fma(re, re, im * im);   // (4 |X[k]|^2, 4 |X[240-k]|^2)
            Pk[k2] = k <= 120 ? pw.x : 0.f;                      // k > 120 is formed by the mirror lane
            Pm[k2] = (k >= 1 && k < 120) ? pw.y : 0.f;           // bin 240 is not used; 120 is its own mirror
        }
        // ---- mel filterbank (rows of 8 filters per pass) + ln, src/mfcc/extractor.rs:121-145.  For mfcc_size 5 and 16
        // the triangles are known at compile time (mel_touches): a filter is only accumulated over the 16-bin groups
        // it can be non-zero in -- 31 multiply-adds and weight loads instead of 96 (41 instead of 272), bit-identical
        // (the skipped terms are exact + 0.0; the table is checked against the mask on upload).
#ifdef RP_ABL_NOMEL
        {
            float sacc = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) sacc += Pk[k2] + Pm[k2];
            if (l < 8 && l < K1) lgb[l] = sacc;
        }
#endif
        {
            // one pass = 8 filters; i0v is the pass's first filter: a compile-time constant for the sparse kernels (K1T > 0),
            // a run-time value for the general one
            auto mel_pass = [&](auto i0v) {
                const int i0 = (int)i0v;
                float acc[8];
                if constexpr (K1T > 0) {
                    constexpr int I0 = decltype(i0v)::value;
                    // this lane's weights of the pass, 16 bytes per LDS read (its row holds them in accumulation order)
                    constexpr int e0 = mel_index<K1T>(I0 < K1T ? I0 : K1T, 0, false), e1 = mel_index<K1T>(I0 + 8 < K1T ? I0 + 8 : K1T, 0, false);
                    constexpr int v0 = e0 / 4, nv = (e1 + 3) / 4 - v0;
                    f32x4 wv[nv > 0 ? nv : 1];
                    const f32x4 *row = reinterpret_cast<const f32x4 *>(fb + l * kMelRowPitch<K1T>) + v0;
#pragma unroll
                    for (int v = 0; v < nv; ++v) wv[v] = row[v];
                    int e = e0 - 4 * v0;  // running position in wv: a compile-time constant at every use once the loops are unrolled
#pragma unroll
                    for (int ii = 0; ii < 8; ++ii) {
                        acc[ii] = 0.f;
                        if (I0 + ii < K1T) {
#pragma unroll
                            for (int k2 = 0; k2 < 8; ++k2) {
                                if (mel_touches<K1T>(I0 + ii, k2, false)) { acc[ii] = fmaf(Pk[k2], wv[e / 4][e % 4], acc[ii]); ++e; }
                                if (mel_touches<K1T>(I0 + ii, k2, true)) { acc[ii] = fmaf(Pm[k2], wv[e / 4][e % 4], acc[ii]); ++e; }
                            }
                        }
                    }
                } else {
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
                }
                // every lane ends up with all totals; lane ii keeps filter i0+ii's and one logf serves the whole pass
                float mine = 0.f;
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    if (K1T > 0 && i0 + ii >= K1T) continue;
                    const float tot = row16_sum(acc[ii]);
                    mine = l == ii ? tot : mine;
                }
                if (l < 8 && i0 + l < K1) lgb[i0 + l] = logf(0.25f * mine + FLT_MIN);
            };
#ifdef RP_ABL_NOMEL
            if (false)
#endif
            {
                if constexpr (K1T > 0) {
                    mel_pass(std::integral_constant<int, 0>{});
                    if constexpr (K1T > 8) mel_pass(std::integral_constant<int, 8>{});
                    if constexpr (K1T > 16) mel_pass(std::integral_constant<int, 16>{});
                    static_assert(K1T <= 24, "three passes of eight filters");
                } else {
                    for (int i0 = 0; i0 < K1; i0 += 8) mel_pass(i0);
                }
            }
        }
        wave_lds_sync();
        // ---- DCT-II x2, coefficient 0 dropped, src/mfcc/extractor.rs:84,146-163.  Sequential
        // multiply-then-add in the reference's order (NOT fused): see the file header.
        const size_t j = j0 + grp;
        if (j < first_frame + n_frames) {
#ifndef RP_MFCC_VECTOR_ADDR
            // wave-uniform 64-bit row base + a 32-bit lane offset (frame of the group): the stores take `saddr + voffset`; forming the
            // whole address per lane cost six v_mad_u64_u32 and their moves per tile
            float *dst = out + (s_now * out_frame_pitch + (j0 - first_frame)) * (size_t)K + grp * K;
            // optional second copy, frames packed [n_frames][K] (the single-stream path hands the new frames to the host)
            float *dst2 = out2 ? out2 + (s_now * n_frames + (j0 - first_frame)) * (size_t)K + grp * K : nullptr;
#else
            float *dst = out + (s_now * out_frame_pitch + (j - first_frame)) * (size_t)K;
            float *dst2 = out2 ? out2 + (s_now * n_frames + (j - first_frame)) * (size_t)K : nullptr;
#endif
            for (int c = 1 + l; c <= K; c += 16) {
                float sum = 0.f;
                for (int n = 0; n < K1; ++n) sum += lgb[n] * dct[c * K1 + n];
                dst[c - 1] = 2.f * sum;
                if (dst2) dst2[c - 1] = 2.f * sum;
            }
        }
        wave_lds_sync();  // lgb / scratch are reused by the next tile
    }
}

template <class TIN>
static hipError_t launch_mfcc_t(hipStream_t st, const MfccTablesDev &tb, const TIN *pcm, size_t S, size_t n_samples,
                                size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc,
                                float *mfcc2 = nullptr) {
    if (S == 0 || n_frames == 0) return hipSuccess;
    const size_t tiles = (n_frames + kMfccFramesPerWave - 1) / kMfccFramesPerWave;
    const size_t total = tiles * S;
    if (tiles > 0xffffffffULL) return hipErrorInvalidValue;
    if (mfcc_lds_bytes(tb.K1, false) > 160 * 1024) return hipErrorInvalidValue;
    // 4-sample vector loads need rows aligned to 4 samples (and at least one full vector before the last sample)
    const bool vec4 = (reinterpret_cast<uintptr_t>(pcm) % (4 * sizeof(TIN)) == 0) && (pcm_stride % 4 == 0) && n_samples >= 8;
    size_t blocks = (total + kMfccWaves - 1) / kMfccWaves;
    {   // Waves walk the tiles grid-stride; 16 384 workgroups (16 rounds of the 1 024 resident ones) measured 4-5 % faster than 2 048
        // at C3 (7.7-7.9 against 8.1-8.2 ms): late rounds level out what the CUs finish unevenly, and the tables a workgroup
        // stages (17 KB from L2) are small.  Small batches are the other way round (round 4, BASELINE C2: 101 k wave-tiles): with
        // 16 384 workgroups a wave gets 1.5 tiles and the per-workgroup set-up (tables, twiddles, the first fetch) is most of its
        // life -- 0.152 ms against 0.125-0.135 for any cap from 1 024 to 8 192.  So: at least ~6 tiles per wave, between 1 024 (what
        // is resident) and 16 384 workgroups.  RP_MFCC_BLOCKS overrides the cap for tuning.
        static const size_t env_cap = [] { const char *e = getenv("RP_MFCC_BLOCKS"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)0; }();
        size_t cap = total / (6 * kMfccWaves);
        cap = cap < 1024 ? 1024 : cap > 16384 ? 16384 : cap;
        if (env_cap) cap = env_cap;
        if (blocks > cap) blocks = cap;
    }
    // RP_MFCC_LDS_PAD (tuning knob, round 6): bytes added to the LDS request = fewer workgroups resident per CU (40.8 KB each: four;
    // +14 000: three; +41 000: two) -- the experiment behind DESIGN.md 8.4 "does the clock rise with fewer co-resident waves"
    static const size_t lds_pad = [] { const char *e = getenv("RP_MFCC_LDS_PAD"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)0; }();
#define RP_MFCC_LAUNCH(V, KT)                                                                                              \
    do {                                                                                                                   \
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mfcc_kernel<V, KT, TIN>), 160 * 1024);             \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL((mfcc_kernel<V, KT, TIN>), dim3((unsigned)blocks), dim3(kMfccThreads), mfcc_lds_bytes(tb.K1, KT > 0) + lds_pad, st, pcm, \
                           n_samples, pcm_stride, (unsigned)tiles, total, first_frame, n_frames, out_frame_pitch, tb.K1, tb.hamming,  \
                           tb.tw240, tb.tw480, KT > 0 ? tb.melw : tb.fb, tb.dct, mfcc, mfcc2);                              \
    } while (0)
    if (vec4 && tb.K1 == 6 && tb.mel_sparse) RP_MFCC_LAUNCH(true, 6);
    else if (vec4 && tb.K1 == 14 && tb.mel_sparse) RP_MFCC_LAUNCH(true, 14);
    else if (vec4 && tb.K1 == 17 && tb.mel_sparse) RP_MFCC_LAUNCH(true, 17);
    else if (vec4) RP_MFCC_LAUNCH(true, 0);
    else RP_MFCC_LAUNCH(false, 0);
#undef RP_MFCC_LAUNCH
    return hipGetLastError();
}

// Live-stream form: S streams of [history chunk (hist, f32) | n_chunks new chunks (pcm, any sample type)]; the
// 3 * n_chunks new frames go to mfcc (row pitch out_frame_pitch frames) and the last chunk to hist_out.  Returns
// hipErrorNotSupported when the rows do not allow 4-sample loads or the mel table is not one of the sparse ones.
template <class TIN>
static hipError_t launch_mfcc_stream_t(hipStream_t st, const MfccTablesDev &tb, const TIN *pcm, size_t S, size_t n_chunks,
                                       size_t pcm_stride, const float *hist, size_t hist_pitch, float *hist_out,
                                       size_t out_frame_pitch, float *mfcc) {
    if (S == 0 || n_chunks == 0) return hipSuccess;
    const size_t n_frames = 3 * n_chunks, n_samples = (1 + n_chunks) * kFrame;
    const size_t tiles = (n_frames + kMfccFramesPerWave - 1) / kMfccFramesPerWave, total = tiles * S;
    const size_t lds = mfcc_lds_bytes(tb.K1, false);  // upper bound, for the feasibility check
    const bool vec4 = (reinterpret_cast<uintptr_t>(pcm) % (4 * sizeof(TIN)) == 0) && (pcm_stride % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(hist) % 16 == 0) && (reinterpret_cast<uintptr_t>(hist_out) % 16 == 0) &&
                      (hist_pitch % 4 == 0);
    if (!vec4 || lds > 160 * 1024 || tiles > 0xffffffffULL) return hipErrorNotSupported;
    size_t blocks = (total + kMfccWaves - 1) / kMfccWaves;
    {   // three workgroups per CU are resident (HS form); RP_MFCC_HS_BLOCKS overrides the cap for tuning
        static const size_t cap = [] { const char *e = getenv("RP_MFCC_HS_BLOCKS"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)1536; }();
        if (blocks > cap) blocks = cap;
    }
#define RP_MFCC_LAUNCH_HS(KT)                                                                                               \
    do {                                                                                                                   \
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mfcc_kernel<true, KT, TIN, true>), 160 * 1024);    \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL((mfcc_kernel<true, KT, TIN, true>), dim3((unsigned)blocks), dim3(kMfccThreads), mfcc_lds_bytes(tb.K1, KT > 0), st, pcm, \
                           n_samples, pcm_stride, (unsigned)tiles, total, (size_t)0, n_frames, out_frame_pitch, tb.K1,     \
                           tb.hamming, tb.tw240, tb.tw480, KT > 0 ? tb.melw : tb.fb, tb.dct, mfcc, (float *)nullptr, hist, hist_pitch, hist_out); \
    } while (0)
    if (tb.K1 == 6 && tb.mel_sparse) RP_MFCC_LAUNCH_HS(6);
    else if (tb.K1 == 14 && tb.mel_sparse) RP_MFCC_LAUNCH_HS(14);
    else if (tb.K1 == 17 && tb.mel_sparse) RP_MFCC_LAUNCH_HS(17);
    else RP_MFCC_LAUNCH_HS(0);
#undef RP_MFCC_LAUNCH_HS
    return hipGetLastError();
}

hipError_t launch_mfcc_stream(hipStream_t st, const MfccTablesDev &tb, const void *pcm, int fmt, size_t S, size_t n_chunks,
                              size_t pcm_stride, const float *hist, size_t hist_pitch, float *hist_out, size_t out_frame_pitch,
                              float *mfcc) {
    switch (fmt) {
    case 0: return launch_mfcc_stream_t<int8_t>(st, tb, static_cast<const int8_t *>(pcm), S, n_chunks, pcm_stride, hist, hist_pitch, hist_out, out_frame_pitch, mfcc);
    case 1: return launch_mfcc_stream_t<int16_t>(st, tb, static_cast<const int16_t *>(pcm), S, n_chunks, pcm_stride, hist, hist_pitch, hist_out, out_frame_pitch, mfcc);
    case 2: return launch_mfcc_stream_t<int32_t>(st, tb, static_cast<const int32_t *>(pcm), S, n_chunks, pcm_stride, hist, hist_pitch, hist_out, out_frame_pitch, mfcc);
    case 3: return launch_mfcc_stream_t<float>(st, tb, static_cast<const float *>(pcm), S, n_chunks, pcm_stride, hist, hist_pitch, hist_out, out_frame_pitch, mfcc);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mfcc(hipStream_t st, const MfccTablesDev &tb, const float *pcm, size_t S, size_t n_samples,
                       size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc, float *mfcc2) {
    return launch_mfcc_t<float>(st, tb, pcm, S, n_samples, pcm_stride, first_frame, n_frames, out_frame_pitch, mfcc, mfcc2);
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

}  // namespace rp
