// rp_resample.hip -- sample-rate conversion in front of the path (src/audio/encoder.rs:41-83, rubato FftFixedInOut):
// resample_mfma_kernel (any rate, one f32 matrix product per output frame) and resample48_fft_kernel (48 kHz, pruned
// FFT on the FFT-240 machinery).  profiles/HISTORY.md §4.5-4.6.
#include "rp_device.h"

namespace rp {

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
    f32x4 wreg[NV];  // plain vector values: HIP's float4 struct is copied with memcpy, which keeps the array in scratch
    auto wload = [&](int g) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kRsWaves;
            const int o = i / (kRsKG / 4), c = i - o * (kRsKG / 4);
            if (o < FO) wreg[v] = *reinterpret_cast<const f32x4 *>(g2t + (size_t)o * kpad + g * kRsKG + 4 * c);
        }
    };
    auto wstore = [&](float *dst) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kRsWaves;
            const int o = i / (kRsKG / 4), c = i - o * (kRsKG / 4);
            if (o < FO) *reinterpret_cast<f32x4 *>(dst + o * kRsPitch + 4 * c) = wreg[v];
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
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(resample_mfma_kernel<NT>), (int)lds); e != hipSuccess) return e;
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
    // the next frame's samples are fetched (raw) while this frame is transformed: the HBM round trip is off the chain
    using Raw4 = typename SampleIn<TIN>::Raw4;
    constexpr int kV4 = (360 + 63) / 64, kV1 = (1440 + 63) / 64;
    Raw4 r4[kV4];
    TIN r1[kV1];
    auto fetch = [&](long c) {
        const TIN *x = xs + s * xs_pitch + (size_t)(c + (has_hist ? 1 : 0)) * 1440 * channels;
        if (vec) {
#pragma unroll
            for (int k = 0; k < kV4; ++k) { const int i = k * 64 + lane; r4[k] = SampleIn<TIN>::ldraw(x + 4 * (i < 360 ? i : 359)); }
        } else {
#pragma unroll
            for (int k = 0; k < kV1; ++k) { const int i = k * 64 + lane; r1[k] = x[(size_t)(i < 1440 ? i : 1439) * channels]; }
        }
    };
    auto land = [&]() {
        if (vec) {
#pragma unroll
            for (int k = 0; k < kV4; ++k) {
                const int i = k * 64 + lane;
                if (i < 360) reinterpret_cast<float4 *>(xin)[i] = SampleIn<TIN>::cvt4(r4[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kV1; ++k) { const int i = k * 64 + lane; if (i < 1440) xin[i] = SampleIn<TIN>::cvt(r1[k]); }
        }
    };
    const long c_first = (c0 == 0 && !has_hist && !prev) ? 0 : c0 - 1;
    bool fetched = false;
    for (long c = c_first; c < c1; ++c) {
        wave_lds_sync();
        if (c < 0 && prev) {
            const float *x = prev + s * 1440;
            for (int i = lane; i < 360; i += 64) reinterpret_cast<float4 *>(xin)[i] = reinterpret_cast<const float4 *>(x)[i];
        } else {
            if (!fetched) fetch(c);
            land();
        }
        fetched = c + 1 < c1;
        if (fetched) fetch(c + 1);
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
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(resample48_fft_kernel<TIN>), (int)lds); e != hipSuccess) return e;
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

}  // namespace rp
