// rp_frontend.hip -- decode + GainNormalizerFilter + BandPassFilter over whole streams
// (src/audio/gain_normalizer_filter.rs:14-55, src/audio/band_pass_filter.rs:19-55, src/detector.rs:358-371).
#include "rp_device.h"

namespace rp {

// --------------------------------------------------------------------- front-end
// Sample decode + GainNormalizerFilter + BandPassFilter for whole streams (src/detector.rs:358-371).
// None of it depends on the detection state (reset() leaves both filters alone, :290-302), so it is
// a pure function of the stream: per-chunk RMS in parallel, the gain recursion per stream over the
// chunk RMS values, then gain + biquad per stream along time (one lane per stream: a lane re-reads
// its own 128-byte lines from L1, HBM traffic stays one read + one write of the PCM); chunk_rms_staged_kernel and
// apply_filters_lines_kernel below are the forms used whenever the row pitch allows 4-sample accesses.
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

// The same sums with the chunks of a wave staged through LDS.  One lane per chunk reading its own 1 920 bytes (above) makes
// every load instruction touch 64 lines for 16 bytes each and leaves it to the caches to keep them until the lane comes
// back: 4.3 TB/s (f32 in) / 5.0 (i16) at C3 size.  Here a wave takes 64 consecutive (stream, chunk) rows in slices of 96
// samples: 8 lanes fetch a row's slice piece by piece (whole 128-byte lines for f32, half lines for i16: a chunk is 7.5
// lines of i16), the next slice travels while this one is summed, and the sequential sum of a chunk (the order of
// get_rms_level) runs down a lane's own LDS row (pitch 100 floats: conflict-free 16-byte reads).
constexpr int kRmsSlice = 96, kRmsPitch = 100, kRmsMoves = 64 * (kRmsSlice / 4) / 64;
template <class TIN>
__global__ __launch_bounds__(64) void chunk_rms_staged_kernel(const TIN *__restrict__ pcm, size_t S, size_t n_chunks, size_t pcm_stride,
                                                              float *__restrict__ rms) {
    __shared__ __attribute__((aligned(16))) float tile[64 * kRmsPitch];
    __shared__ unsigned long long rowbase[64];
    const int lane = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * 64, total = S * n_chunks, i = i0 + lane;
    {   // row = lane: where chunk i starts; rows past the end re-read the last chunk
        const size_t ic = i < total ? i : total - 1, s = ic / n_chunks, c = ic - s * n_chunks;
        rowbase[lane] = s * pcm_stride + c * kFrame;
    }
    wave_lds_sync();
    using Raw4 = typename SampleIn<TIN>::Raw4;
    constexpr int LPR = kRmsSlice / 4;  // lanes per row piece: group it * 64 + lane -> (row, 4-sample column)
    Raw4 r[kRmsMoves];
    const TIN *src[kRmsMoves];
    unsigned lds[kRmsMoves];
#pragma unroll
    for (int it = 0; it < kRmsMoves; ++it) {
        const unsigned gi = it * 64 + lane, row = gi / LPR, c4 = gi - row * LPR;
        src[it] = pcm + rowbase[row] + 4 * c4;
        lds[it] = row * kRmsPitch + 4 * c4;
    }
    auto fetch = [&](int sl) {
#pragma unroll
        for (int it = 0; it < kRmsMoves; ++it) r[it] = SampleIn<TIN>::ldraw(src[it] + sl * kRmsSlice);
    };
    fetch(0);
    float sum_squared = 0.0f;
    const float *mine = tile + lane * kRmsPitch;
#pragma unroll 1
    for (int sl = 0; sl < kFrame / kRmsSlice; ++sl) {
#pragma unroll
        for (int it = 0; it < kRmsMoves; ++it) {
            const float4 f = SampleIn<TIN>::cvt4(r[it]);
            *reinterpret_cast<f32x4 *>(&tile[lds[it]]) = f32x4{f.x, f.y, f.z, f.w};
        }
        if (sl + 1 < kFrame / kRmsSlice) fetch(sl + 1);
        wave_lds_sync();
#pragma unroll
        for (int k0 = 0; k0 < kRmsSlice; k0 += 24) {
            f32x4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = *reinterpret_cast<const f32x4 *>(mine + k0 + 4 * j);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                sum_squared += v[j].x * v[j].x; sum_squared += v[j].y * v[j].y;
                sum_squared += v[j].z * v[j].z; sum_squared += v[j].w * v[j].w;
            }
        }
        wave_lds_sync();
    }
    if (i < total) rms[i] = sqrtf(sum_squared / (float)kFrame);
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
            // head + i < 2 * window_size: the wrap is a compare, not a division (0.54 -> 0.1x ms at C3 size)
            if (len < window_size) { const int t = head + len; w[(t >= window_size ? t - window_size : t) * pitch] = r; ++len; }
            else { w[head * pitch] = r; head = head + 1 == window_size ? 0 : head + 1; }  // push + drain(0..1)
            float sum = 0.f;  // oldest first, the order of iter().sum(); four reads in flight per step
            int i = 0, t = head;
            auto step = [&](int u) { const int n = u + 1; return n >= window_size ? n - window_size : n; };
            for (; i + 4 <= len; i += 4) {
                const int t1 = step(t), t2 = step(t1), t3 = step(t2);
                const float a = w[t * pitch], b = w[t1 * pitch], c2 = w[t2 * pitch], d = w[t3 * pitch];
                sum += a; sum += b; sum += c2; sum += d;
                t = step(t3);
            }
            for (; i < len; ++i) { sum += w[t * pitch]; t = step(t); }
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

// The same filter with the streams of a wave staged through LDS in tiles of T = 64 or 128 samples: rows are read from /
// written to HBM along time (whole 128-byte lines both ways: f32 out 256 / 512 bytes per stream and tile, i16 in 128 / 256)
// instead of 16-byte pieces 64 streams apart; the recurrence runs down the columns (lane = stream; row pitch T + 4 floats
// keeps the 16-byte LDS accesses of 16 neighbouring lanes on different banks).  Rounds 2-3 used 120-sample tiles (4 per
// chunk), whose rows start in the middle of a line -- measured in round 4 by tile length at C3 size, i16 in, whole front-end:
// 60 samples 10.2 ms, 120 8.1, 96 (whole lines out, half lines in) 7.3, 128 6.9, 64 with two waves taking turns 6.5.
// T / 4 is a power of two, so the (row, column) of a lane's 16-byte group is a shift and a mask and the addresses of the
// T / 4 wave-wide moves differ by a wave-uniform step: one vector offset, scalar bases.  A tile may straddle one chunk
// boundary: the gain changes at column cb, a multiple of 32 = a block boundary of the column loop.
// W waves per 64 streams take the tiles in turn (W = 1: the plain form): tile k belongs to wave k % W; in phase k that
// wave runs the recurrence down tile k's columns (the only part that is serial along time; x1 x2 y1 y2 travel from wave to
// wave through LDS) while the wave that owned tile k - 1 reads its results back, stores them, decodes its next tile into its
// own LDS buffer and fetches the one after; one workgroup barrier per phase: the LDS tile bounds the occupancy, and two waves
// on a 64-sample tile each cover the other's waiting (one wave per SIMD: SQ_WAIT_INST_ANY 0.34 of SQ_WAVE_CYCLES) in the LDS
// of one 128-sample tile.  Same arithmetic per sample as apply_filters_kernel.
// A call's floor: 3 ms for 64 000 samples per stream whatever the batch (2 048 streams 2.98 ms, 16 384 3.32, 32 768 4.00, 65 536
// 6.1-6.3): the tiles of a stream follow each other, and a phase lasts as long as the longer of its two sides -- the walk (all 17
// instructions of the filter per sample, issued in order) and the other wave's store / decode / fetch, about 3 us each.  Measured
// and dropped in round 4: 32-sample tiles with four waves (a phase still has ONE walk and ONE decode: 3.1-3.7 ms for 2 048-16 384
// streams); the feed-forward half of the biquad and the gain moved into the time-major decode (DPP neighbours, bit-exact) so
// that the walk keeps 4 instructions per sample -- the decode side then is the long one: 4.1 ms at 2 048 streams with two or four
// waves, 7.4 ms at 65 536.  The two sides are balanced as they are; only less work per sample on both would lower the floor.
template <class TIN, bool GAIN, bool BP, int T, int W>
__global__ __launch_bounds__(64 * W) void apply_filters_lines_kernel(const TIN *__restrict__ pcm, size_t S, size_t n_samples,
                                                                     size_t n_chunks, size_t pcm_stride, const float *__restrict__ gains,
                                                                     BiquadCoef q, float *__restrict__ out, size_t out_stride) {
    constexpr int G = T / 4, RPM = 64 / G, PITCH = T + 4, kMoves = G, BLK = 4, HALF = kMoves / 2;
    static_assert(T == 64 || T == 128, "tile shape");
    __shared__ __attribute__((aligned(16))) float tiles[W][64 * PITCH];
    __shared__ float carry[W > 1 ? 4 : 1][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t s0 = (size_t)blockIdx.x * 64, s = s0 + lane;
    // the last workgroup may hold fewer than 64 streams: its loads clamp the row, its stores test it (one launch: a second one for that
    // workgroup alone would add a whole stream's 3 ms to the call)
    const bool FULL = s0 + 64 <= S;   // workgroup-uniform
    const bool live = FULL || s < S;
    const size_t valid = n_chunks * kFrame, n_tiles = (valid + T - 1) / T;
    const unsigned rows_here = FULL ? 64u : (unsigned)(S - s0);
    const size_t s_c = live ? s : S - 1;
    float *tile = tiles[w];
    const unsigned lrow = (unsigned)lane / G, lc4 = (unsigned)lane % G;
    // a lane's share of the addresses: 32-bit element offsets from wave-uniform bases (the host checks the pitches fit)
    const unsigned in_off = lrow * (unsigned)pcm_stride + 4 * lc4, out_off = lrow * (unsigned)out_stride + 4 * lc4;
    const unsigned lds_off = lrow * PITCH + 4 * lc4;
    using Raw4 = typename SampleIn<TIN>::Raw4;
    Raw4 r[kMoves];
    float ga_next = 1.f, gb_next = 1.f, ga = 1.f, gb = 1.f, g = 1.f;
    int cb_next = T, cb = T;
    auto fetch = [&](size_t ti) {
        const bool whole = (ti + 1) * T <= valid;  // the last tile may end early: its spare columns re-read the last group
        const unsigned col_fix = whole ? 0u : (ti * T + 4 * lc4 < valid ? 0u : (unsigned)(ti * T + 4 * lc4 - (valid - 4)));
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            if (FULL) {
                const TIN *base = pcm + (s0 + it * RPM) * pcm_stride + ti * T;
                r[it] = SampleIn<TIN>::ldraw(base + (in_off - col_fix));
            } else {  // rows past S re-read the last stream
                const unsigned row = it * RPM + lrow, rc = row < rows_here ? row : rows_here - 1;
                r[it] = SampleIn<TIN>::ldraw(pcm + (s0 + rc) * pcm_stride + ti * T + 4 * lc4 - col_fix);
            }
        }
        const size_t ca = ti * T / kFrame;
        const size_t edge = (ca + 1) * kFrame - ti * T;
        cb_next = edge < (size_t)T ? (int)edge : T;
        if (GAIN) {
            ga_next = gains[s_c * n_chunks + ca];
            gb_next = gains[s_c * n_chunks + (ca + 1 < n_chunks ? ca + 1 : ca)];
        }
    };
    auto decode = [&]() {
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            const float4 f = SampleIn<TIN>::cvt4(r[it]);
            *reinterpret_cast<f32x4 *>(&tile[it * RPM * PITCH + lds_off]) = f32x4{f.x, f.y, f.z, f.w};
        }
        ga = ga_next; gb = gb_next; cb = cb_next;
    };
    // results of tile ti: LDS -> registers -> HBM, in two halves (T / 8 moves in flight each) to keep the register count down
    auto drain = [&](size_t ti) {
        const bool whole = (ti + 1) * T <= valid;
        const bool col_ok = whole || ti * T + 4 * lc4 < valid;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 o[HALF];
#pragma unroll
            for (int i = 0; i < HALF; ++i) o[i] = *reinterpret_cast<const f32x4 *>(&tile[(h * HALF + i) * RPM * PITCH + lds_off]);
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                const int it = h * HALF + i;
                float *base = out + (s0 + it * RPM) * out_stride + ti * T;
                if (col_ok && (FULL || it * RPM + lrow < rows_here)) *reinterpret_cast<f32x4 *>(base + out_off) = o[i];
            }
        }
    };
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
    auto one = [&](float v) {
        if (GAIN) {  // g == 1 leaves v alone in the reference; v * 1 and the clamp of a value outside [-1, 1] do not
            float u = v * g; u = u < -1.f ? -1.f : u; u = u > 1.f ? 1.f : u;
            v = g != 1.f ? u : v;
        }
        if (BP) {
            const float f = q.a0 * v + q.a1 * x1 + q.a2 * x2 - q.b1 * y1 - q.b2 * y2;
            x2 = x1; x1 = v; y2 = y1; y1 = f;
            v = f;
        }
        return v;
    };
    if ((size_t)w < n_tiles) { fetch(w); decode(); }
    if ((size_t)w + W < n_tiles) fetch((size_t)w + W);
    float *mine = tile + lane * PITCH;
    for (size_t k = 0; k <= n_tiles; ++k) {
        if (k >= 1 && (int)((k - 1) % W) == w) {  // tile k - 1 is done: results out, the next tile of this wave in
            wave_lds_sync();
            drain(k - 1);
            if (k - 1 + W < n_tiles) {
                wave_lds_sync();
                decode();
                if (k - 1 + 2 * W < n_tiles) fetch(k - 1 + 2 * W);
            }
        }
        if ((int)(k % W) == w && k < n_tiles) {  // the recurrence down the columns of tile k
            wave_lds_sync();
            if (W > 1 && BP && k) { x1 = carry[0][lane]; x2 = carry[1][lane]; y1 = carry[2][lane]; y2 = carry[3][lane]; }
            f32x4 cur[BLK], nxt[BLK];
#pragma unroll
            for (int j = 0; j < BLK; ++j) nxt[j] = *reinterpret_cast<f32x4 *>(mine + 4 * j);
#pragma unroll 1
            for (int c = 0; c < T; c += 4 * BLK) {
#pragma unroll
                for (int j = 0; j < BLK; ++j) cur[j] = nxt[j];
                if (c + 4 * BLK < T) {
#pragma unroll
                    for (int j = 0; j < BLK; ++j) nxt[j] = *reinterpret_cast<f32x4 *>(mine + c + 4 * BLK + 4 * j);
                }
                if (GAIN) g = c >= cb ? gb : ga;
#pragma unroll
                for (int j = 0; j < BLK; ++j) {
                    f32x4 v = cur[j];
                    v.x = one(v.x); v.y = one(v.y); v.z = one(v.z); v.w = one(v.w);
                    *reinterpret_cast<f32x4 *>(mine + c + 4 * j) = v;
                }
            }
            if (W > 1 && BP) { carry[0][lane] = x1; carry[1][lane] = x2; carry[2][lane] = y1; carry[3][lane] = y2; }
        }
        if (W > 1) __syncthreads();
    }
    if (live && w == 0)  // tail shorter than a chunk: never framed
        for (size_t i = valid; i < n_samples; ++i) out[s * out_stride + i] = SampleIn<TIN>::cvt(pcm[s * pcm_stride + i]);
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
        static const int rms_staged = getenv("RP_FRONTEND_RMS") ? atoi(getenv("RP_FRONTEND_RMS")) : 1;
        if (vec4 && rms_staged && (n + 63) / 64 <= 0x7fffffffULL)
            hipLaunchKernelGGL(chunk_rms_staged_kernel<TIN>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, pcm, S, n_chunks, pcm_stride, rms);
        else
            hipLaunchKernelGGL(chunk_rms_kernel<TIN>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pcm, S, n_chunks, pcm_stride, vec4, rms);
        if (gain_on) {
            const size_t ring_lds = (size_t)window_size * 64 * sizeof(float);
            const int in_lds = ring_lds <= 48 * 1024;
            hipLaunchKernelGGL(gain_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), in_lds ? ring_lds : 0, st, rms, S, n_chunks,
                               rms_level_ref, min_gain, max_gain, window_size, in_lds, ring, gains);
        }
    }
    if (vec4 && pcm_stride < (1u << 29) && out_stride < (1u << 29)) {  // 32-bit lane offsets inside a wave's rows
        // 64-sample tiles, two waves per 64 streams taking turns; RP_FRONTEND_TILE=128 = one wave on 128-sample tiles (A/B)
        static const int tlen = getenv("RP_FRONTEND_TILE") ? atoi(getenv("RP_FRONTEND_TILE")) : 64;
#define RP_LINES(G, B, T, W)                                                                                                            \
    hipLaunchKernelGGL((apply_filters_lines_kernel<TIN, G, B, T, W>), dim3((unsigned)((S + 63) / 64)), dim3(64 * W), 0, st, pcm, S,     \
                       n_samples, n_chunks, pcm_stride, gains, q, out, out_stride)
#define RP_TILED(G, B)                                                                                                                  \
    do {                                                                                                                                \
        if (tlen == 128) RP_LINES(G, B, 128, 1);                                                                                        \
        else RP_LINES(G, B, 64, 2);                                                                                                     \
    } while (0)
        if (gain_on && band_pass) RP_TILED(true, true);
        else if (gain_on) RP_TILED(true, false);
        else if (band_pass) RP_TILED(false, true);
        else RP_TILED(false, false);
#undef RP_TILED
#undef RP_LINES
    }
    else
        hipLaunchKernelGGL(apply_filters_kernel<TIN>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, pcm, S, n_samples, n_chunks,
                           pcm_stride, gain_on ? gains : nullptr, band_pass, q, 0, out, out_stride);
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

}  // namespace rp
