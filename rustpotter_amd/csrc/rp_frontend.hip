// rp_frontend.hip -- decode + GainNormalizerFilter + BandPassFilter over whole streams
// (src/audio/gain_normalizer_filter.rs:14-55, src/audio/band_pass_filter.rs:19-55, src/detector.rs:358-371).
#include "rp_device.h"

namespace rp {

// --------------------------------------------------------------------- front-end
// Sample decode + GainNormalizerFilter + BandPassFilter for whole streams (src/detector.rs:358-371).
// None of it depends on the detection state (reset() leaves both filters alone, :290-302), so it is
// a pure function of the stream: per-chunk RMS in parallel, the gain recursion per stream over the
// chunk RMS values, then gain + biquad per stream along time (one lane per stream: a lane re-reads
// its own 128-byte lines from L1, HBM traffic stays one read + one write of the PCM); apply_filters_tiled_kernel
// below is the form used whenever the row pitch allows 4-sample accesses.
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

// The same filter with the streams of a wave staged through LDS: 64 streams x 120 samples per tile (4 tiles per chunk).
// Rows are read from / written to HBM along time (480 contiguous bytes of f32 per stream and tile instead of 16-byte
// pieces 64 streams apart), the recurrence runs down the columns (lane = stream, row pitch 124 floats keeps the
// 16-byte LDS accesses of 16 neighbouring lanes on different banks).  The next tile is fetched into registers while
// this one is filtered.  Same arithmetic per sample as apply_filters_kernel.
constexpr int kFeTile = 120, kFeRow = 124, kFeLanes = kFeTile / 4, kFeBlock = 6;
template <class TIN, bool GAIN, bool BP>
__global__ __launch_bounds__(64) void apply_filters_tiled_kernel(const TIN *__restrict__ pcm, size_t S, size_t n_samples, size_t n_chunks,
                                                                 size_t pcm_stride, const float *__restrict__ gains, BiquadCoef q,
                                                                 float *__restrict__ out, size_t out_stride) {
    __shared__ __attribute__((aligned(16))) float tile[64 * kFeRow];
    const int lane = threadIdx.x;
    const size_t s0 = (size_t)blockIdx.x * 64, s = s0 + lane;
    const bool live = s < S;
    const size_t n_tiles = n_chunks * (kFrame / kFeTile);
    const unsigned rows_here = S - s0 < 64 ? (unsigned)(S - s0) : 64u;
    // the tile as 64 rows x 30 four-sample groups = 30 wave-wide moves: group it * 64 + lane -> (row, column); every lane
    // fetches (rows past S re-read the last stream: a guarded load would sit in its own branch with its own wait).
    // 30 loads + 30 stores + the gain in flight stay under the 63 the wait counter can express.
    using Raw4 = typename SampleIn<TIN>::Raw4;
    constexpr int kMoves = 64 * kFeLanes / 64;
    Raw4 r[kMoves];
    float g_next = 1.f;
    const size_t s_c = live ? s : S - 1;
    auto row_of = [&](int it) { return (unsigned)(it * 64 + lane) / (unsigned)kFeLanes; };
    auto fetch = [&](size_t ti) {
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            const unsigned row = row_of(it), c4 = (unsigned)(it * 64 + lane) - row * kFeLanes;
            const unsigned rc = row < rows_here ? row : rows_here - 1;
            r[it] = SampleIn<TIN>::ldraw(pcm + (s0 + rc) * pcm_stride + ti * kFeTile + 4 * c4);
        }
        if (GAIN && ti % (kFrame / kFeTile) == 0) g_next = gains[s_c * n_chunks + ti / (kFrame / kFeTile)];
    };
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f, g = 1.f;
    auto one = [&](float v) {
        if (GAIN) {  // g == 1 leaves v alone in the reference; v * 1 and the clamp of a value outside [-1, 1] do not
            float w = v * g; w = w < -1.f ? -1.f : w; w = w > 1.f ? 1.f : w;
            v = g != 1.f ? w : v;
        }
        if (BP) {
            const float o = q.a0 * v + q.a1 * x1 + q.a2 * x2 - q.b1 * y1 - q.b2 * y2;
            x2 = x1; x1 = v; y2 = y1; y1 = o;
            v = o;
        }
        return v;
    };
    // Order inside one pass, so that nothing that was just issued is waited for: store the previous tile's results from
    // registers, decode this tile into LDS, fetch the next tile, filter, read the results back into registers.  The
    // single wait at the top of the next pass (loads and stores share one in-order counter) then only meets
    // operations that had the whole filter phase to finish.
    f32x4 o[kMoves];
    auto store = [&](size_t ti) {
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            const unsigned row = row_of(it), c4 = (unsigned)(it * 64 + lane) - row * kFeLanes;
            if (row < rows_here) *reinterpret_cast<f32x4 *>(out + (s0 + row) * out_stride + ti * kFeTile + 4 * c4) = o[it];
        }
    };
    if (n_tiles) fetch(0);
    float *mine = tile + lane * kFeRow;
    for (size_t ti = 0; ti < n_tiles; ++ti) {
        if (ti) store(ti - 1);
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            const unsigned row = row_of(it), c4 = (unsigned)(it * 64 + lane) - row * kFeLanes;
            const float4 f = SampleIn<TIN>::cvt4(r[it]);
            *reinterpret_cast<f32x4 *>(&tile[row * kFeRow + 4 * c4]) = f32x4{f.x, f.y, f.z, f.w};
        }
        if (GAIN && ti % (kFrame / kFeTile) == 0) g = g_next;
        if (ti + 1 < n_tiles) fetch(ti + 1);
        wave_lds_sync();
        f32x4 cur[kFeBlock], nxt[kFeBlock];  // the next 24 samples are read from LDS while these 24 go through the recurrence
#pragma unroll
        for (int j = 0; j < kFeBlock; ++j) nxt[j] = *reinterpret_cast<f32x4 *>(mine + 4 * j);
#pragma unroll 1
        for (int k = 0; k < kFeTile; k += 4 * kFeBlock) {
#pragma unroll
            for (int j = 0; j < kFeBlock; ++j) cur[j] = nxt[j];
            if (k + 4 * kFeBlock < kFeTile) {
#pragma unroll
                for (int j = 0; j < kFeBlock; ++j) nxt[j] = *reinterpret_cast<f32x4 *>(mine + k + 4 * kFeBlock + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < kFeBlock; ++j) {
                f32x4 v = cur[j];
                v.x = one(v.x); v.y = one(v.y); v.z = one(v.z); v.w = one(v.w);
                *reinterpret_cast<f32x4 *>(mine + k + 4 * j) = v;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int it = 0; it < kMoves; ++it) {
            const unsigned row = row_of(it), c4 = (unsigned)(it * 64 + lane) - row * kFeLanes;
            o[it] = *reinterpret_cast<const f32x4 *>(&tile[row * kFeRow + 4 * c4]);
        }
        wave_lds_sync();
    }
    if (n_tiles) store(n_tiles - 1);
    if (live)  // tail shorter than a chunk: never framed
        for (size_t k = n_chunks * kFrame; k < n_samples; ++k) out[s * out_stride + k] = SampleIn<TIN>::cvt(pcm[s * pcm_stride + k]);
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
    if (vec4) {
        const dim3 grid((unsigned)((S + 63) / 64));
#define RP_TILED(G, B) hipLaunchKernelGGL((apply_filters_tiled_kernel<TIN, G, B>), grid, dim3(64), 0, st, pcm, S, n_samples, n_chunks, \
                                          pcm_stride, gains, q, out, out_stride)
        if (gain_on && band_pass) RP_TILED(true, true);
        else if (gain_on) RP_TILED(true, false);
        else if (band_pass) RP_TILED(false, true);
        else RP_TILED(false, false);
#undef RP_TILED
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
