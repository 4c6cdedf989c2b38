// rp_scan.hip -- the detection state machine over precomputed window scores: vad_value_kernel, scan_kernel
// (src/detector.rs:290-302,377-454, src/mfcc/vad.rs) and the live-stream variants that carry their state between calls.
#include "rp_device.h"

#include <cstddef>

namespace rp {

// ------------------------------------------------------------------------- scan
// The partial-detection / countdown state machine of src/detector.rs:377-454 with
// reset() of :290-302, one lane per stream, over precomputed window scores.  Frame f
// is emitted while chunk c = f/3 + 1 is processed; after an emit the extractor and the
// window are cleared, the rest of that chunk's frames are dropped (find_map, :372-375),
// chunk c+1 only refills the extractor, so the next frame seen is 3*(f/3) + 6.  In general, with fpf frames per
// input frame (4 behind the 11.025 / 22.05 kHz resampler): frame f's last shift f+3 lies in chunk c = (f+3)/fpf, the
// refill starts with shift fpf*(c+1) and its fourth shift completes frame fpf*(c+1).
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

__global__ __launch_bounds__(64) void scan_kernel(ScanWakewords ww, const float *__restrict__ vad_value, float vad_mode_value,
                                                  size_t S, size_t n_frames, ScanConfig cfg, BatchDetection *__restrict__ det,
                                                  int32_t *__restrict__ det_ww, int32_t *__restrict__ n_det, int max_det) {
    __shared__ float vwin[50][64];  // VadDetector::window, one column per stream (lane)
    __shared__ unsigned long long candidates;  // bit r: stream r of this block has a window that can fire
    const int lane = threadIdx.x;
    const long max_len = cfg.max_len;
    const long n_win = (long)n_frames - max_len + 1;
    // A detection needs at least one window whose aggregate passes the thresholds (run_detection :411-429;
    // the VAD only gates).  The wave first sweeps the block's 64 score rows with coalesced loads; streams
    // without such a window (all of them on non-matching audio) are done, the others run the state machine.
    // When the aggregate pass has already raised a flag per stream (ww.hot), that flag is the answer and nothing is swept.
    bool candidate = true;
    if (ww.hot) {
        const size_t sh = (size_t)blockIdx.x * 64 + lane;
        candidate = sh < S && ww.hot[sh] != 0u;
        if (candidate) ww.hot[sh] = 0u;   // consumed: the flags are zero again for the context's next call (no memset per call; Ctx::hot_flags)
    } else {
        if (lane == 0) candidates = 0;
        __syncthreads();
        const size_t s0 = (size_t)blockIdx.x * 64;
        const size_t rows = S - s0 < 64 ? S - s0 : 64;
        const size_t total = n_win > 0 ? rows * (size_t)n_win : 0;  // the block's score rows are one contiguous range
        unsigned long long mask = 0;
        for (int j = 0; j < ww.n; ++j) {
            const float *a0 = ww.agg[j] + s0 * (size_t)(n_win > 0 ? n_win : 0);
            const float *v0 = ww.avg[j] ? ww.avg[j] + s0 * (size_t)(n_win > 0 ? n_win : 0) : nullptr;
            // eight rows' loads in flight per wait (one load -> wait -> test per element made a block's 64 x n_win aggregates n_win
            // dependent round trips: 0.17 ms for 4 096 streams x 202 windows in the batched model detector)
            const float thr = ww.threshold[j], athr = ww.avg_threshold[j];
            if (v0) {
#pragma unroll 8
                for (size_t e = lane; e < total; e += 64) {
                    const float a = a0[e], v = v0[e];
                    if (a > thr && !(v < athr)) mask |= 1ull << (e / (size_t)n_win);
                }
            } else {
#pragma unroll 8
                for (size_t e = lane; e < total; e += 64)
                    if (a0[e] > thr) mask |= 1ull << (e / (size_t)n_win);
            }
        }
        if (mask) atomicOr(&candidates, mask);
        __syncthreads();
        candidate = (candidates >> lane) & 1ull;
    }
    size_t s = (size_t)blockIdx.x * 64 + lane;
    if (s >= S) return;
    // slots behind the detections a stream reports are zeroed: the output block is a function of the input alone
    auto clear_from = [&](int from) {
        BatchDetection zero{};
        for (int i = from; i < max_det; ++i) {
            det[s * (size_t)max_det + i] = zero;
            if (det_ww) det_ww[s * (size_t)max_det + i] = 0;
        }
    };
    if (!candidate) { n_det[s] = 0; clear_from(0); return; }
    const size_t row0 = s * (size_t)(n_win > 0 ? n_win : 0);
    const float *vv = vad_value ? vad_value + s * n_frames : nullptr;
    // VadDetector state (src/mfcc/vad.rs:3-50)
    int vad_index = 0, voice_countdown = 0;
    if (vv)
        for (int i = 0; i < 50; ++i) vwin[i][lane] = __builtin_nanf("");
    long win_start = 0, resume = 0;
    bool has_partial = false;
    float p_score = 0.f, p_avg = 0.f;
    int p_ww = 0;
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
                        d.stream = (int32_t)s + cfg.stream_base; d.frame = (int32_t)f; d.window = p_window; d.counter = p_counter;
                        d.avg_score = p_avg; d.score = p_score;
                        det[s * (size_t)max_det + nd] = d;
                        if (det_ww) det_ww[s * (size_t)max_det + nd] = p_ww;
                    }
                    ++nd;
                    win_start = resume = cfg.fpf * ((f + 3) / cfg.fpf + 1);  // reset()
                    if (vv) {  // vad.reset()
                        for (int i = 0; i < 50; ++i) vwin[i][lane] = __builtin_nanf("");
                        vad_index = 0; voice_countdown = 0;
                    }
                    continue;
                }
            }
        }
        // run_wakeword_detectors, src/detector.rs:433-447: every wakeword whose own thresholds pass proposes a
        // detection, the best score wins (the first of equals)
        float sc = 0.f, av = 0.f;
        int best = -1;
        for (int j = 0; j < ww.n; ++j) {
            const float sj = ww.agg[j][row0 + w];
            float aj = 0.f;
            bool pass = true;
            if (ww.avg[j]) { aj = ww.avg[j][row0 + w]; pass = !(aj < ww.avg_threshold[j]); }
            if (pass && sj > ww.threshold[j] && (best < 0 || sj > sc)) { best = j; sc = sj; av = aj; }
        }
        if (best >= 0) {
            int counter = has_partial ? p_counter + 1 : 1;
            if (!has_partial || p_score < sc) {
                p_score = sc; p_avg = av; p_window = (int)w; has_partial = true;
                p_ww = ww.label[best] ? ww.label[best][row0 + w] : best;
            }
            p_counter = counter;
            countdown = (int)(max_len / 2);
        }
    }
    n_det[s] = nd;
    clear_from(nd < max_det ? nd : max_det);
}

hipError_t launch_scan_multi(hipStream_t st, const ScanWakewords &ww, const float *vad_value, float vad_mode_value, size_t S,
                             size_t n_frames, const ScanConfig &cfg, BatchDetection *det, int32_t *det_ww, int32_t *n_det, int max_det) {
    if (S == 0) return hipSuccess;
    if (ww.n < 1 || ww.n > kScanMaxWakewords) return hipErrorInvalidValue;
    size_t blocks = (S + 63) / 64;
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)blocks), dim3(64), 0, st, ww, vad_value, vad_mode_value, S, n_frames, cfg, det, det_ww,
                       n_det, max_det);
    return hipGetLastError();
}

hipError_t launch_scan(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                       size_t S, size_t n_frames, const ScanConfig &cfg, BatchDetection *det, int32_t *n_det, int max_det,
                       uint32_t *hot) {
    ScanWakewords ww{};
    ww.n = 1;
    ww.hot = hot;
    ww.agg[0] = agg; ww.avg[0] = cfg.avg_enabled ? avg : nullptr;
    ww.threshold[0] = cfg.threshold; ww.avg_threshold[0] = cfg.avg_threshold;
    return launch_scan_multi(st, ww, vad_value, vad_mode_value, S, n_frames, cfg, det, nullptr, n_det, max_det);
}

// ------------------------------------------------------------- streaming batches
// State of one live stream between rp_stream_batch_process calls: the detector's countdown / partial
// detection / window bookkeeping (src/detector.rs:62-79) in absolute frame numbers, and the VadDetector.
struct StreamState {
    long long win_start, resume;
    int has_partial, p_counter, countdown, vad_index, voice_countdown;
    int p_ww;        // wakeword the partial detection belongs to (detectors that hold several)
    int p_label;     // its label index when that wakeword is a model, else -1
    int pad;
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
    z.p_ww = 0; z.p_label = -1;
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
// f0 + i; the window ending at it is row i of every wakeword's agg / avg (the history prefix is max_len-1 frames long).
// Several wakewords as in scan_kernel (run_wakeword_detectors, src/detector.rs:433-447).
__global__ __launch_bounds__(64) void scan_stream_kernel(ScanWakewords ww, const float *__restrict__ vad_value, float vad_mode_value, size_t S,
                                                         long long f0, int n_new, ScanConfig cfg, StreamState *__restrict__ state,
                                                         BatchDetection *__restrict__ det, int32_t *__restrict__ det_ww,
                                                         int32_t *__restrict__ det_label, int32_t *__restrict__ n_det, int max_det) {
    __shared__ float vwin[50][64];
    const int lane = threadIdx.x;
    const size_t s = (size_t)blockIdx.x * 64 + lane;
    if (s >= S) return;
    // the 64-byte head of the state travels every call, the 200 bytes of the VAD window only in detectors that have a VAD (as one struct
    // copy each way a call moved 34 MB in 264-byte strides for 65 536 streams: 0.019 of a 0.32 ms live call)
    constexpr size_t kHead = offsetof(StreamState, vad_window);
    static_assert(kHead == 64, "StreamState head");
    StreamState *sp = state + s;
    StreamState z;
    __builtin_memcpy(&z, sp, kHead);
    const long long max_len = cfg.max_len;
    const size_t row0 = s * (size_t)n_new;
    const float *vv = vad_value ? vad_value + s * (size_t)n_new : nullptr;
    if (vv)
        for (int i = 0; i < 50; ++i) vwin[i][lane] = sp->vad_window[i];
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
                        if (det_ww) det_ww[s * (size_t)max_det + nd] = z.p_ww;
                        if (det_label) det_label[s * (size_t)max_det + nd] = z.p_label;
                    }
                    ++nd;
                    z.win_start = z.resume = cfg.fpf * ((f + 3) / cfg.fpf + 1);
                    if (vv) { for (int j = 0; j < 50; ++j) vwin[j][lane] = __builtin_nanf(""); z.vad_index = 0; z.voice_countdown = 0; }
                    continue;
                }
            }
        }
        // every wakeword whose own thresholds pass proposes a detection, the best score wins (the first of equals)
        float sc = 0.f, av = 0.f;
        int best = -1;
        for (int j = 0; j < ww.n; ++j) {
            const float sj = ww.agg[j][row0 + i];
            float aj = 0.f;
            bool pass = true;
            if (ww.avg[j]) { aj = ww.avg[j][row0 + i]; pass = !(aj < ww.avg_threshold[j]); }
            if (pass && sj > ww.threshold[j] && (best < 0 || sj > sc)) { best = j; sc = sj; av = aj; }
        }
        if (best >= 0) {
            const int counter = z.has_partial ? z.p_counter + 1 : 1;
            if (!z.has_partial || z.p_score < sc) {
                z.p_score = sc; z.p_avg = av; z.p_window = f - max_len + 1; z.has_partial = 1;
                z.p_ww = best; z.p_label = ww.label[best] ? ww.label[best][row0 + i] : -1;
            }
            z.p_counter = counter;
            z.countdown = (int)(max_len / 2);
        }
    }
    if (vv)
        for (int i = 0; i < 50; ++i) sp->vad_window[i] = vwin[i][lane];
    __builtin_memcpy(sp, &z, kHead);
    n_det[s] = nd;
    for (int i = nd; i < max_det; ++i) {
        det[s * (size_t)max_det + i] = BatchDetection{};
        if (det_ww) det_ww[s * (size_t)max_det + i] = 0;
        if (det_label) det_label[s * (size_t)max_det + i] = -1;   // "no label", as the header says and the single-wakeword path fills
    }
}

hipError_t launch_scan_stream_multi(hipStream_t st, const ScanWakewords &ww, const float *vad_value, float vad_mode_value, size_t S,
                                    long long f0, int n_new, const ScanConfig &cfg, void *state, BatchDetection *det, int32_t *det_ww,
                                    int32_t *det_label, int32_t *n_det, int max_det) {
    if (S == 0) return hipSuccess;
    if (ww.n < 1 || ww.n > kScanMaxWakewords) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scan_stream_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, ww, vad_value, vad_mode_value, S, f0, n_new, cfg,
                       static_cast<StreamState *>(state), det, det_ww, det_label, n_det, max_det);
    return hipGetLastError();
}

hipError_t launch_scan_stream(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                              size_t S, long long f0, int n_new, const ScanConfig &cfg, void *state, BatchDetection *det,
                              int32_t *n_det, int max_det) {
    ScanWakewords ww{};
    ww.n = 1;
    ww.agg[0] = agg; ww.avg[0] = cfg.avg_enabled ? avg : nullptr;
    ww.threshold[0] = cfg.threshold; ww.avg_threshold[0] = cfg.avg_threshold;
    return launch_scan_stream_multi(st, ww, vad_value, vad_mode_value, S, f0, n_new, cfg, state, det, nullptr, nullptr, n_det, max_det);
}

}  // namespace rp
