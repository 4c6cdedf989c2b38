// rp_capi.cpp -- extern "C" surface declared in include/rustpotter_hip.h.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <system_error>
#include <chrono>
#include <thread>

#include "rp_host.h"

namespace rp { const std::string &last_error(); }

using namespace rp;

struct rp_detector {
    std::unique_ptr<Rustpotter> impl;
    // storage backing the last rp_detection handed out
    Detection last;
    std::vector<const char *> name_ptrs;
};
struct rp_ctx { std::unique_ptr<Ctx> impl; };
struct rp_templates { std::unique_ptr<Templates> impl; };
struct rp_model { std::unique_ptr<Model> impl; };
// one wakeword of a live-stream batch that holds several (rp_stream_batch_new_multi): a reference or a model
struct StreamWakeword {
    const Templates *t = nullptr;
    const Model *m = nullptr;
    int none_index = -1, precision = 0;
    float threshold = 0.f, avg_threshold = 0.f;   // the wakeword's own values (the config's where it has none)
    DevBuf agg, avg, label;                       // [S][frames per call] of this wakeword
};

struct rp_stream_batch {
    Ctx *c = nullptr;
    const Templates *t = nullptr;                 // the one wakeword reference of rp_stream_batch_new; nullptr with `ww`
    std::vector<std::unique_ptr<StreamWakeword>> ww;  // rp_stream_batch_new_multi: 1..8 wakewords
    int K = 0, max_len = 0, Tmax = 1;             // mfcc_size, max_mfcc_frames (longest wakeword), most templates of a reference
    DevBuf det_ww, det_label, logits, mean, xrows, xs2;
    rp_detector_config cfg{};
    size_t S = 0, max_chunks = 0, chunks_seen = 0, hist_frames = 0;
    bool poisoned = false;   // a launch failed after part of the persistent state had advanced
    // MFCC window: rows of `cap` frames; a call appends its frames behind the `fill` valid ones and only when a row
    // is full are the last max_len-1 frames moved to the front of the other buffer
    int cur = 0;
    size_t cap = 0, fill = 0;
    // previous chunk | new chunks (f32), ping-pong so that one kernel both carries the old chunk and decodes the new
    int pcur = 0;
    size_t last_off = 0;     // where the last chunk of the previous call sits in pcm[pcur]'s rows
    DevBuf pcm[2], mfcc[2], state, scores, agg, avg, vad, list;
    // AudioEncoder of the streams (src/audio/encoder.rs): channel count and, for input that is not 16 kHz, the
    // resampler plan with every stream's previous input frame
    int channels = 1;
    size_t in_len = 480;
    size_t out_len = 480;    // encoded (16 kHz) samples per input frame: 480, or 640 for the 11.025 / 22.05 kHz family
    size_t fpf() const { return out_len / 160; }  // MFCC frames a stream gains per input frame (3 or 4)
    const Resampler *rs = nullptr;
    DevBuf rs_prev[2], rs_xs, rs_out;
    int rs_cur = 0;
};

static void fill_detection(rp_detector *d, const Detection &src, rp_detection *out) {
    d->last = src;
    d->name_ptrs.clear();
    for (auto &s : d->last.score_names) d->name_ptrs.push_back(s.c_str());
    out->name = d->last.name.c_str();
    out->avg_score = d->last.avg_score;
    out->score = d->last.score;
    out->n_scores = d->last.scores.size();
    out->score_names = d->name_ptrs.data();
    out->scores = d->last.scores.data();
    out->counter = d->last.counter;
    out->gain = d->last.gain;
}

template <class F> static int guarded(F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { set_last_error("out of host memory"); return -1; }
    catch (const std::exception &e) { set_last_error(e.what()); return -1; }
    catch (...) { set_last_error("unknown error"); return -1; }
}

extern "C" {

const char *rp_last_error(void) { return last_error().c_str(); }
const char *rp_version(void) { return "rustpotter_hip 0.1.0 (gfx950; mirrors rustpotter 3.0.2)"; }

void rp_config_default(rp_config *c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->fmt.sample_rate = 16000;  // DETECTOR_INTERNAL_SAMPLE_RATE
    c->fmt.sample_format = RP_SAMPLE_F32;
    c->fmt.channels = 1;
    c->fmt.endianness = RP_ENDIAN_LITTLE;
    c->detector.avg_threshold = 0.2f;  // src/constants.rs:4
    c->detector.threshold = 0.5f;      // :5
    c->detector.min_scores = 5;        // :6
    c->detector.eager = false;
    c->detector.score_ref = 0.22f;     // :7
    c->detector.band_size = 5;         // :3
    c->detector.score_mode = RP_SCORE_MAX;
    c->detector.vad_mode = RP_VAD_NONE;
    c->filters.gain_normalizer.enabled = false;
    c->filters.gain_normalizer.has_gain_ref = false;
    c->filters.gain_normalizer.min_gain = 0.1f;
    c->filters.gain_normalizer.max_gain = 1.0f;
    c->filters.band_pass.enabled = false;
    c->filters.band_pass.low_cutoff = 80.f;
    c->filters.band_pass.high_cutoff = 400.f;
}

int rp_new(const rp_config *config, rp_detector **out) {
    return guarded([&]() -> int {
        if (!config || !out) { set_last_error("null argument"); return -1; }
        *out = nullptr;
        std::unique_ptr<Rustpotter> r(Rustpotter::create(*config));
        if (!r) return -1;
        rp_detector *d = new rp_detector();
        d->impl = std::move(r);
        *out = d;
        return 0;
    });
}
void rp_free(rp_detector *d) { delete d; }

int rp_add_wakeword_from_buffer(rp_detector *d, const char *key, const uint8_t *buffer, size_t len) {
    if (!d || !key || (!buffer && len)) { set_last_error("null argument"); return -1; }
    return guarded([&]() -> int { return d->impl->add_wakeword_from_buffer(key, buffer, len) ? 0 : -1; });
}
int rp_add_wakeword_from_file(rp_detector *d, const char *key, const char *path) {
    if (!d || !key || !path) { set_last_error("null argument"); return -1; }
    return guarded([&]() -> int { return d->impl->add_wakeword_from_file(key, path) ? 0 : -1; });
}
bool rp_remove_wakeword(rp_detector *d, const char *key) { return d && key && d->impl->remove_wakeword(key); }
bool rp_remove_wakewords(rp_detector *d) { return d && d->impl->remove_wakewords(); }
size_t rp_get_samples_per_frame(const rp_detector *d) { return d ? d->impl->get_samples_per_frame() : 0; }
size_t rp_get_bytes_per_frame(const rp_detector *d) { return d ? d->impl->get_bytes_per_frame() : 0; }
int rp_get_partial_detection(const rp_detector *d, rp_detection *out) {
    if (!d || !out) { set_last_error("null argument"); return -1; }
    const Detection *p = d->impl->get_partial_detection();
    if (!p) return 0;
    fill_detection(const_cast<rp_detector *>(d), *p, out);
    return 1;
}
float rp_get_rms_level(const rp_detector *d) { return d ? d->impl->get_rms_level() : 0.f; }
float rp_get_gain(const rp_detector *d) { return d ? d->impl->get_gain() : 0.f; }
float rp_get_rms_level_ref(const rp_detector *d) { return d ? d->impl->get_rms_level_ref() : 0.f; }

#define RP_PROCESS(call)                                              \
    if (!d) { set_last_error("null handle"); return -1; }             \
    return guarded([&]() -> int {                                     \
        Detection det;                                                \
        int r = (call);                                               \
        if (r == 1 && out) fill_detection(d, det, out);               \
        return r;                                                     \
    })

int rp_process_bytes(rp_detector *d, const uint8_t *b, size_t len, rp_detection *out) { RP_PROCESS(d->impl->process_bytes(b, len, &det)); }
int rp_process_samples_i8(rp_detector *d, const int8_t *s, size_t n, rp_detection *out) { RP_PROCESS(d->impl->process_samples<int8_t>(s, n, &det)); }
int rp_process_samples_i16(rp_detector *d, const int16_t *s, size_t n, rp_detection *out) { RP_PROCESS(d->impl->process_samples<int16_t>(s, n, &det)); }
int rp_process_samples_i32(rp_detector *d, const int32_t *s, size_t n, rp_detection *out) { RP_PROCESS(d->impl->process_samples<int32_t>(s, n, &det)); }
int rp_process_samples_f32(rp_detector *d, const float *s, size_t n, rp_detection *out) { RP_PROCESS(d->impl->process_samples<float>(s, n, &det)); }

int rp_update_config(rp_detector *d, const rp_config *c) {
    if (!d || !c) { set_last_error("null argument"); return -1; }
    d->impl->update_detector_config(c->detector);
    d->impl->update_filters_config(c->filters);
    return 0;
}
int rp_update_detector_config(rp_detector *d, const rp_detector_config *c) {
    if (!d || !c) { set_last_error("null argument"); return -1; }
    d->impl->update_detector_config(*c);
    return 0;
}
int rp_update_filters_config(rp_detector *d, const rp_filters_config *c) {
    if (!d || !c) { set_last_error("null argument"); return -1; }
    d->impl->update_filters_config(*c);
    return 0;
}
void rp_reset(rp_detector *d) { if (d) d->impl->reset(); }

// ------------------------------------------------------------------- batched level
int rp_ctx_new(int device, int flags, rp_ctx **out) {
    return guarded([&]() -> int {
        if (!out) { set_last_error("null argument"); return -1; }
        *out = nullptr;
        if ((flags & RP_CTX_ARITH_STRICT_F32) && (flags & RP_CTX_ARITH_FAST_SPLIT)) { set_last_error("RP_CTX_ARITH_STRICT_F32 and RP_CTX_ARITH_FAST_SPLIT exclude each other"); return -1; }
        std::unique_ptr<Ctx> c(Ctx::create(device, flags));
        if (!c) return -1;
        c->arith.mode = (flags & RP_CTX_ARITH_STRICT_F32) ? kArithStrictF32 : (flags & RP_CTX_ARITH_FAST_SPLIT) ? kArithFastSplit : kArithF32Matrix;
        c->arith.ragged = (flags & RP_CTX_RAGGED_MATRIX) ? 1 : 0;
        rp_ctx *h = new rp_ctx();
        h->impl = std::move(c);
        *out = h;
        return 0;
    });
}
void rp_ctx_free(rp_ctx *ctx) { delete ctx; }
int rp_ctx_set_stream(rp_ctx *ctx, void *s) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    ctx->impl->stream = s ? static_cast<hipStream_t>(s) : ctx->impl->own_stream;
    return 0;
}
int rp_ctx_synchronize(rp_ctx *ctx) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    if (!hip_ok(hipSetDevice(ctx->impl->device), "hipSetDevice")) return -1;
    return hip_ok(hipStreamSynchronize(ctx->impl->stream), "hipStreamSynchronize") ? 0 : -1;
}

static_assert(RP_ARITH_F32_MATRIX == kArithF32Matrix && RP_ARITH_STRICT_F32 == kArithStrictF32 && RP_ARITH_FAST_SPLIT == kArithFastSplit, "RP_ARITH_* mirror rp_kernels.h");
int rp_ctx_set_arithmetic(rp_ctx *ctx, int arith, int ragged_matrix) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    if (arith != RP_ARITH_F32_MATRIX && arith != RP_ARITH_STRICT_F32 && arith != RP_ARITH_FAST_SPLIT) { set_last_error("unknown RP_ARITH_* value"); return -1; }
    ctx->impl->arith.mode = arith;
    ctx->impl->arith.ragged = ragged_matrix ? 1 : 0;
    return 0;
}
int rp_ctx_arithmetic(rp_ctx *ctx, int *ragged_matrix) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    if (ragged_matrix) *ragged_matrix = ctx->impl->arith.ragged;
    return ctx->impl->arith.mode;
}

int rp_ctx_dtw_ref_pairs(rp_ctx *ctx, uint64_t *pairs) {
    if (!ctx || !pairs) { set_last_error("null argument"); return -1; }
    Ctx *c = ctx->impl.get();
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return -1;
    unsigned long long v = 0;
    if (!hip_ok(hipMemcpy(&v, dtw_fix_stats(c->dtw_work().fix), sizeof(v), hipMemcpyDeviceToHost), "hipMemcpy(dtw stats)")) return -1;
    *pairs = (uint64_t)v;
    return 0;
}

#ifndef RP_BUILD_ARCH
#define RP_BUILD_ARCH "gfx950"
#endif
#ifndef RP_BUILD_FLAGS_EXTRA
#define RP_BUILD_FLAGS_EXTRA ""
#endif
const char *rp_build_info(void) { return sizeof(RP_BUILD_FLAGS_EXTRA) > 1 ? RP_BUILD_ARCH " +" RP_BUILD_FLAGS_EXTRA : RP_BUILD_ARCH; }

#ifdef RP_MFMA_TRACE   // variant builds only: read the DTW counter block back (tools/r4_mfma_timeline.py)
extern "C" int rp_debug_read_dtw_work(rp_ctx *ctx, uint32_t *dst, size_t words) {
    Ctx *c = ctx->impl.get();
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipStreamSynchronize(c->stream), "sync")) return -1;
    return hip_ok(hipMemcpy(dst, c->dtw_work().sched, words * 4, hipMemcpyDeviceToHost), "hipMemcpy") ? 0 : -1;
}
#endif

int rp_ctx_dtw_kernels(rp_ctx *ctx) {
    if (!ctx) return 0;
    const int m = (int)ctx->impl->dtw_ran;
    ctx->impl->dtw_ran = 0;
    return m;
}

const char *rp_ctx_last_mlp_kernel(rp_ctx *ctx) { return ctx ? ctx->impl->last_mlp_kernel.c_str() : ""; }

size_t rp_mfcc_num_frames(size_t n_samples) {
    size_t chunks = n_samples / 480;
    return chunks >= 1 ? 3 * chunks - 3 : 0;
}

namespace {
struct Staged {  // host<->device staging for RP_CTX_HOST_POINTERS
    Ctx *c;
    bool host;
    explicit Staged(Ctx *ctx) : c(ctx), host((ctx->flags & RP_CTX_HOST_POINTERS) != 0) {}
    const void *in(const void *p, size_t bytes, DevBuf &buf) {
        if (!host || !p) return p;
        if (!buf.reserve(bytes)) return nullptr;
        if (!hip_ok(hipMemcpyAsync(buf.p, p, bytes, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync(H2D)")) return nullptr;
        return buf.p;
    }
    void *out(void *p, size_t bytes, DevBuf &buf) {
        if (!host || !p) return p;
        return buf.reserve(bytes) ? buf.p : nullptr;
    }
    bool back(void *host_p, const void *dev_p, size_t bytes) {
        if (!host || !host_p) return true;
        return hip_ok(hipMemcpyAsync(host_p, dev_p, bytes, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync(D2H)");
    }
    bool finish() { return !host || hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize"); }
};
}  // namespace

static size_t sample_bytes(rp_sample_format f) { return f == RP_SAMPLE_I8 ? 1 : f == RP_SAMPLE_I16 ? 2 : 4; }

int rp_mfcc_batch_fmt(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                      int K, float *mfcc) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (pcm_stride < n_samples) { set_last_error("pcm_stride smaller than n_samples"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        const MfccTablesDev *tb = c->tables_for(K);
        if (!tb) return -1;
        const size_t nf = rp_mfcc_num_frames(n_samples);
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        float *dm = static_cast<float *>(sg.out(mfcc, S * nf * K * sizeof(float), c->stage_out));
        if ((S && nf) && (!dp || !dm)) return -1;
        c->time_begin(kKernelMfcc);
        bool ok = hip_ok(launch_mfcc_fmt(c->stream, *tb, dp, (int)fmt, S, n_samples, pcm_stride, 0, nf, nf, dm), "mfcc_kernel");
        c->time_end();
        if (!ok) return -1;
        if (!sg.back(mfcc, dm, S * nf * K * sizeof(float)) || !sg.finish()) return -1;
        return 0;
    });
}

int rp_mfcc_batch(rp_ctx *ctx, const float *pcm, size_t S, size_t n_samples, size_t pcm_stride, int K, float *mfcc) {
    return rp_mfcc_batch_fmt(ctx, pcm, RP_SAMPLE_F32, S, n_samples, pcm_stride, K, mfcc);
}

int rp_wakeword_ref_build(rp_ctx *ctx, const char *name, const float *threshold, const float *avg_threshold, size_t n,
                          const char *const *sample_names, const uint8_t *const *wav_buffers, const size_t *wav_lens,
                          uint16_t mfcc_size, int rms_from_files, uint8_t **out_rpw, size_t *out_len) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        *out_rpw = nullptr; *out_len = 0;
        WakewordRefData r;
        if (!build_wakeword_ref(ctx->impl.get(), name, threshold, avg_threshold, n, sample_names, wav_buffers, wav_lens,
                                (int)mfcc_size, rms_from_files != 0, &r)) return -1;
        std::vector<uint8_t> bytes = serialize_wakeword_ref(r);
        uint8_t *p = static_cast<uint8_t *>(std::malloc(bytes.size()));
        if (!p) { set_last_error("out of host memory"); return -1; }
        std::memcpy(p, bytes.data(), bytes.size());
        *out_rpw = p; *out_len = bytes.size();
        return 0;
    });
}
void rp_buffer_free(uint8_t *buffer) { std::free(buffer); }

int rp_wakeword_model_train(rp_ctx *ctx, const rp_train_options *options, size_t n_train, const char *const *train_names,
                            const uint8_t *const *train_wavs, const size_t *train_lens, size_t n_test,
                            const char *const *test_names, const uint8_t *const *test_wavs, const size_t *test_lens,
                            const uint8_t *prev_model, size_t prev_model_len, uint8_t **out_rpw, size_t *out_len,
                            float *final_loss, float *test_accuracy) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        *out_rpw = nullptr; *out_len = 0;
        WakewordModelData prev, m;
        bool has_prev = false;
        if (prev_model) {
            RpwKind kind; WakewordRefData ref; std::string err;
            if (!parse_rpw(prev_model, prev_model_len, &kind, &ref, &prev, &err)) { set_last_error(err); return -1; }
            if (kind != RpwKind::Model) { set_last_error("the file to train from is not a wakeword model"); return -1; }
            has_prev = true;
        }
        if (!train_wakeword_model(ctx->impl.get(), *options, n_train, train_names, train_wavs, train_lens, n_test, test_names, test_wavs,
                                  test_lens, has_prev ? &prev : nullptr, &m, final_loss, test_accuracy)) return -1;
        std::vector<uint8_t> bytes = serialize_wakeword_model(m);
        uint8_t *p = static_cast<uint8_t *>(std::malloc(bytes.size()));
        if (!p) { set_last_error("out of host memory"); return -1; }
        std::memcpy(p, bytes.data(), bytes.size());
        *out_rpw = p; *out_len = bytes.size();
        return 0;
    });
}

int rp_frontend_batch(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                      const rp_filters_config *filters, float rms_level_ref, size_t window_size, float *pcm_out,
                      size_t out_stride, float *rms, float *gains) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (pcm_stride < n_samples || out_stride < n_samples) { set_last_error("stride smaller than n_samples"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        if (!filters) { set_last_error("null argument"); return -1; }
        const rp_gain_normalization_config &g = filters->gain_normalizer;
        const rp_band_pass_config &b = filters->band_pass;
        if (g.enabled && g.has_gain_ref) rms_level_ref = g.gain_ref;  // fixed_rms_level, gain_normalizer_filter.rs:56-66
        if (window_size == 0) window_size = 1;                           // set_rms_level_ref :47
        if (window_size > 1u << 20) { set_last_error("window_size too large"); return -1; }
        // BandPassFilter::new, band_pass_filter.rs:31-55 (f32, sample rate 16 kHz)
        float a0 = 0, a1 = 0, a2 = 0, b1 = 0, b2 = 0;
        if (b.enabled) {
            const float kPi = 3.14159274101257324f, sample_rate = 16000.f;
            const float omega_low = 2.0f * kPi * b.low_cutoff / sample_rate, omega_high = 2.0f * kPi * b.high_cutoff / sample_rate;
            const float cos_low = std::cos(omega_low), cos_high = std::cos(omega_high);
            const float alpha_low = std::sin(omega_low) / 2.0f, alpha_high = std::sin(omega_high) / 2.0f;
            a0 = 1.0f / (1.0f + alpha_high - alpha_low);
            a1 = -2.0f * cos_low * a0; a2 = (1.0f - alpha_high - alpha_low) * a0;
            b1 = -2.0f * cos_high * a0; b2 = (1.0f - alpha_high + alpha_low) * a0;
        }
        const size_t n_chunks = n_samples / 480;
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        float *dout = static_cast<float *>(sg.out(pcm_out, S * out_stride * sizeof(float), c->stage_out));
        if (S && (!dp || !dout)) return -1;
        if (!c->ws_rms.reserve(S * n_chunks * 4 + 16) || !c->ws_gain.reserve(S * n_chunks * 4 + 16) ||
            !c->ws_ring.reserve(S * window_size * 4 + 16)) return -1;
        if (!hip_ok(launch_frontend(c->stream, dp, (int)fmt, S, n_samples, pcm_stride, g.enabled ? 1 : 0, rms_level_ref, g.min_gain,
                                    g.max_gain, (int)window_size, b.enabled ? 1 : 0, a0, a1, a2, b1, b2, c->ws_ring.as<float>(),
                                    c->ws_rms.as<float>(), c->ws_gain.as<float>(), dout, out_stride), "front-end kernels")) return -1;
        auto copy_out = [&](float *dst, const float *src_dev, bool valid) {
            if (!dst) return true;
            if (!valid) {  // gain filter off: every chunk has gain 1
                std::vector<float> ones(S * n_chunks, 1.f);
                return hip_ok(hipMemcpyAsync(dst, ones.data(), ones.size() * 4, sg.host ? hipMemcpyHostToHost : hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync") &&
                       hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
            }
            return hip_ok(hipMemcpyAsync(dst, src_dev, S * n_chunks * 4, sg.host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
        };
        if (!copy_out(rms, c->ws_rms.as<float>(), true) || !copy_out(gains, c->ws_gain.as<float>(), g.enabled)) return -1;
        if (!sg.back(pcm_out, dout, S * out_stride * sizeof(float))) return -1;
        return hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? 0 : -1;
    });
}

int rp_templates_new(rp_ctx *ctx, int T, int K, const int *lens, const float *feats, int avg_len, const float *avg,
                     rp_templates **out) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        *out = nullptr;
        std::unique_ptr<Templates> t(Templates::create(ctx->impl.get(), T, K, lens, feats, avg_len, avg));
        if (!t) return -1;
        rp_templates *h = new rp_templates();
        h->impl = std::move(t);
        *out = h;
        return 0;
    });
}
void rp_templates_free(rp_templates *t) { delete t; }
int rp_templates_max_len(const rp_templates *t) { return t ? t->impl->dev.max_len : 0; }

int rp_dtw_score_batch(rp_ctx *ctx, const float *mfcc, size_t S, size_t n_frames, const rp_templates *t,
                       float score_ref, int band_size, rp_score_mode score_mode, int with_avg, float *scores,
                       float *avg, float *agg) {
    return guarded([&]() -> int {
        if (!ctx || !t) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        const TemplatesDev &td = t->impl->dev;
        if (n_frames < (size_t)td.max_len) return 0;  // no complete window
        if (band_size < 0) { set_last_error("band_size must be >= 0"); return -1; }  // 0: no cell is in the band, every score is 0 (dtw.rs:64-75)
        const size_t n_win = n_frames - td.max_len + 1;
        const bool do_avg = with_avg && td.has_avg;
        if (do_avg && !avg) { set_last_error("avg output required when with_avg is set"); return -1; }
        Staged sg(c);
        const size_t rows = S * n_win;
        const float *dm = static_cast<const float *>(sg.in(mfcc, S * n_frames * td.K * sizeof(float), c->stage_in));
        float *ds = static_cast<float *>(sg.out(scores, rows * td.T * sizeof(float), c->stage_out));
        float *da = do_avg ? static_cast<float *>(sg.out(avg, rows * sizeof(float), c->stage_out2)) : nullptr;
        float *dg = agg ? static_cast<float *>(sg.out(agg, rows * sizeof(float), c->stage_out3)) : nullptr;
        if (rows && (!dm || !ds)) return -1;
        c->time_begin(kKernelDtw);
        bool ok = hip_ok(launch_dtw(c->stream, c->dtw_work_for(S, rows * (size_t)(td.rag_count > 1 ? td.rag_count : 1)), td, dm, S, n_frames, 0, n_win, n_win, band_size, score_ref, do_avg ? 1 : 0, ds, da), "dtw kernel");
        c->time_end();
        if (!ok) return -1;
        if (dg) {
            c->time_begin(kKernelAggregate);
            ok = hip_ok(launch_aggregate(c->stream, ds, rows, td.T, (int)score_mode, dg), "aggregate_kernel");
            c->time_end();
            if (!ok) return -1;
        }
        if (!sg.back(scores, ds, rows * td.T * sizeof(float)) || (do_avg && !sg.back(avg, da, rows * sizeof(float))) ||
            (dg && !sg.back(agg, dg, rows * sizeof(float))) || !sg.finish())
            return -1;
        return 0;
    });
}

static float vad_mode_value(rp_vad_mode m) { return m == RP_VAD_EASY ? 2.f : m == RP_VAD_MEDIUM ? 2.5f : 3.f; }  // src/config.rs:140-146

int rp_detect_scan(rp_ctx *ctx, const float *agg, const float *avg, size_t S, size_t n_frames, int max_len,
                   const rp_detector_config *config, int avg_enabled, const float *mfcc, int K,
                   rp_batch_detection *det, int32_t *n_det, int max_det) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        if (!config || (S && (!agg || !det || !n_det))) { set_last_error("null argument"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        const bool vad = config->vad_mode != RP_VAD_NONE;
        if (vad && (!mfcc || K < 1)) { set_last_error("rp_detect_scan: vad_mode needs the MFCC frames"); return -1; }
        static_assert(sizeof(rp_batch_detection) == sizeof(BatchDetection), "layout");
        ScanConfig sc;
        sc.threshold = config->threshold; sc.avg_threshold = config->avg_threshold; sc.min_scores = (int)config->min_scores;
        sc.eager = config->eager ? 1 : 0; sc.max_len = max_len; sc.avg_enabled = (avg_enabled && avg) ? 1 : 0;
        const size_t n_win = n_frames >= (size_t)max_len ? n_frames - max_len + 1 : 0;
        Staged sg(c);
        const float *dg = static_cast<const float *>(sg.in(agg, S * n_win * sizeof(float), c->stage_in));
        const float *da = sc.avg_enabled ? static_cast<const float *>(sg.in(avg, S * n_win * sizeof(float), c->stage_out3)) : nullptr;
        BatchDetection *dd = static_cast<BatchDetection *>(sg.out(det, S * (size_t)max_det * sizeof(BatchDetection), c->stage_out));
        int32_t *dn = static_cast<int32_t *>(sg.out(n_det, S * sizeof(int32_t), c->stage_out2));
        float *dv = nullptr;
        if (vad) {
            const float *dm = static_cast<const float *>(sg.in(mfcc, S * n_frames * K * sizeof(float), c->ws_mfcc));
            if (!c->ws_vad.reserve(S * n_frames * sizeof(float) + 16) || (S * n_frames && !dm)) return -1;
            dv = c->ws_vad.as<float>();
            if (!hip_ok(launch_vad_value(c->stream, dm, S * n_frames, K, dv), "vad_value_kernel")) return -1;
        }
        c->time_begin(kKernelScan);
        bool ok = hip_ok(launch_scan(c->stream, dg, da, dv, vad_mode_value(config->vad_mode), S, n_frames, sc, dd, dn, max_det), "scan_kernel");
        c->time_end();
        if (!ok) return -1;
        if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t)) || !sg.finish()) return -1;
        return 0;
    });
}

int rp_batch_detect(rp_ctx *ctx, const float *pcm, size_t S, size_t n_samples, size_t pcm_stride, const rp_templates *t,
                    const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det, int max_det,
                    float *scores, float *agg) {
    return rp_batch_detect_fmt(ctx, pcm, RP_SAMPLE_F32, S, n_samples, pcm_stride, t, config, det, n_det, max_det, scores, agg);
}

// The body of rp_batch_detect_fmt.  gather (rp_batch_detect_sharded): the detections of this shard are reported with
// stream ids starting at stream_base and, instead of going to `det` / `n_det` directly, are copied from this context's
// buffers into the gathered block `det` / `n_det` (rows stream_base..) that lives in host memory (gather_host) or on
// device gather_device (peer copy over xGMI).
struct GatherTo { bool on = false, host = true; int device = 0; int stream_base = 0; bool device_pcm = false; };   // device_pcm: pcm is a device pointer whatever the context's flags say (rp_batch_detect_ingest)
static int batch_detect_impl(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                             const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                             int max_det, float *scores, float *agg, const GatherTo &gather);

int rp_batch_detect_fmt(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                        const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                        int max_det, float *scores, float *agg) {
    return batch_detect_impl(ctx, pcm, fmt, S, n_samples, pcm_stride, t, config, det, n_det, max_det, scores, agg, GatherTo{});
}

static int batch_detect_impl(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                             const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                             int max_det, float *scores, float *agg, const GatherTo &gather) {
    return guarded([&]() -> int {
        if (!ctx || !t) { set_last_error("null handle"); return -1; }
        if (!config || (S && (!pcm || !det || !n_det))) { set_last_error("null argument"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (pcm_stride < n_samples) { set_last_error("pcm_stride smaller than n_samples"); return -1; }
        const TemplatesDev &td = t->impl->dev;
        const MfccTablesDev *tb = c->tables_for(td.K);
        if (!tb) return -1;
        const size_t nf = rp_mfcc_num_frames(n_samples);
        const size_t n_win = nf >= (size_t)td.max_len ? nf - td.max_len + 1 : 0;
        const size_t rows = S * n_win;
        const bool do_avg = td.has_avg && config->avg_threshold != 0.f;  // wakeword_comp.rs:85
        Staged sg(c);
        if (gather.device_pcm) sg.host = false;
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        BatchDetection *dd = static_cast<BatchDetection *>(sg.out(det, S * (size_t)max_det * sizeof(BatchDetection), c->stage_out));
        int32_t *dn = static_cast<int32_t *>(sg.out(n_det, S * sizeof(int32_t), c->stage_out2));
        if (gather.on) {  // results are produced in this context's own buffers and copied into the gathered block below
            if (!c->stage_out.reserve(S * (size_t)max_det * sizeof(BatchDetection) + 16) || !c->stage_out2.reserve(S * sizeof(int32_t) + 16)) return -1;
            dd = c->stage_out.as<BatchDetection>(); dn = c->stage_out2.as<int32_t>();
        }
        // caller-provided score arrays are used directly when they are device pointers
        float *ds = (scores && !sg.host) ? scores : nullptr, *dg = (agg && !sg.host) ? agg : nullptr;
        if (!c->ws_mfcc.reserve(S * nf * td.K * sizeof(float) + 64 * td.K * sizeof(float))) return -1;  // slack: the list kernel's band reads past a row
        // The averaged-template gate as the reference runs it (wakeword_comp.rs:85-93): a window whose avg_score is below
        // avg_threshold is never compared with the sample templates.  Taken when the caller did not ask for the
        // per-window score arrays (those are defined for every window) and RP_CTX_FULL_SCORES is not set.
        const bool detect_only = !scores && !agg && !(c->flags & RP_CTX_FULL_SCORES);
        const bool gated = do_avg && detect_only && dtw_gate_supported(td, config->band_size, rows);
        // template sets only the generic kernel serves: the same gate at wave granularity (launch_dtw_generic_gated)
        const bool gated_generic = do_avg && detect_only && !gated && rows > 0 && dtw_uses_generic(td, config->band_size, S, n_win);
        // detect-only calls in ScoreMode::Max may also stop DTWs that can no longer reach `threshold` (rp_kernels.h, launch_dtw)
        const float abandon = (detect_only && config->score_mode == RP_SCORE_MAX) ? dtw_abandon_nc(config->threshold, config->score_ref) : __builtin_inff();
        if (gated && !c->ws_list.reserve((rows + 1) * sizeof(uint32_t) + 16)) return -1;
        if (!ds) { if (!c->ws_scores.reserve(rows * td.T * sizeof(float) + 16)) return -1; ds = c->ws_scores.as<float>(); }
        if (!dg) { if (!c->ws_agg.reserve(rows * sizeof(float) + 16)) return -1; dg = c->ws_agg.as<float>(); }
        float *da = nullptr;
        if (do_avg) { if (!c->ws_avg.reserve(rows * sizeof(float) + 16)) return -1; da = c->ws_avg.as<float>(); }
        if (S && (!dp || !dd || !dn)) return -1;
        float *dm = c->ws_mfcc.as<float>();
        c->time_begin(kKernelMfcc);
        bool ok = hip_ok(launch_mfcc_fmt(c->stream, *tb, dp, (int)fmt, S, n_samples, pcm_stride, 0, nf, nf, dm), "mfcc_kernel");
        c->time_end();
        if (!ok) return -1;
        // the aggregate pass also tells the scan which streams can fire at all (a flag per stream, zeroed here) and writes 0 for the
        // windows the averaged-template gate rejected (their `scores` rows were never written)
        AggExtra ax;
        if (n_win) {
            ax.hot = c->hot_flags(S);   // zero: the scan below puts every flag it reads back (no memset per call)
            if (!ax.hot) return -1;
            ax.threshold = config->threshold; ax.n_win = n_win;
            if (gated || gated_generic) { ax.gate_avg = da; ax.gate_threshold = config->avg_threshold; }  // only rows the gate really skipped
        }
        // ScoreMode::Max of a reference whose templates are one chunk of the matrix-core kernel, no averaged template scored: the DTW
        // kernel writes the aggregate and the flags itself (DtwFusedAgg, rp_kernels.h) and the aggregate pass is skipped
        DtwFusedAgg fz;
        if (config->score_mode == RP_SCORE_MAX && !do_avg && n_win) { fz.agg = dg; fz.hot = ax.hot; fz.threshold = config->threshold; }
        c->time_begin(kKernelDtw);
        if (gated) {
            uint32_t *lst = c->ws_list.as<uint32_t>();
            ok = hip_ok(launch_dtw_gated(c->stream, c->dtw_work(), td, dm, S, nf, 0, n_win, config->band_size, config->score_ref, config->avg_threshold,
                                         ds, da, lst + 1, lst, true, abandon), "dtw kernels (gated)");
        } else if (gated_generic) {
            ok = hip_ok(launch_dtw_generic_gated(c->stream, c->dtw_work(), td, dm, S, nf, 0, n_win, n_win, config->band_size, config->score_ref,
                                                 config->avg_threshold, ds, da), "dtw_generic_kernel (gated)");
        } else {
            // ws_mfcc ends with slack: short streams (fewer than 64 windows each) are scored by cross-stream waves like live-stream batches
            ok = hip_ok(launch_dtw(c->stream, c->dtw_work_for(S, rows * (size_t)(td.rag_count > 1 ? td.rag_count : 1)), td, dm, S, nf, 0, n_win, n_win, config->band_size, config->score_ref, do_avg ? 1 : 0, ds, da, true, abandon,
                                   fz.agg ? &fz : nullptr), "dtw kernel");
        }
        c->time_end();
        if (!ok) return -1;
        if (!fz.done) {
            c->time_begin(kKernelAggregate);
            ok = hip_ok(launch_aggregate(c->stream, ds, rows, td.T, (int)config->score_mode, dg, ax), "aggregate_kernel");
            c->time_end();
        }
        if (!ok) return -1;
        ScanConfig sc;
        sc.threshold = config->threshold; sc.avg_threshold = config->avg_threshold; sc.min_scores = (int)config->min_scores;
        sc.eager = config->eager ? 1 : 0; sc.max_len = td.max_len; sc.avg_enabled = do_avg ? 1 : 0;
        sc.stream_base = gather.stream_base;
        float *dv = nullptr;
        if (config->vad_mode != RP_VAD_NONE) {
            if (!c->ws_vad.reserve(S * nf * sizeof(float) + 16)) return -1;
            dv = c->ws_vad.as<float>();
            if (!hip_ok(launch_vad_value(c->stream, dm, S * nf, td.K, dv), "vad_value_kernel")) return -1;
        }
        c->time_begin(kKernelScan);
        ok = hip_ok(launch_scan(c->stream, dg, da, dv, vad_mode_value(config->vad_mode), S, nf, sc, dd, dn, max_det, ax.hot), "scan_kernel");
        c->time_end();
        if (!ok) return -1;
        if (gather.on) {
            // final result gather: this shard's block into the gathered arrays -- device to host, or a peer
            // copy to the gathering device (xGMI between the GPUs of a node)
            rp_batch_detection *gd = det + (size_t)gather.stream_base * (size_t)max_det;
            int32_t *gn = n_det + gather.stream_base;
            const size_t bd = S * (size_t)max_det * sizeof(BatchDetection), bn = S * sizeof(int32_t);
            if (gather.host) {
                if (!hip_ok(hipMemcpyAsync(gd, dd, bd, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync(gather)") ||
                    !hip_ok(hipMemcpyAsync(gn, dn, bn, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync(gather)")) return -1;
            } else {
                if (!hip_ok(hipMemcpyPeerAsync(gd, gather.device, dd, c->device, bd, c->stream), "hipMemcpyPeerAsync(gather)") ||
                    !hip_ok(hipMemcpyPeerAsync(gn, gather.device, dn, c->device, bn, c->stream), "hipMemcpyPeerAsync(gather)")) return -1;
            }
            return hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? 0 : -1;
        }
        if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t))) return -1;
        if (sg.host && scores && !sg.back(scores, ds, rows * td.T * sizeof(float))) return -1;
        if (sg.host && agg && !sg.back(agg, dg, rows * sizeof(float))) return -1;
        return sg.finish() ? 0 : -1;
    });
}

// how the last rp_batch_detect_sharded of this thread gathered its results (rp_sharded_gather_info)
static thread_local std::string g_sharded_info;
const char *rp_sharded_gather_info(void) { return g_sharded_info.c_str(); }

int rp_batch_detect_ingest(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                           const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                           int max_det, size_t block_streams, double *seconds) {
    return guarded([&]() -> int {
        if (!ctx || !t) { set_last_error("null handle"); return -1; }
        if (!config || (S && (!pcm || !det || !n_det))) { set_last_error("null argument"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        if (pcm_stride < n_samples) { set_last_error("pcm_stride smaller than n_samples"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t Sb = block_streams ? block_streams : 8192, row_bytes = pcm_stride * sample_bytes(fmt);
        const size_t blk = std::min(Sb, S), n_blocks = S ? (S + blk - 1) / blk : 0;
        if (S > 0x7fffffffULL) { set_last_error("rp_batch_detect_ingest: too many streams"); return -1; }
        if (n_blocks == 0) { if (seconds) *seconds = 0.0; return 0; }
        if (!c->ingest_ready() || !c->ws_ingest.reserve(2 * blk * row_bytes + 64)) return -1;
        unsigned char *dbuf[2] = {c->ws_ingest.as<unsigned char>(), c->ws_ingest.as<unsigned char>() + blk * row_bytes};
        auto streams_of = [&](size_t k) { return std::min(blk, S - k * blk); };
        auto copy_block = [&](size_t k) {   // on the copy stream, once the kernels that last read this buffer are done
            const int b = (int)(k & 1);
            return hip_ok(hipStreamWaitEvent(c->copy_stream, c->ingest_freed[b], 0), "hipStreamWaitEvent") &&
                   hip_ok(hipMemcpyAsync(dbuf[b], static_cast<const unsigned char *>(pcm) + k * blk * row_bytes, streams_of(k) * row_bytes,
                                         hipMemcpyHostToDevice, c->copy_stream), "hipMemcpyAsync(ingest)") &&
                   hip_ok(hipEventRecord(c->ingest_landed[b], c->copy_stream), "hipEventRecord");
        };
        // every way out after the first copy was queued waits for the copy stream: a copy from the caller's buffer may still be in flight,
        // and the caller is free to release or reuse `pcm` the moment this call returns -- error or not
        auto fail = [&]() {
            const std::string why = last_error();
            (void)hipStreamSynchronize(c->copy_stream);
            (void)hipGetLastError();
            set_last_error(why);
            return -1;
        };
        for (int b = 0; b < 2; ++b)
            if (!hip_ok(hipEventRecord(c->ingest_freed[b], c->stream), "hipEventRecord")) return -1;
        if (!copy_block(0)) return fail();
        for (size_t k = 0; k < n_blocks; ++k) {
            const int b = (int)(k & 1);
            if (k + 1 < n_blocks && !copy_block(k + 1)) return fail();   // the next block's copy goes out before this block's kernels
            if (!hip_ok(hipStreamWaitEvent(c->stream, c->ingest_landed[b], 0), "hipStreamWaitEvent")) return fail();
            GatherTo g;
            g.on = true; g.host = true; g.stream_base = (int)(k * blk); g.device_pcm = true;
            // the block through the ordinary batched path; its detections land in det / n_det at the block's rows (the call waits for them)
            if (batch_detect_impl(ctx, dbuf[b], fmt, streams_of(k), n_samples, pcm_stride, t, config, det, n_det, max_det, nullptr, nullptr, g) != 0) return fail();
            if (!hip_ok(hipEventRecord(c->ingest_freed[b], c->stream), "hipEventRecord")) return fail();
        }
        if (!hip_ok(hipStreamSynchronize(c->copy_stream), "hipStreamSynchronize")) return -1;
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    });
}

int rp_batch_detect_sharded(rp_ctx *const *ctxs, const rp_templates *const *t, int n_shards, const void *const *pcm,
                            rp_sample_format fmt, const size_t *S, size_t n_samples, size_t pcm_stride,
                            const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det, int max_det) {
    return guarded([&]() -> int {
        if (!ctxs || !t || !pcm || !S || !config || !det || !n_det) { set_last_error("null argument"); return -1; }
        if (n_shards < 1 || n_shards > 64) { set_last_error("rp_batch_detect_sharded: 1..64 shards"); return -1; }
        size_t total = 0;
        std::vector<size_t> first((size_t)n_shards);
        for (int g = 0; g < n_shards; ++g) {
            if (!ctxs[g] || !t[g]) { set_last_error("null handle"); return -1; }
            if (t[g]->impl->ctx != ctxs[g]->impl.get()) { set_last_error("rp_batch_detect_sharded: templates[g] must have been created on ctxs[g]"); return -1; }
            if ((ctxs[g]->impl->flags & RP_CTX_HOST_POINTERS) != (ctxs[0]->impl->flags & RP_CTX_HOST_POINTERS)) {
                set_last_error("rp_batch_detect_sharded: all contexts must agree on RP_CTX_HOST_POINTERS"); return -1;
            }
            for (int h = 0; h < g; ++h) if (ctxs[h] == ctxs[g]) { set_last_error("rp_batch_detect_sharded: a context may serve one shard only"); return -1; }
            if (S[g] && !pcm[g]) { set_last_error("null argument"); return -1; }
            first[g] = total; total += S[g];
        }
        if (total > 0x7fffffffULL) { set_last_error("rp_batch_detect_sharded: too many streams"); return -1; }
        GatherTo to;
        to.on = true; to.host = (ctxs[0]->impl->flags & RP_CTX_HOST_POINTERS) != 0; to.device = ctxs[0]->impl->device;
        // one host thread per shard drives that shard's device and stream; the shards share nothing but read-only inputs
        std::vector<int> status((size_t)n_shards, 0);
        std::vector<std::string> errs((size_t)n_shards);
        // device-resident gather: let every other device write into ctxs[0]'s device directly (xGMI) where the node allows it;
        // without peer access hipMemcpyPeerAsync still works (staged by the runtime), so a refusal here is not an error
        std::string info = to.host ? "gather into host memory (RP_CTX_HOST_POINTERS): device-to-host copies" : "gather onto device " + std::to_string(to.device) + ":";
        if (!to.host)
            for (int g = 1; g < n_shards; ++g) {
                const int dev = ctxs[g]->impl->device;
                int can = 0;
                const char *how = dev == to.device ? "same device" : "staged by the runtime (no peer access)";
                if (dev != to.device && hipDeviceCanAccessPeer(&can, dev, to.device) == hipSuccess && can && hipSetDevice(dev) == hipSuccess) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(to.device, 0);
                    if (pe == hipSuccess) how = "peer access enabled (direct write over xGMI)";
                    else if (pe == hipErrorPeerAccessAlreadyEnabled) how = "peer access already enabled (direct write over xGMI)";
                    else how = "staged by the runtime (hipDeviceEnablePeerAccess refused)";
                    if (pe != hipSuccess) (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled or a refusal: both fine
                }
                info += " shard " + std::to_string(g) + " (device " + std::to_string(dev) + "): " + how + ";";
            }
        g_sharded_info = info;
        auto run = [&](int g) noexcept {
            try {
                GatherTo mine = to;
                mine.stream_base = (int)first[g];
                status[g] = S[g] ? batch_detect_impl(ctxs[g], pcm[g], fmt, S[g], n_samples, pcm_stride, t[g], config, det, n_det, max_det, nullptr,
                                                     nullptr, mine) : 0;
                if (status[g] != 0) errs[g] = last_error();  // the error text is thread-local: hand it to the caller's thread
            } catch (...) {  // e.g. bad_alloc while copying the error text: never let an exception leave a thread
                status[g] = -1;
            }
        };
        // shard 0 runs on the caller's thread; if the process cannot start another thread the remaining shards run here too
        std::vector<std::thread> th;
        th.reserve((size_t)n_shards);
        int on_threads = 1;  // shards [1, on_threads) have a thread of their own
        for (int g = 1; g < n_shards; ++g) {
            try { th.emplace_back(run, g); } catch (const std::system_error &) { break; }
            on_threads = g + 1;
        }
        run(0);
        for (int g = on_threads; g < n_shards; ++g) run(g);
        for (auto &x : th) x.join();
        for (int g = 0; g < n_shards; ++g)
            if (status[g] != 0) { set_last_error("shard " + std::to_string(g) + ": " + (errs[g].empty() ? std::string("failed") : errs[g])); return -1; }
        return 0;
    });
}

int rp_batch_detect_multi(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                          size_t n_wakewords, const rp_templates *const *t, const rp_detector_config *config,
                          const float *thresholds, const float *avg_thresholds, rp_batch_detection *det,
                          int32_t *det_wakeword, int32_t *n_det, int max_det) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        if (!config || !t || (S && (!pcm || !det || !n_det))) { set_last_error("null argument"); return -1; }
        for (size_t j = 0; j < n_wakewords && j < (size_t)kScanMaxWakewords; ++j) if (!t[j]) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (n_wakewords < 1 || n_wakewords > (size_t)kScanMaxWakewords) { set_last_error("rp_batch_detect_multi: 1..8 wakewords"); return -1; }
        if (pcm_stride < n_samples) { set_last_error("pcm_stride smaller than n_samples"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        const int K = t[0]->impl->dev.K;
        int max_len = 0;
        for (size_t j = 0; j < n_wakewords; ++j) {
            const TemplatesDev &td = t[j]->impl->dev;
            if (td.K != K) { set_last_error("Usage of wakewords with different mfcc size is not supported, ignoring wakeword"); return -1; }
            max_len = std::max(max_len, td.max_len);  // on_wakeword_change, src/detector.rs:328-334: max over the wakewords' frame sizes
        }
        const MfccTablesDev *tb = c->tables_for(K);
        if (!tb) return -1;
        const size_t nf = rp_mfcc_num_frames(n_samples);
        const size_t n_win = nf >= (size_t)max_len ? nf - max_len + 1 : 0, rows = S * n_win;
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        BatchDetection *dd = static_cast<BatchDetection *>(sg.out(det, S * (size_t)max_det * sizeof(BatchDetection), c->stage_out));
        int32_t *dn = static_cast<int32_t *>(sg.out(n_det, S * sizeof(int32_t), c->stage_out2));
        int32_t *dw = det_wakeword ? static_cast<int32_t *>(sg.out(det_wakeword, S * (size_t)max_det * sizeof(int32_t), c->stage_out3)) : nullptr;
        if (S && (!dp || !dd || !dn)) return -1;
        if (!c->ws_mfcc.reserve(S * nf * K * sizeof(float) + 64 * K * sizeof(float))) return -1;
        float *dm = c->ws_mfcc.as<float>();
        c->time_begin(kKernelMfcc);
        bool ok = hip_ok(launch_mfcc_fmt(c->stream, *tb, dp, (int)fmt, S, n_samples, pcm_stride, 0, nf, nf, dm), "mfcc_kernel");
        c->time_end();
        if (!ok) return -1;
        ScanWakewords ww{};
        ww.n = (int)n_wakewords;
        size_t maxT = 1;
        for (size_t j = 0; j < n_wakewords; ++j) maxT = std::max<size_t>(maxT, (size_t)t[j]->impl->dev.T);
        // one shared per-template score buffer, per wakeword its aggregate / avg rows
        if (!c->ws_scores.reserve(rows * maxT * sizeof(float) + 16) || !c->ws_agg.reserve(n_wakewords * rows * sizeof(float) + 16) ||
            !c->ws_avg.reserve(n_wakewords * rows * sizeof(float) + 16)) return -1;
        float *ds = c->ws_scores.as<float>();
        for (size_t j = 0; j < n_wakewords; ++j) {
            const TemplatesDev &td = t[j]->impl->dev;
            const float thr = thresholds && !std::isnan(thresholds[j]) ? thresholds[j] : config->threshold;
            const float athr = avg_thresholds && !std::isnan(avg_thresholds[j]) ? avg_thresholds[j] : config->avg_threshold;
            const bool do_avg = td.has_avg && athr != 0.f;  // wakeword_comp.rs:85
            float *dg = c->ws_agg.as<float>() + j * rows, *da = do_avg ? c->ws_avg.as<float>() + j * rows : nullptr;
            if (n_win) {
                // windows below the wakeword's avg_threshold are not compared with its sample templates (wakeword_comp.rs:85-93)
                const bool detect_only = !(c->flags & RP_CTX_FULL_SCORES);  // this entry point has no per-window outputs
                const bool gated = do_avg && detect_only && dtw_gate_supported(td, config->band_size, rows);
                const float abandon = (detect_only && config->score_mode == RP_SCORE_MAX) ? dtw_abandon_nc(thr, config->score_ref) : __builtin_inff();
                if (gated && !c->ws_list.reserve((rows + 1) * sizeof(uint32_t) + 16)) return -1;
                c->time_begin(kKernelDtw);
                if (gated) {
                    uint32_t *lst = c->ws_list.as<uint32_t>();
                    ok = hip_ok(launch_dtw_gated(c->stream, c->dtw_work(), td, dm, S, nf, 0, n_win, config->band_size, config->score_ref, athr, ds, da, lst + 1, lst,
                                                 true, abandon), "dtw kernels (gated)");
                } else {
                    ok = hip_ok(launch_dtw(c->stream, c->dtw_work(), td, dm, S, nf, 0, n_win, n_win, config->band_size, config->score_ref, do_avg ? 1 : 0, ds, da, true, abandon), "dtw kernel");
                }
                c->time_end();
                if (!ok) return -1;
                AggExtra ax;   // windows the gate rejected were never scored: their aggregate is 0, not what `scores` held
                if (gated) { ax.gate_avg = da; ax.gate_threshold = athr; }
                c->time_begin(kKernelAggregate);
                ok = hip_ok(launch_aggregate(c->stream, ds, rows, td.T, (int)config->score_mode, dg, ax), "aggregate_kernel");
                c->time_end();
                if (!ok) return -1;
            }
            ww.agg[j] = dg; ww.avg[j] = da; ww.threshold[j] = thr; ww.avg_threshold[j] = athr;
        }
        ScanConfig sc;
        sc.threshold = config->threshold; sc.avg_threshold = config->avg_threshold; sc.min_scores = (int)config->min_scores;
        sc.eager = config->eager ? 1 : 0; sc.max_len = max_len; sc.avg_enabled = 0;
        float *dv = nullptr;
        if (config->vad_mode != RP_VAD_NONE) {
            if (!c->ws_vad.reserve(S * nf * sizeof(float) + 16)) return -1;
            dv = c->ws_vad.as<float>();
            if (!hip_ok(launch_vad_value(c->stream, dm, S * nf, K, dv), "vad_value_kernel")) return -1;
        }
        c->time_begin(kKernelScan);
        ok = hip_ok(launch_scan_multi(c->stream, ww, dv, vad_mode_value(config->vad_mode), S, nf, sc, dd, dw, dn, max_det), "scan_kernel");
        c->time_end();
        if (!ok) return -1;
        if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t))) return -1;
        if (dw && !sg.back(det_wakeword, dw, S * (size_t)max_det * sizeof(int32_t))) return -1;
        return sg.finish() ? 0 : -1;
    });
}

int rp_resampler_frame_lengths(size_t sample_rate, size_t *in_len, size_t *out_len) {
    return guarded([&]() -> int {
        if (!in_len || !out_len) { set_last_error("null argument"); return -1; }
        if (!resampler_frame_lengths(sample_rate, in_len, out_len)) {
            set_last_error("Unsupported sample rate, unable to initialize the resampler");
            return -1;
        }
        return 0;
    });
}

int rp_resample_batch(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, int channels, size_t sample_rate, size_t S,
                      size_t n_samples, size_t pcm_stride, float *out, size_t out_stride) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        if (channels < 1) { set_last_error("Unsupported channel count"); return -1; }
        if (pcm_stride < n_samples * (size_t)channels) { set_last_error("pcm_stride smaller than n_samples * channels"); return -1; }
        const Resampler *rs = c->resampler_for(sample_rate);
        if (!rs) return -1;
        const size_t n_chunks = n_samples / (size_t)rs->dev.fi, n_out = n_chunks * (size_t)rs->dev.fo;
        if (out_stride < n_out) { set_last_error("out_stride smaller than the resampled length"); return -1; }
        if (S == 0 || n_chunks == 0) return 0;
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        float *dout = static_cast<float *>(sg.out(out, S * out_stride * sizeof(float), c->stage_out));
        if (!dp || !dout) return -1;
        bool ok;
        if (resample_reads_in_place(rs->dev, dp, (int)fmt, pcm_stride, dout, out_stride)) {
            c->time_begin(kKernelResample);
            ok = hip_ok(launch_resample_in_place(c->stream, rs->dev, dp, (int)fmt, channels, pcm_stride, nullptr, nullptr, S, n_chunks, dout, out_stride), "resample48_fft_kernel");
            c->time_end();
        } else {
            if (!c->ws_resample.reserve(S * (1 + n_chunks) * (size_t)rs->dev.fi * sizeof(float) + 64)) return -1;
            float *xs = c->ws_resample.as<float>();
            if (!hip_ok(launch_resample_stage(c->stream, dp, (int)fmt, channels, S, n_chunks, rs->dev.fi, pcm_stride, nullptr, xs), "resample_stage_kernel")) return -1;
            c->time_begin(kKernelResample);
            ok = hip_ok(launch_resample(c->stream, rs->dev, xs, S, n_chunks, dout, out_stride), "resample kernel");
            c->time_end();
        }
        if (!ok) return -1;
        if (!sg.back(out, dout, S * out_stride * sizeof(float)) || !sg.finish()) return -1;
        return 0;
    });
}

// buffers of a fresh batch, sized for the current input frame length (30 ms frames: 3 MFCC frames each, 40 ms: 4)
static bool stream_batch_alloc(rp_stream_batch *b) {
    Ctx *c = b->c;
    struct { int K, T; } td{b->K, b->Tmax};
    const size_t S = b->S, fpf = b->fpf();
    b->cap = b->hist_frames + fpf * b->max_chunks * 8;  // compaction every 8 full-size calls
    const size_t pitch = b->cap, rows = S * fpf * b->max_chunks;
    const size_t slack = 64 * (size_t)td.K * sizeof(float);  // the DTW band reads up to band_size frames past a row
    const size_t pcm_bytes = S * (480 + b->max_chunks * b->out_len) * sizeof(float);
    for (auto &w : b->ww)
        if (!w->agg.reserve(rows * sizeof(float) + 16) || !w->avg.reserve(rows * sizeof(float) + 16) || !w->label.reserve(rows * sizeof(int32_t) + 16))
            return false;
    if (!b->pcm[0].reserve(pcm_bytes) || !b->pcm[1].reserve(pcm_bytes) || !b->mfcc[0].reserve(S * pitch * td.K * sizeof(float) + slack) ||
        !b->mfcc[1].reserve(S * pitch * td.K * sizeof(float) + slack) || !b->state.reserve(S * stream_state_bytes()) ||
        !b->scores.reserve(rows * td.T * sizeof(float) + 16) || !b->agg.reserve(rows * sizeof(float) + 16) ||
        !b->avg.reserve(rows * sizeof(float) + 16) || !b->vad.reserve(rows * sizeof(float) + 16) ||
        !b->list.reserve((rows + 1) * sizeof(uint32_t) + 16))
        return false;
    if (!hip_ok(hipMemsetAsync(b->pcm[0].p, 0, b->pcm[0].cap, c->stream), "hipMemsetAsync") ||
        !hip_ok(hipMemsetAsync(b->pcm[1].p, 0, b->pcm[1].cap, c->stream), "hipMemsetAsync") ||
        !hip_ok(hipMemsetAsync(b->mfcc[0].p, 0, b->mfcc[0].cap, c->stream), "hipMemsetAsync") ||
        !hip_ok(hipMemsetAsync(b->mfcc[1].p, 0, b->mfcc[1].cap, c->stream), "hipMemsetAsync") ||
        !hip_ok(launch_stream_state_init(c->stream, b->state.p, S), "stream_state_init_kernel"))
        return false;
    b->cur = 0; b->fill = b->hist_frames;  // an all-zero history nobody scores against (frames < 0)
    b->pcur = 0; b->last_off = 0;
    return true;
}

int rp_stream_batch_new(rp_ctx *ctx, const rp_templates *t, const rp_detector_config *config, size_t S,
                        size_t max_chunks_per_call, rp_stream_batch **out) {
    return guarded([&]() -> int {
        if (!ctx || !t) { set_last_error("null handle"); return -1; }
        if (!config || !out) { set_last_error("null argument"); return -1; }
        *out = nullptr;
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (S == 0 || max_chunks_per_call == 0) { set_last_error("rp_stream_batch_new: S and max_chunks_per_call must be >= 1"); return -1; }
        const TemplatesDev &td = t->impl->dev;
        if (!c->tables_for(td.K)) return -1;
        std::unique_ptr<rp_stream_batch> b(new rp_stream_batch());
        b->c = c; b->t = t->impl.get(); b->cfg = *config; b->S = S; b->max_chunks = max_chunks_per_call;
        b->K = td.K; b->max_len = td.max_len; b->Tmax = td.T;
        b->hist_frames = (size_t)td.max_len - 1;
        if (!stream_batch_alloc(b.get())) return -1;
        *out = b.release();
        return 0;
    });
}
void rp_stream_batch_free(rp_stream_batch *b) { delete b; }
size_t rp_stream_batch_chunks_seen(const rp_stream_batch *b) { return b ? b->chunks_seen : 0; }

int rp_stream_batch_set_input(rp_stream_batch *b, size_t sample_rate, int channels) {
    return guarded([&]() -> int {
        if (!b) { set_last_error("null handle"); return -1; }
        Ctx *c = b->c;
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (b->chunks_seen) { set_last_error("rp_stream_batch_set_input: the streams have already received audio"); return -1; }
        if (channels < 1) { set_last_error("Unsupported channel count"); return -1; }
        size_t fi = 480, fo = 480;
        if (!resampler_frame_lengths(sample_rate, &fi, &fo)) { set_last_error("Unsupported sample rate, unable to initialize the resampler"); return -1; }
        b->channels = channels; b->in_len = fi; b->rs = nullptr;
        if (fo != b->out_len) {  // 11.025 / 22.05 kHz: 40 ms frames of four 10 ms shifts
            b->out_len = fo;
            if (!stream_batch_alloc(b)) return -1;
        }
        if (sample_rate != 16000) {
            b->rs = c->resampler_for(sample_rate);
            if (!b->rs) return -1;
            if (!b->rs_prev[0].reserve(b->S * fi * sizeof(float)) || !b->rs_prev[1].reserve(b->S * fi * sizeof(float)) ||
                !b->rs_out.reserve(b->S * b->max_chunks * fo * sizeof(float))) return -1;
            if (!b->rs->dev.fft48 && !b->rs_xs.reserve(b->S * (1 + b->max_chunks) * fi * sizeof(float) + 64)) return -1;
            if (!hip_ok(hipMemsetAsync(b->rs_prev[0].p, 0, b->S * fi * sizeof(float), c->stream), "hipMemsetAsync")) return -1;
            b->rs_cur = 0;
        }
        return 0;
    });
}
size_t rp_stream_batch_samples_per_chunk(const rp_stream_batch *b) { return b ? b->in_len * (size_t)b->channels : 0; }

int rp_stream_batch_reset(rp_stream_batch *b, long long stream) {
    return guarded([&]() -> int {
        if (!b) { set_last_error("null handle"); return -1; }
        Ctx *c = b->c;
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (b->poisoned) { set_last_error("stream batch is in a failed state (an earlier call failed half way); free it and create a new one"); return -1; }
        if (stream >= (long long)b->S) { set_last_error("rp_stream_batch_reset: no such stream"); return -1; }
        // the next chunk only refills the extractor: its three frames (3C-3 .. 3C-1) are never emitted
        return hip_ok(launch_stream_state_reset(c->stream, b->state.p, b->S, stream, (long long)b->fpf() * (long long)b->chunks_seen), "stream_state_reset_kernel") ? 0 : -1;
    });
}

static int stream_batch_process_impl(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                                     rp_batch_detection *det, int32_t *n_det, int max_det, float *agg, int32_t *det_wakeword,
                                     int32_t *det_label, bool *state_touched);
static int stream_batch_score_multi(rp_stream_batch *b, Staged &sg, const float *now, size_t fill, size_t n_new, BatchDetection *dd,
                                    int32_t *dn, int max_det, int32_t *det_wakeword, int32_t *det_label);

int rp_stream_batch_process(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                            rp_batch_detection *det, int32_t *n_det, int max_det, float *agg) {
    if (!b) { set_last_error("null handle"); return -1; }
    if (b->poisoned) { set_last_error("stream batch is in a failed state (an earlier call failed half way); free it and create a new one"); return -1; }
    // The call advances device-resident state launch by launch (resampler tail, history chunk, MFCC rows, scan state);
    // a failure after the first such step cannot be rolled back, so the batch refuses further work instead of pairing
    // the wrong history with later chunks.
    bool touched = false;
    const int r = stream_batch_process_impl(b, pcm, fmt, n_chunks, pcm_stride, det, n_det, max_det, agg, nullptr, nullptr, &touched);
    if (r != 0 && touched) b->poisoned = true;
    return r;
}

int rp_stream_batch_process_multi(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                                  rp_batch_detection *det, int32_t *det_wakeword, int32_t *det_label, int32_t *n_det, int max_det) {
    if (!b) { set_last_error("null handle"); return -1; }
    if (b->poisoned) { set_last_error("stream batch is in a failed state (an earlier call failed half way); free it and create a new one"); return -1; }
    bool touched = false;
    const int r = stream_batch_process_impl(b, pcm, fmt, n_chunks, pcm_stride, det, n_det, max_det, nullptr, det_wakeword, det_label, &touched);
    if (r != 0 && touched) b->poisoned = true;
    return r;
}

static int stream_batch_process_impl(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                                     rp_batch_detection *det, int32_t *n_det, int max_det, float *agg, int32_t *det_wakeword,
                                     int32_t *det_label, bool *state_touched) {
    return guarded([&]() -> int {
        Ctx *c = b->c;
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (n_chunks == 0 || n_chunks > b->max_chunks) { set_last_error("rp_stream_batch_process: n_chunks out of range"); return -1; }
        const size_t in_chunk = b->in_len * (size_t)b->channels;
        if (pcm_stride < n_chunks * in_chunk) { set_last_error("pcm_stride smaller than n_chunks * samples per chunk"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        const bool multi = !b->ww.empty();
        if (multi && agg) { set_last_error("rp_stream_batch_process: a batch of several wakewords has no single aggregate per window"); return -1; }
        static const TemplatesDev no_templates{};
        const TemplatesDev &td_one = multi ? no_templates : b->t->dev;
        struct { int K, T, max_len, has_avg; } td{b->K, td_one.T, b->max_len, td_one.has_avg};
        const MfccTablesDev *tb = c->tables_for(td.K);
        if (!tb) return -1;
        const size_t fo = b->out_len, new_len = n_chunks * fo;  // encoded samples this call adds to every stream
        const size_t S = b->S, n_new = b->fpf() * n_chunks, hist = b->hist_frames, pitch = b->cap, rows = S * n_new;
        // a row is [the last 480 encoded samples of the previous call | the new ones]
        const size_t n_samples = 480 + new_len, pcm_pitch = 480 + b->max_chunks * fo;
        const bool do_avg = td.has_avg && b->cfg.avg_threshold != 0.f;
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        BatchDetection *dd = static_cast<BatchDetection *>(sg.out(det, S * (size_t)max_det * sizeof(BatchDetection), c->stage_out));
        int32_t *dn = static_cast<int32_t *>(sg.out(n_det, S * sizeof(int32_t), c->stage_out2));
        if (!dp || !dd || !dn) { if (!pcm || !det || !n_det) set_last_error("null argument"); return -1; }
        const float *hp_old = b->pcm[b->pcur].as<float>();
        float *hp = b->pcm[b->pcur ^ 1].as<float>();
        *state_touched = true;  // from here on every launch moves persistent state
        // 16 kHz mono input is read where it lies: the MFCC kernel takes [history chunk | new chunks] from two buffers and
        // leaves the last chunk as the next call's history.  Other inputs are staged into one row per stream first.
        bool staged = false;
        if (b->rs) {  // previous input frame | new input frames -> 16 kHz (the resampler never resets, src/detector.rs:290-302)
            const size_t fi = b->in_len;
            float *ro = b->rs_out.as<float>();
            float *pv = b->rs_prev[b->rs_cur].as<float>(), *pn = b->rs_prev[b->rs_cur ^ 1].as<float>();
            if (resample_reads_in_place(b->rs->dev, dp, (int)fmt, pcm_stride, ro, new_len)) {
                c->time_begin(kKernelResample);
                bool okr = hip_ok(launch_resample_in_place(c->stream, b->rs->dev, dp, (int)fmt, b->channels, pcm_stride, pv, pn, S, n_chunks, ro, new_len), "resample48_fft_kernel");
                c->time_end();
                if (!okr) return -1;
            } else {
                if (!b->rs_xs.reserve(S * (1 + b->max_chunks) * fi * sizeof(float) + 64)) return -1;
                float *xs = b->rs_xs.as<float>();
                if (!hip_ok(launch_resample_stage(c->stream, dp, (int)fmt, b->channels, S, n_chunks, (int)fi, pcm_stride, pv, xs), "resample_stage_kernel")) return -1;
                c->time_begin(kKernelResample);
                bool okr = hip_ok(launch_resample(c->stream, b->rs->dev, xs, S, n_chunks, ro, new_len), "resample kernel");
                c->time_end();
                if (!okr) return -1;
                if (!hip_ok(launch_carry_rows(c->stream, xs, S, (1 + n_chunks) * fi, n_chunks * fi, fi, pn, fi), "carry_rows_kernel")) return -1;
            }
            b->rs_cur ^= 1;
            if (!hip_ok(launch_stream_stage(c->stream, ro, 3, 1, S, new_len, new_len, hp_old, b->last_off, hp, pcm_pitch), "stream_stage_kernel")) return -1;
            staged = true;
        } else if (b->channels != 1) {  // previous chunk | new chunks (first channel), decoded to f32
            if (!hip_ok(launch_stream_stage(c->stream, dp, (int)fmt, b->channels, S, new_len, pcm_stride, hp_old, b->last_off, hp, pcm_pitch), "stream_stage_kernel")) return -1;
            staged = true;
        }
        // MFCC window rows: [.. valid frames .. | the 3*n_chunks new frames]; a full row keeps its last max_len-1 frames
        if (b->fill + n_new > b->cap) {
            if (!hip_ok(launch_carry_rows(c->stream, b->mfcc[b->cur].as<float>(), S, pitch * td.K, (b->fill - hist) * td.K, hist * td.K,
                                          b->mfcc[b->cur ^ 1].as<float>(), pitch * td.K), "carry_rows_kernel")) return -1;
            b->cur ^= 1; b->fill = hist;
        }
        float *now = b->mfcc[b->cur].as<float>();
        const size_t fill = b->fill;
        bool ok = true;
        if (!staged) {
            c->time_begin(kKernelMfcc);
            hipError_t e = launch_mfcc_stream(c->stream, *tb, dp, (int)fmt, S, n_chunks, pcm_stride, hp_old + b->last_off, pcm_pitch, hp, pitch,
                                              now + fill * td.K);
            c->time_end();
            if (e == hipErrorNotSupported) {  // rows that do not allow 4-sample loads
                if (!hip_ok(launch_stream_stage(c->stream, dp, (int)fmt, 1, S, new_len, pcm_stride, hp_old, b->last_off, hp, pcm_pitch), "stream_stage_kernel")) return -1;
                staged = true;
            } else {
                if (!hip_ok(e, "mfcc_kernel")) return -1;
                b->pcur ^= 1; b->last_off = 0;  // hist_out: the last chunk of this call at the start of the other buffer's rows
            }
        }
        if (staged) {
            b->pcur ^= 1; b->last_off = new_len;  // the last 480 samples of this call are the extractor history of the next
            c->time_begin(kKernelMfcc);
            ok = hip_ok(launch_mfcc(c->stream, *tb, hp, S, n_samples, pcm_pitch, 0, n_new, pitch, now + fill * td.K), "mfcc_kernel");
            c->time_end();
            if (!ok) return -1;
        }
        b->fill += n_new;
        if (multi) {
            if (stream_batch_score_multi(b, sg, now, fill, n_new, dd, dn, max_det, det_wakeword, det_label) != 0) return -1;
            b->chunks_seen += n_chunks;
            if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t))) return -1;
            return sg.finish() ? 0 : -1;
        }
        float *ds = b->scores.as<float>(), *dg = b->agg.as<float>(), *da = do_avg ? b->avg.as<float>() : nullptr;
        // the averaged-template gate as a skip (wakeword_comp.rs:85-93), unless the caller wants every window's aggregate
        const bool detect_only = !agg && !(c->flags & RP_CTX_FULL_SCORES);
        // (a single live stream with a handful of windows skips the gate's three passes: launch_dtw scores it -- with the batch kernels when
        // the matrix-core kernel serves its templates (a stream's bits must not depend on the batch it is in), else one wave per DTW)
        const bool gated = do_avg && detect_only && dtw_gate_supported(td_one, b->cfg.band_size, rows) && !(S == 1 && n_new <= 8);
        const float abandon = (detect_only && b->cfg.score_mode == RP_SCORE_MAX) ? dtw_abandon_nc(b->cfg.threshold, b->cfg.score_ref) : __builtin_inff();
        // ScoreMode::Max inside the matrix-core DTW kernel when one chunk holds the reference's templates (DtwFusedAgg, rp_kernels.h)
        DtwFusedAgg fz;
        if (b->cfg.score_mode == RP_SCORE_MAX && !do_avg && rows) { fz.agg = dg; fz.threshold = b->cfg.threshold; }
        c->time_begin(kKernelDtw);
        if (gated) {
            uint32_t *lst = b->list.as<uint32_t>();
            ok = hip_ok(launch_dtw_gated(c->stream, c->dtw_work(), td_one, now, S, pitch, fill - hist, n_new, b->cfg.band_size, b->cfg.score_ref, b->cfg.avg_threshold,
                                         ds, da, lst + 1, lst, true, abandon), "dtw kernels (gated)");
        } else {
            ok = hip_ok(launch_dtw(c->stream, c->dtw_work(), td_one, now, S, pitch, fill - hist, n_new, n_new, b->cfg.band_size, b->cfg.score_ref, do_avg ? 1 : 0, ds, da, true, abandon,
                                   fz.agg ? &fz : nullptr), "dtw kernel");
        }
        c->time_end();
        if (!ok) return -1;
        AggExtra ax;   // windows the gate rejected were never scored: their aggregate is 0, not what `scores` held
        if (gated) { ax.gate_avg = da; ax.gate_threshold = b->cfg.avg_threshold; }
        if (!fz.done) {
            c->time_begin(kKernelAggregate);
            ok = hip_ok(launch_aggregate(c->stream, ds, rows, td.T, (int)b->cfg.score_mode, dg, ax), "aggregate_kernel");
            c->time_end();
        }
        if (!ok) return -1;
        float *dv = nullptr;
        if (b->cfg.vad_mode != RP_VAD_NONE) {
            dv = b->vad.as<float>();
        }
        ScanConfig sc;
        sc.threshold = b->cfg.threshold; sc.avg_threshold = b->cfg.avg_threshold; sc.min_scores = (int)b->cfg.min_scores;
        sc.eager = b->cfg.eager ? 1 : 0; sc.max_len = td.max_len; sc.avg_enabled = do_avg ? 1 : 0; sc.fpf = (int)b->fpf();
        if (dv && !hip_ok(launch_vad_value_rows(c->stream, now + fill * td.K, S, n_new, pitch, td.K, dv), "vad_value_kernel")) return -1;
        c->time_begin(kKernelScan);
        ok = hip_ok(launch_scan_stream(c->stream, dg, da, dv, vad_mode_value(b->cfg.vad_mode), S, (long long)b->fpf() * (long long)b->chunks_seen - 3, (int)n_new,
                                       sc, b->state.p, dd, dn, max_det), "scan_stream_kernel");
        c->time_end();
        if (!ok) return -1;
        b->chunks_seen += n_chunks;
        if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t))) return -1;
        if (det_wakeword || det_label) {   // one wakeword reference: wakeword 0, no label
            const size_t nb = S * (size_t)max_det * sizeof(int32_t);
            if (sg.host) { if (det_wakeword) std::memset(det_wakeword, 0, nb); if (det_label) std::memset(det_label, 0xff, nb); }
            else if ((det_wakeword && !hip_ok(hipMemsetAsync(det_wakeword, 0, nb, c->stream), "hipMemsetAsync")) ||
                     (det_label && !hip_ok(hipMemsetAsync(det_label, 0xff, nb, c->stream), "hipMemsetAsync"))) return -1;
        }
        if (agg) {
            if (sg.host) { if (!sg.back(agg, dg, rows * sizeof(float))) return -1; }
            else if (!hip_ok(hipMemcpyAsync(agg, dg, rows * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync(D2D)")) return -1;
        }
        return sg.finish() ? 0 : -1;
    });
}

static hipError_t mlp_rows_mfma(Ctx *c, const Model &m, const float *dx, size_t B, int precision, float *out);

// ---- live-stream batches that hold several wakewords and / or a wakeword model (src/detector.rs:304-346,433-447)
int rp_stream_batch_new_multi(rp_ctx *ctx, size_t n_wakewords, const rp_wakeword_spec *wakewords, int mfcc_size,
                              const rp_detector_config *config, size_t S, size_t max_chunks_per_call, rp_stream_batch **out) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        if (!config || !out || !wakewords) { set_last_error("null argument"); return -1; }
        *out = nullptr;
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (S == 0 || max_chunks_per_call == 0) { set_last_error("rp_stream_batch_new: S and max_chunks_per_call must be >= 1"); return -1; }
        if (n_wakewords < 1 || n_wakewords > (size_t)kScanMaxWakewords) { set_last_error("rp_stream_batch_new_multi: 1..8 wakewords"); return -1; }
        if (mfcc_size < 1) { set_last_error("rp_stream_batch_new_multi: mfcc_size must be >= 1"); return -1; }
        std::unique_ptr<rp_stream_batch> b(new rp_stream_batch());
        b->c = c; b->t = nullptr; b->cfg = *config; b->S = S; b->max_chunks = max_chunks_per_call;
        b->K = mfcc_size; b->max_len = 0; b->Tmax = 1;
        for (size_t j = 0; j < n_wakewords; ++j) {
            const rp_wakeword_spec &w = wakewords[j];
            if ((w.templates != nullptr) == (w.model != nullptr)) { set_last_error("rp_stream_batch_new_multi: every wakeword is a reference OR a model"); return -1; }
            std::unique_ptr<StreamWakeword> e(new StreamWakeword());
            e->threshold = std::isnan(w.threshold) ? config->threshold : w.threshold;
            e->avg_threshold = std::isnan(w.avg_threshold) ? config->avg_threshold : w.avg_threshold;
            if (w.templates) {
                e->t = w.templates->impl.get();
                if (e->t->ctx != c) { set_last_error("rp_stream_batch_new_multi: the wakewords must have been created on this context"); return -1; }
                // add_wakeword, src/detector.rs:316-319
                if (e->t->dev.K != mfcc_size) { set_last_error("Usage of wakewords with different mfcc size is not supported, ignoring wakeword"); return -1; }
                b->max_len = std::max(b->max_len, e->t->dev.max_len);
                b->Tmax = std::max(b->Tmax, e->t->dev.T);
            } else {
                e->m = w.model->impl.get();
                if (e->m->ctx != c) { set_last_error("rp_stream_batch_new_multi: the wakewords must have been created on this context"); return -1; }
                const int nl = (int)e->m->dims.size() - 1;
                if (e->m->dims[0] % mfcc_size != 0) { set_last_error("Usage of wakewords with different mfcc size is not supported, ignoring wakeword"); return -1; }
                if (w.none_index >= e->m->dims[nl]) { set_last_error("none_index out of range"); return -1; }
                if (w.precision != RP_MLP_F32 && w.precision != RP_MLP_BF16 && w.precision != RP_MLP_F32_STRICT && w.precision != RP_MLP_F32_FAST) { set_last_error("unknown MLP precision"); return -1; }
                if (!e->m->mfma_ok && w.precision == RP_MLP_BF16) { set_last_error("this layer-1 shape has no bf16 MFMA kernel"); return -1; }
                e->none_index = w.none_index; e->precision = w.precision;
                b->max_len = std::max(b->max_len, e->m->dims[0] / mfcc_size);
            }
            b->ww.push_back(std::move(e));
        }
        if (!c->tables_for(b->K)) return -1;
        b->hist_frames = (size_t)b->max_len - 1;   // on_wakeword_change, src/detector.rs:328-334: the longest wakeword sets the window
        if (!stream_batch_alloc(b.get())) return -1;
        *out = b.release();
        return 0;
    });
}

// scores of this call's n_new windows per stream for every wakeword, then the state machine over all of them
static int stream_batch_score_multi(rp_stream_batch *b, Staged &sg, const float *now, size_t fill, size_t n_new, BatchDetection *dd,
                                    int32_t *dn, int max_det, int32_t *det_wakeword, int32_t *det_label) {
    Ctx *c = b->c;
    const size_t S = b->S, hist = b->hist_frames, pitch = b->cap, rows = S * n_new;
    const int K = b->K;
    const bool detect_only = !(c->flags & RP_CTX_FULL_SCORES);
    ScanWakewords sw{};
    sw.n = (int)b->ww.size();
    bool ok = true;
    for (size_t j = 0; j < b->ww.size(); ++j) {
        StreamWakeword &w = *b->ww[j];
        float *dg = w.agg.as<float>();
        if (w.t) {
            const TemplatesDev &td = w.t->dev;
            const bool do_avg = td.has_avg && w.avg_threshold != 0.f;  // wakeword_comp.rs:85
            float *da = do_avg ? w.avg.as<float>() : nullptr, *ds = b->scores.as<float>();
            // the window starts where the longest wakeword's does and this one scores its oldest frames (wakeword_comp.rs:22-27)
            const bool gated = do_avg && detect_only && dtw_gate_supported(td, b->cfg.band_size, rows);
            const float abandon = (detect_only && b->cfg.score_mode == RP_SCORE_MAX) ? dtw_abandon_nc(w.threshold, b->cfg.score_ref) : __builtin_inff();
            c->time_begin(kKernelDtw);
            if (gated) {
                uint32_t *lst = b->list.as<uint32_t>();
                ok = hip_ok(launch_dtw_gated(c->stream, c->dtw_work(), td, now, S, pitch, fill - hist, n_new, b->cfg.band_size, b->cfg.score_ref, w.avg_threshold,
                                             ds, da, lst + 1, lst, true, abandon), "dtw kernels (gated)");
            } else {
                ok = hip_ok(launch_dtw(c->stream, c->dtw_work(), td, now, S, pitch, fill - hist, n_new, n_new, b->cfg.band_size, b->cfg.score_ref, do_avg ? 1 : 0, ds, da, true, abandon), "dtw kernel");
            }
            c->time_end();
            if (!ok) return -1;
            AggExtra ax;
            if (gated) { ax.gate_avg = da; ax.gate_threshold = w.avg_threshold; }
            c->time_begin(kKernelAggregate);
            ok = hip_ok(launch_aggregate(c->stream, ds, rows, td.T, (int)b->cfg.score_mode, dg, ax), "aggregate_kernel");
            c->time_end();
            if (!ok) return -1;
            sw.agg[j] = dg; sw.avg[j] = da; sw.threshold[j] = w.threshold; sw.avg_threshold[j] = w.avg_threshold; sw.label[j] = nullptr;
        } else {
            const Model &m = *w.m;
            const int nl_layers = (int)m.dims.size() - 1, L = m.dims[0] / K, n_labels = m.dims[nl_layers];
            if (!b->logits.reserve(rows * (size_t)n_labels * sizeof(float) + 16)) return -1;
            float *dlog = b->logits.as<float>();
            const float *first = now + (fill - hist) * K;   // window i of stream s starts at frame s * pitch + i from here
            const float *wsum = (m.mfma_ok && K % 4 == 0) ? const_cast<Model &>(m).wsum_for(K) : nullptr;  // as rp_batch_detect_model
            if (wsum) {
                if (!b->mean.reserve(rows * (size_t)K * sizeof(float) + 16)) return -1;
                float *dmean = b->mean.as<float>();
                if (!hip_ok(launch_window_means(c->stream, first, S, pitch, n_new, L, K, dmean), "window_means_kernel")) return -1;
                c->time_begin(kKernelMlp);
                uint32_t *redo = c->mlp_redo(rows);
                if (!redo) return -1;
                ok = hip_ok(launch_mlp_mfma_windows(c->stream, m.dev, first, S, pitch, n_new, K, dmean, wsum, dlog, redo, pitch, w.precision == RP_MLP_F32_STRICT ? (int)kMlpStrictF32 : w.precision == RP_MLP_F32_FAST ? (int)kMlpF16x2 : (int)kMlpF32), "mlp_mfma_kernel");
                c->time_end();
                if (!ok) return -1;
            } else {
                int maxd = 0;
                for (int d : m.dims) maxd = std::max(maxd, d);
                if (!b->xrows.reserve(rows * (size_t)m.dims[0] * sizeof(float) + 64)) return -1;
                if (!m.mfma_ok && !b->xs2.reserve(2 * rows * (size_t)maxd * sizeof(float) + 16)) return -1;
                float *dx = b->xrows.as<float>();
                if (!hip_ok(launch_normalize_windows_batch(c->stream, first, pitch, n_new, 0, rows, L, K, dx), "normalize_windows_kernel")) return -1;
                c->time_begin(kKernelMlp);
                if (m.mfma_ok) ok = hip_ok(mlp_rows_mfma(c, m, dx, rows, w.precision, dlog), "mlp_mfma_kernel");
                else ok = hip_ok(launch_mlp(c->stream, dx, rows, nl_layers, m.dims.data(), m.W.data(), m.B.data(), b->xs2.as<float>(),
                                            b->xs2.as<float>() + rows * (size_t)maxd, dlog), "mlp_layer_kernel");
                c->time_end();
                if (!ok) return -1;
            }
            float *da = w.avg.as<float>();
            int32_t *dlab = w.label.as<int32_t>();
            if (!hip_ok(launch_nn_score(c->stream, dlog, rows, n_labels, w.none_index, b->cfg.score_ref * 10.f, w.avg_threshold != 0.f ? 1 : 0,
                                        w.threshold, w.avg_threshold, dg, da, dlab), "nn_score_kernel")) return -1;
            sw.agg[j] = dg; sw.avg[j] = da; sw.label[j] = dlab;
            sw.threshold[j] = -1.f; sw.avg_threshold[j] = -1.f;  // the gates were applied by nn_score_kernel (>=, not >)
        }
    }
    float *dv = nullptr;
    if (b->cfg.vad_mode != RP_VAD_NONE) {
        dv = b->vad.as<float>();
        if (!hip_ok(launch_vad_value_rows(c->stream, now + fill * K, S, n_new, pitch, K, dv), "vad_value_kernel")) return -1;
    }
    ScanConfig sc;
    sc.threshold = b->cfg.threshold; sc.avg_threshold = b->cfg.avg_threshold; sc.min_scores = (int)b->cfg.min_scores;
    sc.eager = b->cfg.eager ? 1 : 0; sc.max_len = b->max_len; sc.avg_enabled = 0; sc.fpf = (int)b->fpf();
    int32_t *dw = det_wakeword ? static_cast<int32_t *>(sg.out(det_wakeword, S * (size_t)max_det * sizeof(int32_t), b->det_ww)) : nullptr;
    int32_t *dl = det_label ? static_cast<int32_t *>(sg.out(det_label, S * (size_t)max_det * sizeof(int32_t), b->det_label)) : nullptr;
    if ((det_wakeword && !dw) || (det_label && !dl)) return -1;
    c->time_begin(kKernelScan);
    ok = hip_ok(launch_scan_stream_multi(c->stream, sw, dv, vad_mode_value(b->cfg.vad_mode), S, (long long)b->fpf() * (long long)b->chunks_seen - 3,
                                         (int)n_new, sc, b->state.p, dd, dw, dl, dn, max_det), "scan_stream_kernel");
    c->time_end();
    if (!ok) return -1;
    if ((dw && !sg.back(det_wakeword, dw, S * (size_t)max_det * sizeof(int32_t))) || (dl && !sg.back(det_label, dl, S * (size_t)max_det * sizeof(int32_t)))) return -1;
    return 0;
}

int rp_model_new(rp_ctx *ctx, int n_layers, const int *dims, const float *const *weights, const float *const *biases,
                 rp_model **out) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        *out = nullptr;
        std::unique_ptr<Model> m(Model::create(ctx->impl.get(), n_layers, dims, weights, biases));
        if (!m) return -1;
        rp_model *h = new rp_model();
        h->impl = std::move(m);
        *out = h;
        return 0;
    });
}
void rp_model_free(rp_model *m) { delete m; }

// Dense rows through layer 1 on the matrix cores: the line-streaming kernel (rp_mlp_stream.hip), where the pass is bound by the HBM
// stream (bf16 inputs: 0.132 against 0.155 ms at BASELINE config C5; f32 callers: its f16 two-way split form).  With the f32 matrix
// instructions themselves the f32 matrix rate binds and the register-fragment kernel's 24 waves per CU overlap better (0.198 against
// 0.207 ms): RP_MLP_STREAM=0 forces that kernel, RP_MLP_STREAM=2 the streaming kernel in the caller's own precision (benchmarks, tests).
// RP_MLP_STREAM (a tuning knob of benchmarks and tests, not an arithmetic switch -- the arithmetic is the call's `precision`): 0 = the
// register-fragment kernel for every shape, 2 = the streaming kernel with the f32 matrix instructions for RP_MLP_F32_STRICT too.
static int mlp_internal_precision(int precision) {
    return precision == RP_MLP_F32 ? (int)kMlpBf16x3 : precision == RP_MLP_F32_FAST ? (int)kMlpF16x2 : precision == RP_MLP_F32_STRICT ? (int)kMlpStrictF32 : (int)kMlpBf16;
}
static hipError_t mlp_rows_mfma(Ctx *c, const Model &m, const float *dx, size_t B, int precision, float *out) {
    const char *e = std::getenv("RP_MLP_STREAM");
    const int mode = e ? (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1) : 1;
    MlpStreamPlan plan;
    // f32 callers (RP_MLP_F32): the streaming kernel with three-part bf16 splits of inputs and weights (kMlpBf16x3: exact operands, f32
    // accumulate) runs at the HBM stream's rate like the bf16 form, where the f32 matrix rate bound both exact kernels (0.19 ms at C5)
    uint32_t *redo = c->mlp_redo(B);
    if (!redo) return hipErrorOutOfMemory;
    if (precision == RP_MLP_F32_STRICT && mode != 2) {   // the f32 matrix instructions for every row: the register-fragment kernel overlaps them best
        c->last_mlp_kernel = "mlp_mfma_kernel<f32 matrix instructions>";
        return launch_mlp_mfma(c->stream, m.dev, dx, B, kMlpStrictF32, out, redo);
    }
    const int iprec = mlp_internal_precision(precision);
    const int sprec = iprec == kMlpStrictF32 ? (int)kMlpF32 : iprec;   // (the stream kernel's kMlpF32 IS the f32 matrix instructions)
    if (mode != 0 && const_cast<Model &>(m).stream_plan(dx, B, sprec, &plan)) {
        c->last_mlp_kernel = sprec == kMlpBf16x3 ? "mlp_stream_kernel<bf16x3 splits>"
                             : sprec == kMlpF16x2 ? "mlp_stream_kernel<f16x2 splits> + mlp_mfma_kernel<f32> on listed rows"
                             : sprec == kMlpBf16 ? "mlp_stream_kernel<bf16>" : "mlp_stream_kernel<f32 matrix instructions>";
        return launch_mlp_stream(c->stream, m.dev, plan, dx, B, sprec, out, c->n_cu, redo);
    }
    c->last_mlp_kernel = precision == RP_MLP_BF16 ? "mlp_mfma_kernel<bf16>"
                         : (precision == RP_MLP_F32 && m.dev.w1t) ? "mlp_mfma_kernel<bf16x3 splits>"
                         : (precision == RP_MLP_F32_FAST && m.dev.w1s) ? "mlp_mfma_kernel<f16x2 splits> + mlp_mfma_kernel<f32> on listed rows" : "mlp_mfma_kernel<f32 matrix instructions>";
    return launch_mlp_mfma(c->stream, m.dev, dx, B, precision == RP_MLP_BF16 ? (int)kMlpBf16 : precision == RP_MLP_F32 ? (int)kMlpF32 : iprec, out, redo);
}

int rp_mlp_forward_batch(rp_ctx *ctx, const rp_model *model, const float *x, size_t B, int precision, float *logits) {
    return guarded([&]() -> int {
        if (!ctx || !model) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        const Model &m = *model->impl;
        const int nl = (int)m.dims.size() - 1;
        if (precision != RP_MLP_F32 && precision != RP_MLP_BF16 && precision != RP_MLP_F32_STRICT && precision != RP_MLP_F32_FAST) { set_last_error("unknown MLP precision"); return -1; }
        Staged sg(c);
        const float *dx = static_cast<const float *>(sg.in(x, B * (size_t)m.dims[0] * 4, c->stage_in));
        float *dl = static_cast<float *>(sg.out(logits, B * (size_t)m.dims[nl] * 4, c->stage_out));
        if (B && (!dx || !dl)) return -1;
        bool ok;
        if (m.mfma_ok) {
            c->time_begin(kKernelMlp);
            ok = hip_ok(mlp_rows_mfma(c, m, dx, B, precision, dl), "mlp_mfma_kernel");
            c->time_end();
        } else {
            if (precision == RP_MLP_BF16) { set_last_error("this layer-1 shape has no bf16 MFMA kernel"); return -1; }
            int maxd = 0;
            for (int d : m.dims) maxd = std::max(maxd, d);
            if (!c->stage_out2.reserve(B * (size_t)maxd * 4) || !c->stage_out3.reserve(B * (size_t)maxd * 4)) return -1;
            c->time_begin(kKernelMlp);
            ok = hip_ok(launch_mlp(c->stream, dx, B, nl, m.dims.data(), m.W.data(), m.B.data(), c->stage_out2.as<float>(),
                                   c->stage_out3.as<float>(), dl), "mlp_layer_kernel");
            c->time_end();
        }
        if (!ok) return -1;
        if (!sg.back(logits, dl, B * (size_t)m.dims[nl] * 4) || !sg.finish()) return -1;
        return 0;
    });
}

// Logits of every window of L frames of S streams' MFCC rows dm [S][nf][K] (WakewordNN::run_detection's forward, window by window,
// src/wakewords/nn/wakeword_nn.rs:101-159): dlog [S * n_win][labels].  Shared by rp_batch_detect_model and rp_mlp_forward_windows.
static bool window_logits(Ctx *c, const Model &m, const float *dm, size_t S, size_t nf, size_t n_win, int L, int K, int precision, float *dlog) {
    const size_t rows = S * n_win;
    const int nl_layers = (int)m.dims.size() - 1, n_labels = m.dims[nl_layers];
    bool ok = true;
    {
        // (RP_MLP_BF16 only permits bf16 inputs: the in-place window kernel is f32 and faster than materialising rows)
        const float *wsum = (m.mfma_ok && K % 4 == 0) ? const_cast<Model &>(m).wsum_for(K) : nullptr;
        if (wsum) {
            // windows read in place from the frame array, the window mean taken out after layer 1
            if (!c->ws_gain.reserve(rows * (size_t)K * sizeof(float) + 16)) return false;
            float *dmean = c->ws_gain.as<float>();
            if (!hip_ok(launch_window_means(c->stream, dm, S, nf, n_win, L, K, dmean), "window_means_kernel")) return false;
            c->time_begin(kKernelMlp);
            uint32_t *redo = c->mlp_redo(rows);
            if (!redo) return false;
            ok = hip_ok(launch_mlp_mfma_windows(c->stream, m.dev, dm, S, nf, n_win, K, dmean, wsum, dlog, redo, 0, precision == RP_MLP_F32_STRICT ? (int)kMlpStrictF32 : precision == RP_MLP_F32_FAST ? (int)kMlpF16x2 : (int)kMlpF32), "mlp_mfma_kernel");
            c->time_end();
            if (!ok) return false;
            c->last_mlp_kernel = precision == RP_MLP_F32_STRICT ? "mlp_mfma_kernel<f32 matrix instructions>, windows read in place"
                                 : (precision != RP_MLP_F32_FAST && mlp_windows_supported(m.dev, n_win, K, true) == 1) ? "mlp_windows_kernel<bf16x3 splits>"
                                 : (precision != RP_MLP_F32_FAST && mlp_windows_supported(m.dev, n_win, K, true) == 2) ? "mlp_windows_wide_kernel<bf16x3 splits>"
                                 : precision != RP_MLP_F32_FAST ? "mlp_mfma_kernel<bf16x3 splits>, windows read in place"
                                 : mlp_windows_supported(m.dev, n_win, K) == 1 ? "mlp_windows_kernel<f16x2 splits> + mlp_mfma_kernel<f32> on listed rows"
                                 : mlp_windows_supported(m.dev, n_win, K) == 2 ? "mlp_windows_wide_kernel<f16x2 splits> + mlp_mfma_kernel<f32> on listed rows"
                                 : "mlp_mfma_kernel<f16x2 splits>, windows read in place, + mlp_mfma_kernel<f32> on listed rows";
        } else {
            // windows are materialised slab by slab (a row is dims[0] floats): <= 4 GiB of rows at a time
            const size_t row_bytes = (size_t)m.dims[0] * sizeof(float);
            size_t slab = std::max<size_t>(1, ((size_t)4 << 30) / row_bytes);
            if (slab > rows) slab = rows;
            int maxd = 0;
            for (int d : m.dims) maxd = std::max(maxd, d);
            if (!c->ws_scores.reserve(slab * row_bytes + 64)) return false;
            if (!m.mfma_ok && !c->ws_gain.reserve(2 * slab * (size_t)maxd * sizeof(float) + 16)) return false;
            float *dx = c->ws_scores.as<float>();
            for (size_t r0 = 0; r0 < rows; r0 += slab) {
                const size_t nr = std::min(slab, rows - r0);
                if (!hip_ok(launch_normalize_windows_batch(c->stream, dm, nf, n_win, r0, nr, L, K, dx), "normalize_windows_kernel")) return false;
                c->time_begin(kKernelMlp);
                if (m.mfma_ok) ok = hip_ok(mlp_rows_mfma(c, m, dx, nr, precision, dlog + r0 * n_labels), "mlp_mfma_kernel");
                else ok = hip_ok(launch_mlp(c->stream, dx, nr, nl_layers, m.dims.data(), m.W.data(), m.B.data(), c->ws_gain.as<float>(),
                                            c->ws_gain.as<float>() + slab * (size_t)maxd, dlog + r0 * n_labels), "mlp_layer_kernel");
                c->time_end();
                if (!ok) return false;
            }
        }
    }
    return ok;
}

int rp_mlp_forward_windows(rp_ctx *ctx, const rp_model *model, const float *mfcc, size_t S, size_t n_frames, int mfcc_size, int precision,
                           float *logits) {
    return guarded([&]() -> int {
        if (!ctx || !model) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        const Model &m = *model->impl;
        const int nl = (int)m.dims.size() - 1, K = mfcc_size;
        if (precision != RP_MLP_F32 && precision != RP_MLP_BF16 && precision != RP_MLP_F32_STRICT && precision != RP_MLP_F32_FAST) { set_last_error("unknown MLP precision"); return -1; }
        if (K < 1 || m.dims[0] % K != 0) { set_last_error("Model input size does not match the mfcc size"); return -1; }
        const int L = m.dims[0] / K;
        const size_t n_win = n_frames >= (size_t)L ? n_frames - L + 1 : 0, rows = S * n_win;
        if (rows == 0) return 0;
        if (!mfcc || !logits) { set_last_error("null argument"); return -1; }
        Staged sg(c);
        const float *dm = static_cast<const float *>(sg.in(mfcc, S * n_frames * (size_t)K * 4, c->stage_in));
        float *dl = static_cast<float *>(sg.out(logits, rows * (size_t)m.dims[nl] * 4, c->stage_out));
        if (!dm || !dl) return -1;
        if (!window_logits(c, m, dm, S, n_frames, n_win, L, K, precision, dl)) return -1;
        if (!sg.back(logits, dl, rows * (size_t)m.dims[nl] * 4) || !sg.finish()) return -1;
        return 0;
    });
}

int rp_batch_detect_model(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                          const rp_model *model, int mfcc_size, int none_index, const rp_detector_config *config, int precision,
                          rp_batch_detection *det, int32_t *det_label, int32_t *n_det, int max_det) {
    return guarded([&]() -> int {
        if (!ctx || !model) { set_last_error("null handle"); return -1; }
        if (!config || (S && (!pcm || !det || !n_det))) { set_last_error("null argument"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        if (pcm_stride < n_samples) { set_last_error("pcm_stride smaller than n_samples"); return -1; }
        if ((int)fmt < 0 || (int)fmt > 3) { set_last_error("unknown sample format"); return -1; }
        if (precision != RP_MLP_F32 && precision != RP_MLP_BF16 && precision != RP_MLP_F32_STRICT && precision != RP_MLP_F32_FAST) { set_last_error("unknown MLP precision"); return -1; }
        const Model &m = *model->impl;
        const int nl_layers = (int)m.dims.size() - 1, K = mfcc_size;
        if (K < 1 || m.dims[0] % K != 0) { set_last_error("Incorrect model layers"); return -1; }
        const int L = m.dims[0] / K, n_labels = m.dims[nl_layers];
        if (none_index >= n_labels) { set_last_error("none_index out of range"); return -1; }
        if (!m.mfma_ok && precision == RP_MLP_BF16) { set_last_error("this layer-1 shape has no bf16 MFMA kernel"); return -1; }
        const MfccTablesDev *tb = c->tables_for(K);
        if (!tb) return -1;
        const size_t nf = rp_mfcc_num_frames(n_samples);
        const size_t n_win = nf >= (size_t)L ? nf - L + 1 : 0, rows = S * n_win;
        Staged sg(c);
        const void *dp = sg.in(pcm, S * pcm_stride * sample_bytes(fmt), c->stage_in);
        BatchDetection *dd = static_cast<BatchDetection *>(sg.out(det, S * (size_t)max_det * sizeof(BatchDetection), c->stage_out));
        int32_t *dn = static_cast<int32_t *>(sg.out(n_det, S * sizeof(int32_t), c->stage_out2));
        int32_t *dl = det_label ? static_cast<int32_t *>(sg.out(det_label, S * (size_t)max_det * sizeof(int32_t), c->stage_out3)) : nullptr;
        if (S && (!dp || !dd || !dn)) return -1;
        if (!c->ws_mfcc.reserve(S * nf * K * sizeof(float) + 64)) return -1;
        float *dm = c->ws_mfcc.as<float>();
        c->time_begin(kKernelMfcc);
        bool ok = hip_ok(launch_mfcc_fmt(c->stream, *tb, dp, (int)fmt, S, n_samples, pcm_stride, 0, nf, nf, dm), "mfcc_kernel");
        c->time_end();
        if (!ok) return -1;
        if (!c->ws_ring.reserve(rows * (size_t)n_labels * sizeof(float) + 16) || !c->ws_agg.reserve(rows * sizeof(float) + 16) ||
            !c->ws_avg.reserve(rows * sizeof(float) + 16) || !c->ws_rms.reserve(rows * sizeof(int32_t) + 16)) return -1;
        float *dlog = c->ws_ring.as<float>();
        if (!window_logits(c, m, dm, S, nf, n_win, L, K, precision, dlog)) return -1;
        float *dg = c->ws_agg.as<float>(), *da = c->ws_avg.as<float>();
        int32_t *dlab = c->ws_rms.as<int32_t>();
        // nn_score_kernel also tells the scan which streams have a window that passed (a flag per stream, like the aggregate pass of
        // the reference path): the others are not swept
        uint32_t *hot = nullptr;
        if (n_win) {
            hot = c->hot_flags(S);
            if (!hot) return -1;
        }
        if (!hip_ok(launch_nn_score(c->stream, dlog, rows, n_labels, none_index, config->score_ref * 10.f, config->avg_threshold != 0.f ? 1 : 0,
                                    config->threshold, config->avg_threshold, dg, da, dlab, hot, n_win), "nn_score_kernel")) return -1;
        ScanWakewords ww{};
        ww.n = 1; ww.agg[0] = dg; ww.avg[0] = da; ww.label[0] = dlab; ww.hot = hot;
        ww.threshold[0] = -1.f; ww.avg_threshold[0] = -1.f;  // the gates were applied by nn_score_kernel (>=, not >)
        ScanConfig sc;
        sc.threshold = config->threshold; sc.avg_threshold = config->avg_threshold; sc.min_scores = (int)config->min_scores;
        sc.eager = config->eager ? 1 : 0; sc.max_len = L; sc.avg_enabled = 0;
        float *dv = nullptr;
        if (config->vad_mode != RP_VAD_NONE) {
            if (!c->ws_vad.reserve(S * nf * sizeof(float) + 16)) return -1;
            dv = c->ws_vad.as<float>();
            if (!hip_ok(launch_vad_value(c->stream, dm, S * nf, K, dv), "vad_value_kernel")) return -1;
        }
        c->time_begin(kKernelScan);
        ok = hip_ok(launch_scan_multi(c->stream, ww, dv, vad_mode_value(config->vad_mode), S, nf, sc, dd, dl, dn, max_det), "scan_kernel");
        c->time_end();
        if (!ok) return -1;
        if (!sg.back(det, dd, S * (size_t)max_det * sizeof(BatchDetection)) || !sg.back(n_det, dn, S * sizeof(int32_t))) return -1;
        if (dl && !sg.back(det_label, dl, S * (size_t)max_det * sizeof(int32_t))) return -1;
        return sg.finish() ? 0 : -1;
    });
}

int rp_synth_pcm_batch(rp_ctx *ctx, uint64_t seed, uint64_t first_stream, size_t S, size_t n_samples, size_t pcm_stride,
                       float *pcm) {
    return guarded([&]() -> int {
        if (!ctx) { set_last_error("null handle"); return -1; }
        Ctx *c = ctx->impl.get();
        if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
        Staged sg(c);
        float *dp = static_cast<float *>(sg.out(pcm, S * pcm_stride * sizeof(float), c->stage_out));
        if (S && n_samples && !dp) return -1;
        if (!hip_ok(launch_synth(c->stream, seed, first_stream, S, n_samples, pcm_stride, dp), "synth_kernel")) return -1;
        if (!sg.back(pcm, dp, S * pcm_stride * sizeof(float)) || !sg.finish()) return -1;
        return 0;
    });
}

int rp_ctx_timing_enable(rp_ctx *ctx, int enable) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    ctx->impl->timing = enable != 0;
    return 0;
}
int rp_ctx_timing_reset(rp_ctx *ctx) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    ctx->impl->time_collect();
    for (int i = 0; i < kKernelCount; ++i) { ctx->impl->sum_ms[i] = 0; ctx->impl->count[i] = 0; }
    return 0;
}
int rp_ctx_timing_read(rp_ctx *ctx, int kernel, double *avg_ms, int *launches) {
    if (!ctx) { set_last_error("null handle"); return -1; }
    if (kernel < 0 || kernel >= kKernelCount) { set_last_error("unknown kernel id"); return -1; }
    ctx->impl->time_collect();
    int n = ctx->impl->count[kernel];
    if (avg_ms) *avg_ms = n ? ctx->impl->sum_ms[kernel] / n : 0.0;
    if (launches) *launches = n;
    return 0;
}

}  // extern "C"
