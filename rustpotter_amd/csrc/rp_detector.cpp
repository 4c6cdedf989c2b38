// rp_detector.cpp -- C++ mirror of the reference's `Rustpotter` (src/detector.rs) for a
// single live stream.  Host code keeps the integer state machine, VAD and the two
// (sequential, per-sample) audio filters; every 30 ms chunk's MFCC frames and window
// scores come from the HIP kernels -- there is no CPU implementation of either here.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>

#include "rp_host.h"

namespace rp {

namespace {
constexpr float kPi = 3.14159274101257324f;
constexpr const char *kNoneLabel = "none";  // src/constants.rs:12

// GainNormalizerFilter::get_rms_level, src/audio/gain_normalizer_filter.rs:49-55
float rms_level_of(const float *s, int n) {
    float sum_squared = 0.0f;
    for (int i = 0; i < n; ++i) sum_squared += s[i] * s[i];
    return std::sqrt(sum_squared / (float)n);
}
// src/wakewords/nn/wakeword_nn.rs:161-163
float calc_inverse_similarity(float n1, float n2, float reference) {
    return 1.f - (1.f / (1.f + std::exp(((n1 - n2) - reference) / reference)));
}
}  // namespace

struct Rustpotter::Wakeword {
    bool is_model = false;
    WakewordRefData ref;
    std::unique_ptr<Templates> tmpl;
    WakewordModelData model;
    int n_layers = 0, none_index = -1;
    std::vector<int> dims;
    std::unique_ptr<Model> net;  // device
    // per-call result offsets (floats) inside result_
    size_t off_scores = 0, off_avg = 0, off_agg = 0, off_logits = 0;
    bool with_avg = false, shape_ok = true;
    size_t frame_size() const {  // get_mfcc_frame_size
        if (is_model) return model.train_size;
        int m = 0; for (int l : ref.lens) m = std::max(m, l); return (size_t)m;
    }
    float rms_level() const { return is_model ? model.rms_level : ref.rms_level; }
};

// ------------------------------------------------------------------------- VAD
// src/mfcc/vad.rs:11-50
void Rustpotter::Vad::reset() { for (float &w : window) w = NAN; voice_countdown = 0; index = 0; }
bool Rustpotter::Vad::is_voice(const float *mfcc, int K) {
    float s = 0.f;
    for (int i = 0; i < K; ++i) s += std::fabs(mfcc[i]);
    window[index] = s / (float)K;
    index = index >= 49 ? 0 : index + 1;
    float mn = INFINITY; bool any = false;
    for (float w : window) if (!std::isnan(w) && (!any || w < mn)) { mn = w; any = true; }
    mn = std::fmax(mn, 0.01f);
    const float th = mn * mode_value;
    int n_high = 0;
    for (float w : window) if (w > th) ++n_high;
    if (n_high > 10) voice_countdown = 500;
    if (voice_countdown > 0) { --voice_countdown; return true; }
    return false;
}

// --------------------------------------------------------------------- lifecycle
Rustpotter *Rustpotter::create(const rp_config &cfg) {
    if (cfg.fmt.channels < 1) { set_last_error("Unsupported channel count"); return nullptr; }
    std::unique_ptr<Rustpotter> r(new Rustpotter());
    // AudioEncoder::new, src/audio/encoder.rs:63-80
    if (!resampler_frame_lengths(cfg.fmt.sample_rate, &r->in_len_, &r->out_len_)) {
        set_last_error("Unsupported sample rate, unable to initialize the resampler");
        return nullptr;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);  // the calling thread's current device (one process per GPU)
    r->ctx_.reset(Ctx::create(dev, RP_CTX_DEVICE_POINTERS));
    if (!r->ctx_) return nullptr;
    if (cfg.fmt.sample_rate != 16000) {
        r->rs_ = r->ctx_->resampler_for(cfg.fmt.sample_rate);
        if (!r->rs_) return nullptr;
        if (!r->rs_x_.reserve(2 * r->in_len_ * sizeof(float)) || !r->enc_.reserve(r->out_len_ * sizeof(float))) return nullptr;
        std::memset(r->rs_x_.p, 0, 2 * r->in_len_ * sizeof(float));
    }
    r->fmt_ = cfg.fmt;
    r->det_ = cfg.detector;
    r->filt_ = cfg.filters;
    r->has_vad_ = cfg.detector.vad_mode != RP_VAD_NONE;
    r->vad_.mode_value = cfg.detector.vad_mode == RP_VAD_EASY ? 2.f : cfg.detector.vad_mode == RP_VAD_MEDIUM ? 2.5f : 3.f;
    r->vad_.reset();
    r->configure_filters();
    return r.release();
}

Rustpotter::~Rustpotter() { wakewords_.clear(); }

void Rustpotter::configure_filters() {
    // From<&GainNormalizationConfig>, gain_normalizer_filter.rs:68-80 / GainNormalizerFilter::new :56-66
    const rp_gain_normalization_config &g = filt_.gain_normalizer;
    gainf_ = Gain();
    gainf_.enabled = g.enabled;
    gainf_.min_gain = g.min_gain; gainf_.max_gain = g.max_gain;
    gainf_.fixed = g.has_gain_ref;
    gainf_.rms_level_ref = g.has_gain_ref ? g.gain_ref : NAN;
    gainf_.rms_level_sqrt = g.has_gain_ref ? std::sqrt(g.gain_ref) : NAN;
    gainf_.window_size = 1;
    // BandPassFilter::new, band_pass_filter.rs:31-55
    const rp_band_pass_config &b = filt_.band_pass;
    bp_ = BandPass();
    bp_.enabled = b.enabled;
    if (b.enabled) {
        const float sample_rate = 16000.f;
        float omega_low = 2.0f * kPi * b.low_cutoff / sample_rate, omega_high = 2.0f * kPi * b.high_cutoff / sample_rate;
        float cos_low = std::cos(omega_low), cos_high = std::cos(omega_high);
        float alpha_low = std::sin(omega_low) / 2.0f, alpha_high = std::sin(omega_high) / 2.0f;
        float a0 = 1.0f / (1.0f + alpha_high - alpha_low);
        bp_.a0 = a0; bp_.a1 = -2.0f * cos_low * a0; bp_.a2 = (1.0f - alpha_high - alpha_low) * a0;
        bp_.b1 = -2.0f * cos_high * a0; bp_.b2 = (1.0f - alpha_high + alpha_low) * a0;
        bp_.x1 = bp_.x2 = bp_.y1 = bp_.y2 = 0.f;
    }
}

// Rustpotter::reset, src/detector.rs:290-302
void Rustpotter::reset() {
    has_partial_ = false;
    win_len_ = 0;
    n_hist_ = 0;
    shifts_seen_ = 0;  // mfcc_extractor.reset()
    if (has_vad_) vad_.reset();
}

void Rustpotter::update_detector_config(const rp_detector_config &c) {  // src/detector.rs:263-281
    det_ = c;
    has_vad_ = c.vad_mode != RP_VAD_NONE;
    vad_.mode_value = c.vad_mode == RP_VAD_EASY ? 2.f : c.vad_mode == RP_VAD_MEDIUM ? 2.5f : 3.f;
    vad_.reset();
    reset();
}
void Rustpotter::update_filters_config(const rp_filters_config &c) {  // src/detector.rs:284-288
    filt_ = c;
    configure_filters();
    // the new gain filter has no reference level until the wakeword set changes again
    // (the reference behaves the same: set_rms_level_ref is only called from on_wakeword_change)
    reset();
}

float Rustpotter::get_rms_level_ref() const { return gainf_.enabled ? gainf_.rms_level_ref : NAN; }

size_t Rustpotter::get_bytes_per_frame() const {
    size_t b = fmt_.sample_format == RP_SAMPLE_I8 ? 1 : fmt_.sample_format == RP_SAMPLE_I16 ? 2 : 4;
    return get_samples_per_frame() * b;
}

// on_wakeword_change, src/detector.rs:328-346
void Rustpotter::on_wakeword_change() {
    size_t mx = 0;
    float target = NAN;
    for (auto &kv : wakewords_) {
        mx = std::max(mx, kv.second->frame_size());
        target = std::fmax(kv.second->rms_level(), target);  // f32::max ignores NaN
    }
    max_mfcc_frames_ = mx;
    if (gainf_.enabled) {  // set_rms_level_ref, gain_normalizer_filter.rs:42-48
        if (!gainf_.fixed) { gainf_.rms_level_ref = target; gainf_.rms_level_sqrt = std::sqrt(target); }
        size_t ws = max_mfcc_frames_ / 3;
        gainf_.window_size = ws != 0 ? ws : 1;
    }
}

bool Rustpotter::prepare_first(int K) {  // add_wakeword with an empty registry, src/detector.rs:305-307
    reset();
    K_ = K;  // set_out_size
    if (!ctx_->tables_for(K)) return false;
    hist_cap_ = 0;  // row width changed
    return true;
}

bool Rustpotter::add_wakeword_ref(const std::string &key, WakewordRefData &&ref) {
    if (wakewords_.empty()) { if (!prepare_first(ref.mfcc_size)) return false; }
    else if (K_ != ref.mfcc_size) {
        set_last_error("Usage of wakewords with different mfcc size is not supported, ignoring wakeword");
        return false;
    }
    std::unique_ptr<Wakeword> w(new Wakeword());
    w->ref = std::move(ref);
    std::vector<float> flat;
    for (auto &f : w->ref.feats) flat.insert(flat.end(), f.begin(), f.end());
    w->tmpl.reset(Templates::create(ctx_.get(), (int)w->ref.lens.size(), w->ref.mfcc_size, w->ref.lens.data(), flat.data(),
                                    w->ref.has_avg ? w->ref.avg_len : 0, w->ref.has_avg ? w->ref.avg.data() : nullptr));
    if (!w->tmpl) return false;
    bool replaced = false;
    for (auto &kv : wakewords_) if (kv.first == key) { kv.second = std::move(w); replaced = true; break; }
    if (!replaced) wakewords_.emplace_back(key, std::move(w));
    on_wakeword_change();
    return true;
}

bool Rustpotter::add_wakeword_model(const std::string &key, WakewordModelData &&model) {
    if (wakewords_.empty()) { if (!prepare_first(model.mfcc_size)) return false; }
    else if (K_ != model.mfcc_size) {
        set_last_error("Usage of wakewords with different mfcc size is not supported, ignoring wakeword");
        return false;
    }
    std::unique_ptr<Wakeword> w(new Wakeword());
    w->is_model = true;
    w->model = std::move(model);
    // init_model, src/wakewords/nn/wakeword_nn.rs:165-389: Tiny has ln1,ln2; the others ln1..ln3
    std::string mt = w->model.m_type;
    std::transform(mt.begin(), mt.end(), mt.begin(), ::tolower);
    w->n_layers = mt == "tiny" ? 2 : (mt == "small" || mt == "medium" || mt == "large") ? 3 : 0;
    if (w->n_layers == 0) { set_last_error("Unknown model type"); return false; }
    std::vector<const float *> wp, bp;
    for (int l = 0; l < w->n_layers; ++l) {
        auto wi = w->model.weights.find("ln" + std::to_string(l + 1) + ".weight");
        auto bi = w->model.weights.find("ln" + std::to_string(l + 1) + ".bias");
        if (wi == w->model.weights.end() || bi == w->model.weights.end() || wi->second.first.size() != 2) {
            set_last_error("Incorrect model layers");  // wakeword_nn.rs:252-256
            return false;
        }
        if (l == 0) w->dims.push_back((int)wi->second.first[1]);
        else if ((int)wi->second.first[1] != w->dims.back()) { set_last_error("Incorrect model layers"); return false; }
        w->dims.push_back((int)wi->second.first[0]);
        if (bi->second.second.size() != wi->second.first[0]) { set_last_error("Incorrect model layers"); return false; }
        // the tensor must hold exactly out x in values (the dims are attacker-controlled: Model::create copies out*in floats)
        if (wi->second.first[0] == 0 || wi->second.first[1] == 0 || wi->second.first[0] > 0x7fffffffULL || wi->second.first[1] > 0x7fffffffULL ||
            wi->second.second.size() / wi->second.first[0] != wi->second.first[1] ||
            wi->second.second.size() % wi->second.first[0] != 0) { set_last_error("Incorrect model layers"); return false; }
        wp.push_back(wi->second.second.data());
        bp.push_back(bi->second.second.data());
    }
    w->net.reset(Model::create(ctx_.get(), w->n_layers, w->dims.data(), wp.data(), bp.data()));
    if (!w->net) return false;
    if (w->dims.back() != (int)w->model.labels.size()) { set_last_error("Incorrect model layers"); return false; }
    for (size_t i = 0; i < w->model.labels.size(); ++i) if (w->model.labels[i] == kNoneLabel) { w->none_index = (int)i; break; }
    bool replaced = false;
    for (auto &kv : wakewords_) if (kv.first == key) { kv.second = std::move(w); replaced = true; break; }
    if (!replaced) wakewords_.emplace_back(key, std::move(w));
    on_wakeword_change();
    return true;
}

bool Rustpotter::add_wakeword_from_buffer(const std::string &key, const uint8_t *buf, size_t len) {
    RpwKind kind; WakewordRefData ref; WakewordModelData model; std::string err;
    if (!parse_rpw(buf, len, &kind, &ref, &model, &err)) { set_last_error(err); return false; }
    return kind == RpwKind::Ref ? add_wakeword_ref(key, std::move(ref)) : add_wakeword_model(key, std::move(model));
}

bool Rustpotter::add_wakeword_from_file(const std::string &key, const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_last_error("Unable to open file " + path + ": " + std::strerror(errno)); return false; }  // wakeword_file.rs:28-33
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return add_wakeword_from_buffer(key, buf.data(), buf.size());
}

bool Rustpotter::remove_wakeword(const std::string &key) {  // src/detector.rs:180-189
    size_t len = wakewords_.size();
    wakewords_.erase(std::remove_if(wakewords_.begin(), wakewords_.end(), [&](const auto &kv) { return kv.first == key; }),
                     wakewords_.end());
    if (len != wakewords_.size()) { on_wakeword_change(); return true; }
    return false;
}
bool Rustpotter::remove_wakewords() {  // src/detector.rs:193-202
    size_t len = wakewords_.size();
    wakewords_.clear();
    if (len != 0) { on_wakeword_change(); return true; }
    return false;
}

// ------------------------------------------------------------------- processing
// encode_audio_bytes + reencode_to_mono, src/audio/encoder.rs:26-47,104-116 (no resampler)
int Rustpotter::process_bytes(const uint8_t *bytes, size_t len, Detection *out) {
    if (len != get_bytes_per_frame()) return 0;  // src/detector.rs:235-237
    const size_t ch = fmt_.channels;
    std::vector<float> mono(in_len_);
    float *buf = mono.data();
    const bool le = fmt_.endianness == RP_ENDIAN_LITTLE || fmt_.endianness == RP_ENDIAN_NATIVE;  // host is little-endian
    for (size_t i = 0; i < in_len_; ++i) {
        const size_t si = i * ch;  // first channel of each interleaved frame
        switch (fmt_.sample_format) {
        case RP_SAMPLE_I8: buf[i] = (float)(int8_t)bytes[si] / 127.f; break;
        case RP_SAMPLE_I16: {
            const uint8_t *p = bytes + si * 2;
            uint16_t u = le ? (uint16_t)(p[0] | (p[1] << 8)) : (uint16_t)(p[1] | (p[0] << 8));
            buf[i] = (float)(int16_t)u / 32767.f; break;
        }
        case RP_SAMPLE_I32: {
            const uint8_t *p = bytes + si * 4;
            uint32_t u = le ? ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24))
                            : ((uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24));
            buf[i] = (float)(int32_t)u / 2147483648.f; break;  // i32::MAX as f32 == 2^31
        }
        default: {
            const uint8_t *p = bytes + si * 4;
            uint32_t u = le ? ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24))
                            : ((uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24));
            std::memcpy(&buf[i], &u, 4); break;
        }
        }
    }
    return encode_and_process(buf, out);
}

template <class T> static inline float sample_into_f32(T v);
template <> inline float sample_into_f32<int8_t>(int8_t v) { return (float)v / 127.f; }          // audio_types.rs:98-107
template <> inline float sample_into_f32<int16_t>(int16_t v) { return (float)v / 32767.f; }      // :108-117
template <> inline float sample_into_f32<int32_t>(int32_t v) { return (float)v / 2147483648.f; } // :118-127
template <> inline float sample_into_f32<float>(float v) { return v; }                           // :128-137

template <class T> int Rustpotter::process_samples(const T *samples, size_t n, Detection *out) {
    if (n != get_samples_per_frame()) return 0;  // src/detector.rs:249-251
    const size_t ch = fmt_.channels;
    std::vector<float> mono(in_len_);
    for (size_t i = 0; i < in_len_; ++i) mono[i] = sample_into_f32<T>(samples[i * ch]);
    return encode_and_process(mono.data(), out);
}

// reencode_to_mono_with_sample_rate, src/audio/encoder.rs:41-60: input that is not 16 kHz goes through the
// resampler (one launch; the previous input frame stays on the device), then process_audio sees out_len samples
int Rustpotter::encode_and_process(float *mono, Detection *out) {
    if (!rs_) return process_audio(mono, in_len_, out);
    hipStream_t st = ctx_->stream;
    if (!hip_ok(hipSetDevice(ctx_->device), "hipSetDevice")) return -1;
    float *x2 = rs_x_.as<float>();  // the previous call has been synchronised: the host may rewrite the buffer
    std::memcpy(x2, x2 + in_len_, in_len_ * sizeof(float));
    std::memcpy(x2 + in_len_, mono, in_len_ * sizeof(float));
    if (!hip_ok(launch_resample(st, rs_->dev, rs_x_.dev_as<float>(), 1, 1, enc_.dev_as<float>(), out_len_), "resample kernel") ||
        !hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"))
        return -1;
    if (wakewords_.empty()) return 0;  // the encoder runs before process_audio's early return (src/detector.rs:252-254,347-350)
    return process_audio(enc_.as<float>(), out_len_, out);
}
template int Rustpotter::process_samples<int8_t>(const int8_t *, size_t, Detection *);
template int Rustpotter::process_samples<int16_t>(const int16_t *, size_t, Detection *);
template int Rustpotter::process_samples<int32_t>(const int32_t *, size_t, Detection *);
template int Rustpotter::process_samples<float>(const float *, size_t, Detection *);

// process_audio, src/detector.rs:347-376
int Rustpotter::process_audio(float *buf, size_t n, Detection *out) {
    if (wakewords_.empty()) return 0;
    rms_level_ = rms_level_of(buf, n);
    if (gainf_.enabled) {  // GainNormalizerFilter::filter, gain_normalizer_filter.rs:14-41
        float gain = 1.f;
        if (!std::isnan(gainf_.rms_level_ref) && rms_level_ != 0.f) {
            gainf_.win.push_back(rms_level_);
            if (gainf_.win.size() > gainf_.window_size) gainf_.win.erase(gainf_.win.begin());
            float s = 0.f;
            for (float v : gainf_.win) s += v;
            float frame_rms_level = s / (float)gainf_.win.size();
            gain = gainf_.rms_level_sqrt / std::sqrt(frame_rms_level);
            gain = std::round(gain * 10.f) / 10.f;
            if (gain < gainf_.min_gain) gain = gainf_.min_gain;
            if (gain > gainf_.max_gain) gain = gainf_.max_gain;
            if (gain != 1.f)
                for (size_t i = 0; i < n; ++i) { float v = buf[i] * gain; if (v < -1.f) v = -1.f; if (v > 1.f) v = 1.f; buf[i] = v; }
        }
        gain_ = gain;
    }
    if (bp_.enabled) {  // BandPassFilter::filter, band_pass_filter.rs:19-30
        for (size_t i = 0; i < n; ++i) {
            float x = buf[i];
            float y = bp_.a0 * x + bp_.a1 * bp_.x1 + bp_.a2 * bp_.x2 - bp_.b1 * bp_.y1 - bp_.b2 * bp_.y2;
            buf[i] = y;
            bp_.x2 = bp_.x1; bp_.x1 = x; bp_.y2 = bp_.y1; bp_.y1 = y;
        }
    }
    // MfccExtractor::compute, src/mfcc/extractor.rs:60-79: chunks_exact(160) shifts (a remainder is dropped); after a
    // reset the first three shifts only fill the extractor, every later shift yields one frame made of the last
    // three.  The kernel's frame j covers shifts j+1..j+3 of the buffer it is given, so it gets
    // [one shift of padding | the two buffered shifts | the new shifts] and is asked for the frames that exist.
    const size_t n_shifts = n / 160;
    if (n_shifts == 0) return 0;
    const size_t i0 = shifts_seen_ >= 3 ? 0 : 3 - shifts_seen_;       // first new shift that completes a frame
    const size_t nfr = n_shifts > i0 ? n_shifts - i0 : 0;             // frames this call
    const size_t up_len = 480 + n_shifts * 160;
    if (up_.cap < up_len * sizeof(float)) {  // grows once: the buffered shifts move along
        float keep[480] = {0.f};
        if (up_.p) std::memcpy(keep, up_.p, sizeof(keep));
        if (!up_.reserve(up_len * sizeof(float))) return -1;
        std::memcpy(up_.p, keep, sizeof(keep));
    }
    float *up = up_.as<float>();
    std::memcpy(up + 480, buf, n_shifts * 160 * sizeof(float));
    auto keep_last_two = [&]() {  // the two newest shifts become the extractor's buffered ones
        std::memmove(up + 160, up + 160 + n_shifts * 160, 320 * sizeof(float));
        shifts_seen_ = std::min<size_t>(3, shifts_seen_ + n_shifts);
    };
    if (nfr == 0) { keep_last_two(); return 0; }
    const int K = K_;
    const int NF = (int)nfr;
    hipStream_t st = ctx_->stream;
    if (!hip_ok(hipSetDevice(ctx_->device), "hipSetDevice")) return -1;
    // window history: grow / compact so that the new frames fit
    const size_t want_cap = 4 * std::max<size_t>(max_mfcc_frames_, 64) + 16 + nfr;
    if (hist_cap_ < want_cap || n_hist_ + nfr > hist_cap_) {
        const size_t keep = std::min(win_len_, n_hist_);
        const size_t new_cap = std::max(want_cap, 2 * keep + 16 + nfr);
        DevBuf nb;
        if (!nb.reserve(new_cap * K * sizeof(float))) return -1;
        if (keep && hist_.p &&
            !hip_ok(hipMemcpyAsync(nb.p, hist_.as<float>() + (n_hist_ - keep) * K, keep * K * sizeof(float),
                                   hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(hist)")) return -1;
        if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
        std::swap(hist_.p, nb.p); std::swap(hist_.cap, nb.cap);
        hist_cap_ = new_cap; n_hist_ = keep;
    }
    const MfccTablesDev *tb = ctx_->tables_for(K);
    if (!tb) return -1;
    float *hist = hist_.as<float>();

    // which of the new frames complete a window (process_new_mfccs, src/detector.rs:384-395)
    size_t wl = win_len_, first_win = 0;
    int w0 = -1;
    for (int i = 0; i < NF; ++i) {
        wl += 1;
        if (wl >= max_mfcc_frames_) { if (w0 < 0) { w0 = i; first_win = n_hist_ + i + 1 - wl; } wl -= 1; }
    }
    const size_t cnt = w0 < 0 ? 0 : (size_t)(NF - w0);
    // result buffer: [NF][K] frames, then per wakeword its score blocks
    size_t total = (size_t)NF * (size_t)K;
    for (auto &kv : wakewords_) {
        Wakeword &w = *kv.second;
        if (w.is_model) { w.off_logits = total; total += cnt * w.model.labels.size(); }
        else { const size_t T = w.ref.lens.size(); w.off_scores = total; total += cnt * T; w.off_avg = total; total += cnt; w.off_agg = total; total += cnt; }
    }
    if (!result_.reserve(total * sizeof(float))) return -1;
    float *res = result_.dev_as<float>();  // page-locked host memory: the kernels write the results where the host reads them
    // the new frames go behind the window history and, packed, to the front of the result buffer
    if (!hip_ok(launch_mfcc(st, *tb, up_.dev_as<float>(), 1, up_len, up_len, i0, nfr, nfr, hist + n_hist_ * K, res), "mfcc_kernel")) return -1;
    // The averaged-template gate (WakewordComparator::run_detection :83-93): `avg_score < avg_threshold -> None`, the sample
    // templates are never compared.  One wave scores one (window, template) pair, so all T+1 DTWs of a chunk run side by side
    // in one 19 us launch and skipping T of them saves no time by itself; scoring the averaged template FIRST and the others
    // only when a window is left costs a second launch + synchronise when the gate lets the window through (75 against 44 us
    // per call on noise, whose avg_score 0.28-0.40 passes the default 0.2) and saves the aggregate launch and T DTWs of
    // device work when it does not (40 us).  So the order is chosen from the previous chunk: gate-first while the gate has
    // been rejecting every window (silence, quiet rooms), everything at once otherwise; a wrong guess costs one slow call.
    std::vector<Wakeword *> gated;
    const size_t frames_valid = n_hist_ + nfr;
    if (cnt) {
        for (auto &kv : wakewords_) {
            Wakeword &w = *kv.second;
            if (!w.is_model) {
                // WakewordComparator::run_detection :83-85: the avg DTW only runs when avg_threshold != 0
                const float avg_thr = w.ref.has_avg_threshold ? w.ref.avg_threshold : det_.avg_threshold;
                w.with_avg = w.ref.has_avg && avg_thr != 0.f;
                const int T = (int)w.ref.lens.size();
                if (w.with_avg && gate_first_) {
                    const hipError_t e = launch_dtw_single_part(st, ctx_->dtw_work(), w.tmpl->dev, hist, frames_valid, first_win, cnt, cnt, det_.band_size,
                                                                det_.score_ref, T, 1, res + w.off_scores, res + w.off_avg);
                    if (e == hipSuccess) { gated.push_back(&w); continue; }
                    if (e != hipErrorNotSupported) { hip_ok(e, "dtw kernel"); return -1; }
                    (void)hipGetLastError();
                }
                if (!hip_ok(launch_dtw(st, ctx_->dtw_work(), w.tmpl->dev, hist, 1, frames_valid, first_win, cnt, cnt, det_.band_size, det_.score_ref,
                                       w.with_avg ? 1 : 0, res + w.off_scores, res + w.off_avg), "dtw kernel")) return -1;
                if (!hip_ok(launch_aggregate(st, res + w.off_scores, cnt, T, (int)det_.score_mode, res + w.off_agg), "aggregate_kernel")) return -1;
            } else {
                const int L = (int)w.model.train_size;
                w.shape_ok = (size_t)L * K == (size_t)w.dims[0] && first_win + cnt - 1 + L <= frames_valid;
                if (!w.shape_ok) continue;  // candle shape error -> None, wakeword_nn.rs:107-111
                int maxd = 0; for (int d : w.dims) maxd = std::max(maxd, d);
                if (!nn_x_.reserve(cnt * (size_t)w.dims[0] * 4) || !nn_s0_.reserve(cnt * (size_t)maxd * 4) || !nn_s1_.reserve(cnt * (size_t)maxd * 4)) return -1;
                if (!hip_ok(launch_normalize_windows(st, hist, first_win, cnt, L, K, nn_x_.as<float>()), "normalize_windows_kernel")) return -1;
                // exact-f32 path (f32-input MFMA == fmaf chain); bf16 is only offered on the batched operator
                if (w.net->mfma_ok) {
                    uint32_t *redo = ctx_->mlp_redo(cnt);
                    if (!redo || !hip_ok(launch_mlp_mfma(st, w.net->dev, nn_x_.as<float>(), cnt, kMlpF32, res + w.off_logits, redo), "mlp_mfma_kernel")) return -1;
                } else if (!hip_ok(launch_mlp(st, nn_x_.as<float>(), cnt, w.n_layers, w.dims.data(), w.net->W.data(), w.net->B.data(),
                                              nn_s0_.as<float>(), nn_s1_.as<float>(), res + w.off_logits), "mlp kernel")) return -1;
            }
        }
    }
    bool any_passed = false, any_gate = false;   // for the next chunk's order
    if (!gated.empty()) {
        if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
        const float *hres = result_.as<float>();
        bool again = false;
        for (Wakeword *w : gated) {
            const float avg_thr = w->ref.has_avg_threshold ? w->ref.avg_threshold : det_.avg_threshold;
            bool any = false;
            for (size_t i = 0; i < cnt; ++i) any = any || !(hres[w->off_avg + i] < avg_thr);
            any_gate = true;
            if (!any) continue;   // every window of this chunk is `None` for this wakeword
            any_passed = true;
            const int T = (int)w->ref.lens.size();
            if (!hip_ok(launch_dtw_single_part(st, ctx_->dtw_work(), w->tmpl->dev, hist, frames_valid, first_win, cnt, cnt, det_.band_size, det_.score_ref, 0, T,
                                               res + w->off_scores, res + w->off_avg), "dtw kernel")) return -1;
            if (!hip_ok(launch_aggregate(st, res + w->off_scores, cnt, T, (int)det_.score_mode, res + w->off_agg), "aggregate_kernel")) return -1;
            again = true;
        }
        if (!again) { gate_first_ = true; keep_last_two(); goto scored; }
    }
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
    keep_last_two();  // after the synchronise: the upload above read up_
    if (cnt) {   // did the gate reject every window of every gated wakeword?  (all-at-once chunks: read the avg scores now)
        const float *hres = result_.as<float>();
        for (auto &kv : wakewords_) {
            Wakeword &w = *kv.second;
            if (w.is_model || !w.with_avg) continue;
            any_gate = true;
            const float avg_thr = w.ref.has_avg_threshold ? w.ref.avg_threshold : det_.avg_threshold;
            for (size_t i = 0; i < cnt; ++i) any_passed = any_passed || !(hres[w.off_avg + i] < avg_thr);
        }
        gate_first_ = any_gate && !any_passed;
    }
scored:

    // .into_iter().find_map(process_new_mfccs), src/detector.rs:372-397
    for (int i = 0; i < NF; ++i) {
        const float *frame = result_.as<float>() + (size_t)i * K;
        const bool should_run = has_partial_ || (has_vad_ ? vad_.is_voice(frame, K) : true);
        win_len_ += 1; n_hist_ += 1;
        bool fired = false;
        if (win_len_ >= max_mfcc_frames_ && should_run) fired = run_detection(i - w0, out);
        if (fired) return 1;  // reset() already cleared the window; the chunk's remaining frames are dropped
        if (win_len_ >= max_mfcc_frames_ && win_len_ > 0) win_len_ -= 1;  // drain(0..1)
    }
    return 0;
}

// run_detection + run_wakeword_detectors, src/detector.rs:398-447
bool Rustpotter::run_detection(int slot, Detection *out) {
    if (countdown_ != 0) countdown_ -= 1;
    if (has_partial_) {
        const bool done = countdown_ == 0 ? true : (det_.eager && partial_.counter >= det_.min_scores);  // :448-454
        if (done) {
            Detection taken = std::move(partial_);
            has_partial_ = false;
            if (taken.counter >= det_.min_scores) { reset(); *out = std::move(taken); return true; }
        }
    }
    const float *res = result_.as<float>();
    bool found = false;
    Detection best;
    for (auto &kv : wakewords_) {
        Wakeword &w = *kv.second;
        Detection d;
        bool ok = false;
        if (!w.is_model) {  // WakewordComparator::run_detection, wakeword_comp.rs:77-152
            const float avg_thr = w.ref.has_avg_threshold ? w.ref.avg_threshold : det_.avg_threshold;
            float avg_score = 0.f;
            if (w.ref.has_avg && avg_thr != 0.f) {
                avg_score = res[w.off_avg + slot];
                if (avg_score < avg_thr) continue;
            }
            const float thr = w.ref.has_threshold ? w.ref.threshold : det_.threshold;
            const size_t T = w.ref.lens.size();
            const float score = res[w.off_agg + slot];
            if (score > thr) {
                d.name = w.ref.name; d.avg_score = avg_score; d.score = score;
                d.score_names = w.ref.tnames;
                d.scores.assign(res + w.off_scores + (size_t)slot * T, res + w.off_scores + (size_t)(slot + 1) * T);
                ok = true;
            }
        } else if (w.shape_ok) {  // WakewordNN::run_detection, wakeword_nn.rs:39-159
            const size_t nl = w.model.labels.size();
            const float *logits = res + w.off_logits + (size_t)slot * nl;
            size_t bi = 0;
            for (size_t i = 1; i < nl; ++i) if (!(logits[i] < logits[bi])) bi = i;  // max_by(total_cmp): last maximum
            if ((int)bi == w.none_index) continue;
            const float ref = det_.score_ref * 10.f;
            const float none_prob = w.none_index >= 0 ? logits[w.none_index] : 0.f;
            const float label_prob = logits[bi];
            const bool calc_avg = det_.avg_threshold != 0.f;
            float second = 0.f;
            if (calc_avg) {  // max_by(|a,b| b.total_cmp(a)) over p != label_prob: the smallest other logit, :75-83
                bool any = false;
                for (size_t i = 0; i < nl; ++i) {
                    if (logits[i] == label_prob) continue;
                    if (!any || !(logits[i] > second)) { second = logits[i]; any = true; }
                }
                if (!any) second = 0.f;
            }
            d.name = w.model.labels[bi];
            d.avg_score = calc_avg ? calc_inverse_similarity(label_prob, second, ref) : 0.f;
            d.score = calc_inverse_similarity(label_prob, none_prob, ref);
            d.score_names = w.model.labels;
            d.scores.assign(logits, logits + nl);
            ok = d.score >= det_.threshold && d.avg_score >= det_.avg_threshold;  // validate_scores :113-123
        }
        if (ok && (!found || d.score > best.score)) { best = std::move(d); found = true; }
    }
    if (found) {
        best.counter = has_partial_ ? partial_.counter + 1 : 1;
        best.gain = gain_;
        if (!has_partial_ || partial_.score < best.score) { partial_ = std::move(best); has_partial_ = true; }
        else partial_.counter = best.counter;
        countdown_ = max_mfcc_frames_ / 2;
    }
    return false;
}

}  // namespace rp
