// rp_builder.cpp -- building and saving wakeword references: WakewordRef::new_from_sample_buffers /
// new_from_sample_files (src/wakewords/comp/wakeword_ref_build.rs), MfccWavFileExtractor::compute_mfccs
// (src/mfcc/wav_file_extractor.rs:18-69), MfccAverager::average (src/mfcc/averager.rs) with the unbanded
// Dtw + back-trace (src/mfcc/dtw.rs:11-55,106-138), and WakewordSave (src/wakewords/wakeword_file.rs:10-26,
// CBOR as ciborium writes it).  The MFCC frames come from the HIP kernel; the averaging is a short
// sequential host computation over a handful of templates (offline tooling in the reference too).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>

#include "rp_host.h"

namespace rp {
namespace {

// ------------------------------------------------------------------------- wav
struct Wav { int channels = 0, rate = 0, bits = 0; bool is_float = false; std::vector<float> mono; };

// hound::WavReader + AudioFmt::try_from (src/mfcc/wav_file_extractor.rs:93-112) + the per-type
// Sample::into_f32 (src/audio/audio_types.rs:98-137) + first-channel mono (src/audio/encoder.rs:41-47).
// 8-bit PCM is unsigned in the file and signed (-128) in hound.
bool parse_wav(const uint8_t *b, size_t n, Wav *w, std::string *err) {
    auto u16 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (n < 12 || std::memcmp(b, "RIFF", 4) != 0 || std::memcmp(b + 8, "WAVE", 4) != 0) { *err = "no RIFF tag found"; return false; }
    size_t p = 12, data_off = 0, data_len = 0;
    int fmt_tag = 0;
    bool have_fmt = false;
    while (p + 8 <= n) {
        const uint32_t sz = u32(p + 4);
        if (std::memcmp(b + p, "fmt ", 4) == 0 && p + 8 + 16 <= n) {
            fmt_tag = (int)u16(p + 8); w->channels = (int)u16(p + 10); w->rate = (int)u32(p + 12); w->bits = (int)u16(p + 22);
            if (fmt_tag == 0xFFFE && sz >= 40 && p + 8 + 26 <= n) fmt_tag = (int)u16(p + 8 + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
            have_fmt = true;
        } else if (std::memcmp(b + p, "data", 4) == 0) {
            data_off = p + 8; data_len = std::min<size_t>(sz, n - data_off);
            break;
        }
        p += 8 + (size_t)sz + (sz & 1);
    }
    if (!have_fmt || !data_off) { *err = "invalid wav file: missing fmt or data chunk"; return false; }
    w->is_float = fmt_tag == 3;
    const bool ok_fmt = (fmt_tag == 1 && (w->bits == 8 || w->bits == 16 || w->bits == 32)) || (fmt_tag == 3 && w->bits == 32);
    if (!ok_fmt || w->channels < 1) { *err = "Unsupported wav format"; return false; }  // wav_file_extractor.rs:109
    const size_t bps = (size_t)w->bits / 8, frame = bps * (size_t)w->channels, frames = data_len / frame;
    w->mono.resize(frames);
    for (size_t i = 0; i < frames; ++i) {
        const uint8_t *s = b + data_off + i * frame;  // first channel
        float v;
        if (w->is_float) { uint32_t u = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24); std::memcpy(&v, &u, 4); }
        else if (w->bits == 8) v = (float)(int8_t)(int)((int)s[0] - 128) / 127.f;
        else if (w->bits == 16) v = (float)(int16_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8)) / 32767.f;
        else v = (float)(int32_t)((uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24)) / 2147483648.f;
        w->mono[i] = v;
    }
    return true;
}

// GainNormalizerFilter::get_rms_level, src/audio/gain_normalizer_filter.rs:49-55
float rms_level_of(const float *s, int n) {
    float sum_squared = 0.0f;
    for (int i = 0; i < n; ++i) sum_squared += s[i] * s[i];
    return std::sqrt(sum_squared / (float)n);
}

// ------------------------------------------------------------------ averager
// src/mfcc/comparator.rs:15-48
float cosine_distance(const float *a, const float *b, int K) {
    float dot_ab = 0.f, dot_a = 0.f, dot_b = 0.f;
    for (int d = 0; d < K; ++d) { dot_ab += a[d] * b[d]; dot_a += a[d] * a[d]; dot_b += b[d] * b[d]; }
    const float magnitude = std::sqrt(dot_a * dot_b);
    return 1.f - (magnitude == 0.f ? 0.f : dot_ab / magnitude);
}
float min3_fold(float ins, float del, float mat) { return std::fmin(std::fmin(std::fmin(INFINITY, ins), del), mat); }

// one fold step of MfccAverager::average (src/mfcc/averager.rs:7-35): origin [m][K] <- aligned mean with frames [n][K]
void average_step(std::vector<float> &origin, int m, const std::vector<float> &frames, int n, int K) {
    std::vector<float> D((size_t)m * n);
    auto at = [&](int r, int c) -> float & { return D[(size_t)r * n + c]; };
    // Dtw::compute_optimal_path, src/mfcc/dtw.rs:11-55
    at(0, 0) = cosine_distance(&origin[0], &frames[0], K);
    for (int r = 1; r < m; ++r) at(r, 0) = cosine_distance(&origin[(size_t)r * K], &frames[0], K) + at(r - 1, 0);
    for (int c = 1; c < n; ++c) at(0, c) = cosine_distance(&origin[0], &frames[(size_t)c * K], K) + at(0, c - 1);
    for (int r = 1; r < m; ++r)
        for (int c = 1; c < n; ++c)
            at(r, c) = cosine_distance(&origin[(size_t)r * K], &frames[(size_t)c * K], K) + min3_fold(at(r - 1, c), at(r, c - 1), at(r - 1, c - 1));
    // retrieve_optimal_path, src/mfcc/dtw.rs:106-138: the vec starts with min(m-1, n-1) [0,0] entries, each move
    // pushes the NEW position (the end cell itself is never pushed), then the vec is reversed
    int r = m - 1, c = n - 1;
    std::vector<std::pair<int, int>> path((size_t)std::min(r, c), {0, 0});
    while (r > 0 || c > 0) {
        if (r > 0 && c > 0) {
            const float ins = at(r - 1, c), del = at(r, c - 1), mat = at(r - 1, c - 1), mn = min3_fold(ins, del, mat);
            if (mn == mat) { --r; --c; } else if (mn == ins) { --r; } else if (mn == del) { --c; }
        } else if (r > 0) { --r; } else { --c; }
        path.emplace_back(r, c);
    }
    std::reverse(path.begin(), path.end());
    std::vector<float> sum(origin);  // avgs[x][k] starts with origin[x][k], values appended in path order
    std::vector<int> cnt((size_t)m, 1);
    for (auto &pr : path) {
        for (int k = 0; k < K; ++k) sum[(size_t)pr.first * K + k] += frames[(size_t)pr.second * K + k];
        cnt[pr.first] += 1;
    }
    for (int x = 0; x < m; ++x)
        for (int k = 0; k < K; ++k) origin[(size_t)x * K + k] = sum[(size_t)x * K + k] / (float)cnt[x];
}

// ---------------------------------------------------------------------- CBOR
struct Cbor {
    std::vector<uint8_t> out;
    void head(int major, uint64_t v) {
        if (v < 24) out.push_back((uint8_t)((major << 5) | v));
        else if (v < 256) { out.push_back((uint8_t)((major << 5) | 24)); out.push_back((uint8_t)v); }
        else if (v < 65536) { out.push_back((uint8_t)((major << 5) | 25)); out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); }
        else if (v < (1ull << 32)) { out.push_back((uint8_t)((major << 5) | 26)); for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); }
        else { out.push_back((uint8_t)((major << 5) | 27)); for (int s = 56; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); }
    }
    void text(const std::string &s) { head(3, s.size()); out.insert(out.end(), s.begin(), s.end()); }
    void null() { out.push_back(0xf6); }
    // ciborium writes the narrowest float that round-trips (f16, else f32)
    void f32(float f) {
        uint32_t u; std::memcpy(&u, &f, 4);
        const uint32_t sign = u >> 31, exp = (u >> 23) & 0xff, man = u & 0x7fffff;
        bool half_ok = false; uint16_t h = 0;
        if (exp == 0xff) { half_ok = (man & 0x1fff) == 0; h = (uint16_t)((sign << 15) | 0x7c00 | (man >> 13)); if (man && !(man >> 13)) half_ok = false; }
        else if (exp == 0 && man == 0) { half_ok = true; h = (uint16_t)(sign << 15); }
        else {
            const int e = (int)exp - 127;
            if (e >= -14 && e <= 15 && (man & 0x1fff) == 0) { half_ok = true; h = (uint16_t)((sign << 15) | ((uint32_t)(e + 15) << 10) | (man >> 13)); }
            else if (e >= -24 && e < -14) {  // f16 subnormal
                const int shift = -14 - e;  // 1..10
                const uint32_t full = man | 0x800000;
                if ((full & ((1u << (13 + shift)) - 1)) == 0) { half_ok = true; h = (uint16_t)((sign << 15) | (full >> (13 + shift))); }
            }
        }
        if (half_ok) { out.push_back(0xf9); out.push_back((uint8_t)(h >> 8)); out.push_back((uint8_t)h); }
        else { out.push_back(0xfa); for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(u >> s)); }
    }
    void matrix(const std::vector<float> &m, int rows, int K) {
        head(4, (uint64_t)rows);
        for (int r = 0; r < rows; ++r) { head(4, (uint64_t)K); for (int k = 0; k < K; ++k) f32(m[(size_t)r * K + k]); }
    }
};

}  // namespace

// MfccWavFileExtractor::compute_mfccs, src/mfcc/wav_file_extractor.rs:18-69: whole-matrix-normalised MFCCs
// [frames][K] of one wav buffer + the median chunk RMS.
bool compute_wav_mfccs(Ctx *ctx, const uint8_t *buf, size_t len, int K, std::vector<float> *mfcc, int *frames, float *rms_level) {
    Wav w; std::string err;
    if (!parse_wav(buf, len, &w, &err)) { set_last_error(err); return false; }
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return false;
    // encode_samples, :71-96: chunks_exact(input frame) -> resample -> RMS of every encoded buffer -> concatenate
    size_t enc_chunk = 480;
    if (w.rate != 16000) {
        const Resampler *rs = ctx->resampler_for(w.rate);
        if (!rs) return false;
        const size_t fi = (size_t)rs->dev.fi, fo = (size_t)rs->dev.fo, nch = w.mono.size() / fi;
        std::vector<float> enc(nch * fo);
        if (nch) {
            DevBuf din, dxs, dout;
            if (!din.reserve(nch * fi * 4) || !dxs.reserve((1 + nch) * fi * 4 + 64) || !dout.reserve(nch * fo * 4)) return false;
            if (!hip_ok(hipMemcpyAsync(din.p, w.mono.data(), nch * fi * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync") ||
                !hip_ok(launch_resample_stage(ctx->stream, din.p, 3, 1, 1, nch, (int)fi, nch * fi, nullptr, dxs.as<float>()), "resample_stage_kernel") ||
                !hip_ok(launch_resample(ctx->stream, rs->dev, dxs.as<float>(), 1, nch, dout.as<float>(), nch * fo), "resample_mfma_kernel") ||
                !hip_ok(hipMemcpyAsync(enc.data(), dout.p, nch * fo * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync") ||
                !hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize"))
                return false;
        }
        w.mono.swap(enc);
        enc_chunk = fo;
    }
    std::vector<float> rms;
    for (size_t c = 0; c + enc_chunk <= (w.mono.size() / enc_chunk) * enc_chunk; c += enc_chunk) rms.push_back(rms_level_of(&w.mono[c], enc_chunk));
    if (!rms.empty()) { std::sort(rms.begin(), rms.end()); *rms_level = rms[rms.size() / 2]; }  // :54-58
    const size_t n = (w.mono.size() / 480) * 480;  // chunks_exact(output frame): a tail shorter than 30 ms is dropped
    const size_t nf = n >= 480 ? 3 * (n / 480) - 3 : 0;
    *frames = (int)nf;
    mfcc->assign(nf * K, 0.f);
    if (nf == 0) return true;
    const MfccTablesDev *tb = ctx->tables_for(K);
    if (!tb) return false;
    DevBuf dp, dm;
    if (!dp.reserve(n * 4) || !dm.reserve(nf * K * 4)) return false;
    if (!hip_ok(hipMemcpyAsync(dp.p, w.mono.data(), n * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync")) return false;
    if (!hip_ok(launch_mfcc(ctx->stream, *tb, dp.as<float>(), 1, n, n, 0, nf, nf, dm.as<float>()), "mfcc_kernel")) return false;
    std::vector<float> raw(nf * K);
    if (!hip_ok(hipMemcpyAsync(raw.data(), dm.p, nf * K * 4, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync")) return false;
    if (!hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) return false;
    // MfccNormalizer::normalize over the whole matrix (:67), sequential column sums
    std::vector<float> sum((size_t)K, 0.f);
    for (size_t i = 0; i < nf; ++i) for (int j = 0; j < K; ++j) sum[j] += raw[i * K + j];
    for (size_t i = 0; i < nf; ++i) for (int j = 0; j < K; ++j) (*mfcc)[i * K + j] = raw[i * K + j] - sum[j] / (float)nf;
    return true;
}

// compute_avg_samples_features, src/wakewords/comp/wakeword_ref_build.rs:90-110
static bool average_templates(const WakewordRefData &r, std::vector<float> *avg, int *avg_len) {
    const size_t T = r.tnames.size();
    if (T <= 1) return false;
    std::vector<size_t> order(T);
    for (size_t i = 0; i < T; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (r.lens[a] != r.lens[b]) return r.lens[a] > r.lens[b];
        return r.tnames[a] < r.tnames[b];
    });
    *avg = r.feats[order[0]];
    *avg_len = r.lens[order[0]];
    for (size_t i = 1; i < T; ++i) average_step(*avg, *avg_len, r.feats[order[i]], r.lens[order[i]], r.mfcc_size);
    return true;
}

// WakewordRef::new_from_sample_buffers / _files; rms_median: files take the median sample level (:80-81),
// buffers the maximum (:24-26).
bool build_wakeword_ref(Ctx *ctx, const std::string &name, const float *threshold, const float *avg_threshold, size_t n,
                        const char *const *sample_names, const uint8_t *const *wavs, const size_t *wav_lens, int mfcc_size,
                        bool rms_median, WakewordRefData *out) {
    if (mfcc_size < 1) { set_last_error("mfcc_size must be >= 1"); return false; }
    WakewordRefData r;
    r.name = name;
    r.mfcc_size = mfcc_size;
    r.has_threshold = threshold != nullptr; r.threshold = threshold ? *threshold : 0.f;
    r.has_avg_threshold = avg_threshold != nullptr; r.avg_threshold = avg_threshold ? *avg_threshold : 0.f;
    std::vector<float> levels;
    for (size_t i = 0; i < n; ++i) {
        std::vector<float> m; int frames = 0; float level = 0.f;
        if (!compute_wav_mfccs(ctx, wavs[i], wav_lens[i], mfcc_size, &m, &frames, &level)) return false;
        if (frames == 0) { set_last_error(std::string("sample too short: ") + sample_names[i]); return false; }
        // HashMap::insert: a repeated name replaces the earlier sample
        auto it = std::find(r.tnames.begin(), r.tnames.end(), sample_names[i]);
        if (it != r.tnames.end()) { size_t k = (size_t)(it - r.tnames.begin()); r.feats[k] = std::move(m); r.lens[k] = frames; levels[k] = level; }
        else { r.tnames.push_back(sample_names[i]); r.feats.push_back(std::move(m)); r.lens.push_back(frames); levels.push_back(level); }
    }
    if (r.tnames.empty()) { set_last_error("Can not create an empty wakeword"); return false; }  // wakeword_ref.rs:52-54
    if (rms_median) { std::vector<float> s(levels); std::sort(s.begin(), s.end()); r.rms_level = s[s.size() / 2]; }
    else { float mx = 0.f; for (float v : levels) if (v > mx) mx = v; r.rms_level = mx; }
    r.has_avg = average_templates(r, &r.avg, &r.avg_len);
    *out = std::move(r);
    return true;
}

// WakewordSave::save_to_buffer, src/wakewords/wakeword_file.rs:22-26 (struct field order of wakeword_ref.rs:12-20)
std::vector<uint8_t> serialize_wakeword_ref(const WakewordRefData &r) {
    Cbor c;
    c.head(5, 7);
    c.text("name"); c.text(r.name);
    c.text("avg_features"); if (r.has_avg) c.matrix(r.avg, r.avg_len, r.mfcc_size); else c.null();
    c.text("samples_features"); c.head(5, r.tnames.size());
    for (size_t t = 0; t < r.tnames.size(); ++t) { c.text(r.tnames[t]); c.matrix(r.feats[t], r.lens[t], r.mfcc_size); }
    c.text("threshold"); if (r.has_threshold) c.f32(r.threshold); else c.null();
    c.text("avg_threshold"); if (r.has_avg_threshold) c.f32(r.avg_threshold); else c.null();
    c.text("rms_level"); c.f32(r.rms_level);
    c.text("mfcc_size"); c.head(0, (uint64_t)r.mfcc_size);
    return c.out;
}

// WakewordModel through WakewordSave::save_to_buffer (wakeword_model.rs:11-18, TensorData :68-72): the weight
// bytes are written as a CBOR ARRAY of small integers (serde's Vec<u8>), little-endian f32
std::vector<uint8_t> serialize_wakeword_model(const WakewordModelData &m) {
    Cbor c;
    c.head(5, 6);
    c.text("labels"); c.head(4, m.labels.size()); for (const std::string &l : m.labels) c.text(l);
    c.text("train_size"); c.head(0, (uint64_t)m.train_size);
    c.text("mfcc_size"); c.head(0, (uint64_t)m.mfcc_size);
    c.text("m_type"); c.text(m.m_type);
    c.text("weights"); c.head(5, m.weights.size());
    for (const auto &kv : m.weights) {
        c.text(kv.first);
        c.head(5, 3);
        const std::vector<float> &w = kv.second.second;
        c.text("bytes"); c.head(4, w.size() * 4);
        for (float f : w) { uint8_t b[4]; std::memcpy(b, &f, 4); for (int i = 0; i < 4; ++i) c.head(0, b[i]); }
        c.text("dims"); c.head(4, kv.second.first.size()); for (size_t d : kv.second.first) c.head(0, (uint64_t)d);
        c.text("d_type"); c.text("f32");
    }
    c.text("rms_level"); c.f32(m.rms_level);
    return c.out;
}

}  // namespace rp
