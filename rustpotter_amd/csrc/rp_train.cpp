// rp_train.cpp -- WakewordModelTrain::train_from_buffers (src/wakewords/nn/wakeword_model_train.rs:44-168) on the
// device: MFCC features of labelled wav samples (MFCC kernel, resampler for wavs that are not 16 kHz), the
// reference's MLP shapes (wakeword_nn.rs:305-389), full-batch log-softmax / nll / SGD epochs, and the result as a
// WakewordModel .rpw (wakeword_model.rs:11-18).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rp_host.h"

namespace rp {

namespace {
// label = lower-cased text between the first '[' and the first ']' of the file name, "none" without one (:283-295)
std::string label_of(const std::string &name) {
    // the reference works on chars; the names it is used with are ASCII around the brackets
    const size_t a = name.find('['), b = name.find(']');
    if (a == std::string::npos || b == std::string::npos || !(a < b)) return "none";
    std::string l = name.substr(a + 1, b - a - 1);
    for (char &c : l) c = (char)std::tolower((unsigned char)c);
    return l;
}

struct Rng {  // seeded generator for a fresh model's weights (the reference draws from the thread RNG: never reproducible)
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return ((double)(next() >> 11) + 0.5) / 9007199254740992.0; }  // (0,1)
    double normal() { const double u = uni(), v = uni(); return std::sqrt(-2.0 * std::log(u)) * std::cos(2.0 * M_PI * v); }
};

struct Sample { std::vector<float> feat; int label; };

// get_mfccs_labeled, :274-325
bool labelled_features(Ctx *ctx, size_t n, const char *const *names, const uint8_t *const *wavs, const size_t *lens, int K,
                       std::vector<std::string> *labels, bool new_labels, float *rms_level, std::vector<Sample> *out) {
    for (size_t i = 0; i < n; ++i) {
        const std::string label = label_of(names[i]);
        auto it = std::find(labels->begin(), labels->end(), label);
        if (it == labels->end()) {
            if (!new_labels) {
                set_last_error("Forbidden label '" + label + "', it doesn't exists on the training data or in the model you are training from.");
                return false;
            }
            labels->push_back(label);
            it = labels->end() - 1;
        }
        Sample s;
        s.label = (int)(it - labels->begin());
        int frames = 0; float level = 0.f;
        if (!compute_wav_mfccs(ctx, wavs[i], lens[i], K, &s.feat, &frames, &level)) return false;
        if (rms_level && label != "none") *rms_level = std::isnan(*rms_level) ? level : (*rms_level + level) / 2.f;
        out->push_back(std::move(s));
    }
    return true;
}
}  // namespace

// the layer widths of Tiny / Small / Medium / Large, wakeword_nn.rs:305-389 (MFCCS_EXTRACTOR_OUT_SHIFTS = 3)
bool model_dims(int m_type, size_t input_len, int mfcc_size, size_t n_labels, std::vector<int> *dims) {
    const size_t fr = input_len / (size_t)mfcc_size;
    dims->clear();
    dims->push_back((int)input_len);
    switch (m_type) {
    case 0: dims->push_back((int)(fr / 15)); break;
    case 1: dims->push_back((int)(fr / 6)); dims->push_back((int)((fr / 6) / 2)); break;
    case 2: dims->push_back((int)(fr / 3)); dims->push_back((int)(fr / 6)); break;
    case 3: dims->push_back((int)((fr / 3) * 2)); dims->push_back((int)(fr / 6)); break;
    default: set_last_error("unknown model type"); return false;
    }
    dims->push_back((int)n_labels);
    for (size_t i = 1; i + 1 < dims->size(); ++i)
        if ((*dims)[i] < 1) { set_last_error("training samples too short for this model type"); return false; }
    return true;
}

static const char *kTypeNames[4] = {"Tiny", "Small", "Medium", "Large"};

bool train_wakeword_model(Ctx *ctx, const rp_train_options &opt, size_t n_train, const char *const *train_names,
                          const uint8_t *const *train_wavs, const size_t *train_lens, size_t n_test, const char *const *test_names,
                          const uint8_t *const *test_wavs, const size_t *test_lens, const WakewordModelData *prev,
                          WakewordModelData *out, float *final_loss, float *test_accuracy) {
    if (n_train == 0) { set_last_error("No training data provided"); return false; }
    if (n_test == 0) { set_last_error("No test data provided"); return false; }
    std::vector<std::string> labels;
    int m_type = opt.m_type, K = (int)opt.mfcc_size;
    if (prev) {  // "Training from previous model, some options will be ignored."
        labels = prev->labels;
        K = prev->mfcc_size;
        m_type = -1;
        for (int t = 0; t < 4; ++t) if (prev->m_type == kTypeNames[t]) m_type = t;
        if (m_type < 0) { set_last_error("unknown model type in the model to train from"); return false; }
    }
    if (K < 1) { set_last_error("mfcc_size must be >= 1"); return false; }
    float rms_level = NAN;
    std::vector<Sample> train, test;
    if (!labelled_features(ctx, n_train, train_names, train_wavs, train_lens, K, &labels, prev == nullptr, &rms_level, &train)) return false;
    if (!labelled_features(ctx, n_test, test_names, test_wavs, test_lens, K, &labels, false, nullptr, &test)) return false;
    if (labels.size() < 2) { set_last_error("Your training data need to contain at least two labels"); return false; }
    size_t input_len = 0;
    if (prev) input_len = prev->train_size * (size_t)K;
    else for (const Sample &s : train) input_len = std::max(input_len, s.feat.size());
    if (input_len == 0) { set_last_error("training samples too short"); return false; }
    std::vector<int> dims;
    if (!model_dims(m_type, input_len, K, labels.size(), &dims)) return false;
    const int nl = (int)dims.size() - 1;
    // weights: the previous model's, or candle_nn::linear's initialisation (weights N(0, sqrt(2/fan_in)), biases
    // U(-1/sqrt(fan_in), 1/sqrt(fan_in))) from a seeded generator
    std::vector<std::vector<float>> W((size_t)nl), Bv((size_t)nl);
    Rng rng{opt.seed};
    for (int l = 0; l < nl; ++l) {
        const size_t in = (size_t)dims[l], on = (size_t)dims[l + 1];
        const std::string wn = "ln" + std::to_string(l + 1) + ".weight", bn = "ln" + std::to_string(l + 1) + ".bias";
        if (prev) {
            auto wi = prev->weights.find(wn), bi = prev->weights.find(bn);
            if (wi == prev->weights.end() || bi == prev->weights.end() || wi->second.second.size() != in * on || bi->second.second.size() != on) {
                set_last_error("Incorrect model layers");
                return false;
            }
            W[l] = wi->second.second; Bv[l] = bi->second.second;
        } else {
            W[l].resize(in * on); Bv[l].resize(on);
            const double std_w = std::sqrt(2.0) / std::sqrt((double)in), bound = 1.0 / std::sqrt((double)in);
            for (float &v : W[l]) v = (float)(rng.normal() * std_w);
            for (float &v : Bv[l]) v = (float)((rng.uni() * 2.0 - 1.0) * bound);
        }
    }
    // pad / truncate to input_len (:113-116), upload
    auto pack = [&](std::vector<Sample> &set, std::vector<float> *x, std::vector<int32_t> *y) {
        x->assign(set.size() * input_len, 0.f); y->resize(set.size());
        for (size_t i = 0; i < set.size(); ++i) {
            std::memcpy(x->data() + i * input_len, set[i].feat.data(), std::min(set[i].feat.size(), input_len) * sizeof(float));
            (*y)[i] = set[i].label;
        }
    };
    std::vector<float> xtr, xte; std::vector<int32_t> ytr, yte;
    pack(train, &xtr, &ytr); pack(test, &xte, &yte);
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return false;
    hipStream_t st = ctx->stream;
    const size_t B = train.size(), Bt = test.size(), Bmax = std::max(B, Bt);
    DevBuf dx, dxt, dy, dloss;
    if (!dx.reserve(xtr.size() * 4) || !dxt.reserve(xte.size() * 4) || !dy.reserve(B * 4) || !dloss.reserve(B * 4)) return false;
    std::vector<DevBuf> dW((size_t)nl), dB((size_t)nl), dact((size_t)nl), ddz((size_t)nl);
    std::vector<float *> pW, pB, pact, pdz;
    for (int l = 0; l < nl; ++l) {
        if (!dW[l].reserve(W[l].size() * 4) || !dB[l].reserve(Bv[l].size() * 4) || !dact[l].reserve(Bmax * (size_t)dims[l + 1] * 4) ||
            !ddz[l].reserve(Bmax * (size_t)dims[l + 1] * 4)) return false;
        if (!hip_ok(hipMemcpyAsync(dW[l].p, W[l].data(), W[l].size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync") ||
            !hip_ok(hipMemcpyAsync(dB[l].p, Bv[l].data(), Bv[l].size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync")) return false;
        pW.push_back(dW[l].as<float>()); pB.push_back(dB[l].as<float>()); pact.push_back(dact[l].as<float>()); pdz.push_back(ddz[l].as<float>());
    }
    if (!hip_ok(hipMemcpyAsync(dx.p, xtr.data(), xtr.size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync") ||
        !hip_ok(hipMemcpyAsync(dxt.p, xte.data(), xte.size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync") ||
        !hip_ok(hipMemcpyAsync(dy.p, ytr.data(), B * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync")) return false;
    // training_loop :170-222
    for (size_t epoch = 1; epoch <= opt.epochs; ++epoch)
        if (!hip_ok(launch_train_step(st, dx.as<float>(), dy.as<int32_t>(), B, nl, dims.data(), pW.data(), pB.data(), pact.data(), pdz.data(),
                                      opt.learning_rate, dloss.as<float>()), "training step")) return false;
    // last epoch's loss (computed before its update, like candle's loss value) and the test accuracy of the final weights
    std::vector<float> loss_rows(B, 0.f), tlog(Bt * labels.size());
    if (opt.epochs && !hip_ok(hipMemcpyAsync(loss_rows.data(), dloss.p, B * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync")) return false;
    if (!hip_ok(launch_train_forward(st, dxt.as<float>(), Bt, nl, dims.data(), pW.data(), pB.data(), pact.data()), "test forward")) return false;
    if (!hip_ok(hipMemcpyAsync(tlog.data(), pact[nl - 1], tlog.size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync")) return false;
    for (int l = 0; l < nl; ++l)
        if (!hip_ok(hipMemcpyAsync(W[l].data(), dW[l].p, W[l].size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync") ||
            !hip_ok(hipMemcpyAsync(Bv[l].data(), dB[l].p, Bv[l].size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync")) return false;
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return false;
    if (final_loss) { float s = 0.f; for (float v : loss_rows) s += v; *final_loss = opt.epochs ? s / (float)B : NAN; }
    if (test_accuracy) {  // test_model :251-272: argmax (first maximum) == label
        size_t ok = 0;
        const size_t C = labels.size();
        for (size_t b = 0; b < Bt; ++b) {
            size_t best = 0;
            for (size_t c = 1; c < C; ++c) if (tlog[b * C + c] > tlog[b * C + best]) best = c;
            ok += (int)best == yte[b] ? 1 : 0;
        }
        *test_accuracy = (float)ok / (float)Bt;
    }
    WakewordModelData m;
    m.labels = labels; m.train_size = input_len / (size_t)K; m.mfcc_size = K; m.m_type = kTypeNames[m_type]; m.rms_level = rms_level;
    for (int l = 0; l < nl; ++l) {
        m.weights["ln" + std::to_string(l + 1) + ".weight"] = {{(size_t)dims[l + 1], (size_t)dims[l]}, W[l]};
        m.weights["ln" + std::to_string(l + 1) + ".bias"] = {{(size_t)dims[l + 1]}, Bv[l]};
    }
    *out = std::move(m);
    return true;
}

}  // namespace rp
