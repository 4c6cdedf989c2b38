// rp_tables.cpp -- constant tables of the MFCC pipeline, computed on the host in
// strict f32 in the reference's evaluation order, then uploaded once per mfcc_size.
#include <cmath>

#include "rp_host.h"

namespace rp {

namespace {
constexpr float kPi = 3.14159274101257324f;  // std::f32::consts::PI
constexpr int kSampleRate = 16000;           // src/constants.rs:1

// src/mfcc/extractor.rs:132-134
float frequency_to_mel(int frequency) { return 1127.f * std::log(1.f + ((float)frequency / 700.0f)); }
}  // namespace

HostTables build_tables(int K) {
    HostTables t;
    const int K1 = K + 1;  // set_out_size: num_coefficients = out_size + 1, extractor.rs:48
    t.K1 = K1;
    // new_hamming_window, extractor.rs:115-120
    t.hamming.resize(kFrame);
    for (int s = 0; s < kFrame; ++s)
        t.hamming[s] = 0.54f - (0.46f * std::cos(2.f * kPi * ((float)s / (float)(kFrame - 1))));
    // new_mel_filter_bank, extractor.rs:164-198 (min_frequency 0, max_frequency sample_rate/2)
    const float max_mel = std::floor(frequency_to_mel(kSampleRate / 2));
    const float min_mel = std::floor(frequency_to_mel(0));
    t.centres.resize(K1 + 2);
    for (int i = 0; i < K1 + 2; ++i) {
        float f = (float)i * (max_mel - min_mel) / (float)(K1 + 1) + min_mel;
        float tmp = std::log(1.f + 1000.0f / 700.0f) / 1000.0f;
        tmp = (std::exp(f * tmp) - 1.f) / ((float)kSampleRate / 2.f);
        t.centres[i] = (int)std::floor(0.5f + 700.f * (float)kBins * tmp);
    }
    t.fb.assign((size_t)K1 * kBins, 0.f);
    for (int i = 0; i < K1; ++i) {
        const int b = t.centres[i], c = t.centres[i + 1], e = t.centres[i + 2];
        for (int k = b; k < c && k < kBins; ++k) t.fb[(size_t)i * kBins + k] = (float)(k - b) / (float)(c - b);
        for (int k = c; k < e && k < kBins; ++k) t.fb[(size_t)i * kBins + k] = (float)(e - k) / (float)(e - c);
    }
    // discrete_cosine_transform argument, extractor.rs:146-163: cos(pi_over_n * (n + 0.5) * k) in f32
    t.dct.resize((size_t)K1 * K1);
    const float pi_over_n = kPi / (float)K1;
    for (int k = 0; k < K1; ++k)
        for (int n = 0; n < K1; ++n) t.dct[(size_t)k * K1 + n] = std::cos(pi_over_n * ((float)n + 0.5f) * (float)k);
    // DFT twiddles (rustfft 6.1.0 computes its twiddles in f64 and stores f32)
    t.tw240.resize(240);
    t.tw480.resize(240);
    for (int k = 0; k < 240; ++k) {
        double a = -2.0 * 3.14159265358979323846 * (double)k / 240.0;
        double b = -2.0 * 3.14159265358979323846 * (double)k / 480.0;
        t.tw240[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        t.tw480[k] = make_float2((float)std::cos(b), (float)std::sin(b));
    }
    return t;
}

}  // namespace rp
