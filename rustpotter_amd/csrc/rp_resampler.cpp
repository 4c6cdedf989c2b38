// rp_resampler.cpp -- plan for the sample-rate converter in front of the detector.
//
// The reference resamples any input that is not 16 kHz with rubato 0.14.1's FftFixedInOut<f32>
// (src/audio/encoder.rs:72-83): per input frame, zero-pad to 2*fft_size_in, real FFT, multiply by the
// spectrum of a BlackmanHarris^2 windowed sinc, truncate to the output band, inverse real FFT of
// 2*fft_size_out, overlap-add.  Every step is linear and the same for every frame, so one output frame
// is a fixed linear map of the previous and the current input frame:
//     out[c*fo + j] = sum_{n < 2*fi} x[(c-1)*fi + n] * G[j][n]
// with G[j][n] = g((j*fi - (n - fi)*fo) / gcd(fi,fo) mod N), g the inverse DFT of the filter spectrum on
// the common grid of N = 2*fi*fo/gcd points.  The plan evaluates g in f64 from the f32 filter spectrum
// (the quantities rubato itself holds in f32) and stores G for the device kernel (resample_mfma_kernel).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "rp_host.h"

namespace rp {

static const float kPiF = 3.14159274101257324f;

// Anti-aliasing cutoff of the windowed sinc relative to the narrower Nyquist band: rubato's
// calculate_cutoff::<f32>(npoints, BlackmanHarris2) = 1 / (k1/n + k2/n^2 + k3/n^3 + 1), evaluated in f32 like the crate
// does.  The constants are the crate's published ones; for n = 480 (48 kHz -> 16 kHz) the value, 0.9716114, was
// confirmed independently by fitting the cutoff against the reference's 48 kHz goldens (DESIGN.md "Resampler").
static float resampler_cutoff(int npoints) {
    const float k1 = 13.745202940783823f, k2 = 121.73532586374934f, k3 = 5964.163279612051f;
    const float n = (float)npoints;
    return 1.0f / (k1 / n + k2 / (n * n) + k3 / (n * n * n) + 1.0f);
}

// FftFixedInOut::new(fs_in, 16000, 480, 1): the wanted chunk is divided by the OUTPUT granule
bool resampler_frame_lengths(size_t fs_in, size_t *in_len, size_t *out_len) {
    const size_t fs_out = 16000;
    if (fs_in == 0) return false;  // validate_sample_rates
    if (fs_in == fs_out) { *in_len = *out_len = 480; return true; }
    const size_t g = std::gcd(fs_in, fs_out);
    const size_t chunks = (size_t)std::ceil(480.0f / (float)(fs_out / g));
    *out_len = chunks * fs_out / g;
    *in_len = chunks * fs_in / g;
    return true;
}

Resampler *Resampler::create(Ctx *ctx, size_t fs_in) {
    size_t fi_, fo_;
    if (!resampler_frame_lengths(fs_in, &fi_, &fo_) || fs_in == 16000) {
        set_last_error("Unsupported sample rate, unable to initialize the resampler");
        return nullptr;
    }
    const int fi = (int)fi_, fo = (int)fo_;
    // what the device kernel is built for: output frames of 480 or 640 samples (every standard audio rate)
    // and a matrix that stays small; anything else has no kernel
    const size_t kpad = ((size_t)2 * fi + 15) / 16 * 16;
    if ((fo != 480 && fo != 640) || kpad * (size_t)fo * sizeof(float) > ((size_t)256 << 20)) {
        set_last_error("Unsupported sample rate, unable to initialize the resampler (no device kernel for this frame size)");
        return nullptr;
    }
    // ---- windows.rs blackman_harris (periodic) squared, sinc.rs make_sincs(fi, 1, cutoff), all in f32
    const float cutoff = fi > fo ? resampler_cutoff(fo) * (float)fo / (float)fi : resampler_cutoff(fi);
    std::vector<float> taps((size_t)fi);
    {
        const float pi2 = 2.0f * kPiF, pi4 = 4.0f * kPiF, pi6 = 6.0f * kPiF, np_f = (float)fi;
        float sum = 0.f;
        for (int x = 0; x < fi; ++x) {
            const float xf = (float)x;
            float w = 0.35875f - 0.48829f * std::cos(pi2 * xf / np_f) + 0.14128f * std::cos(pi4 * xf / np_f) -
                      0.01168f * std::cos(pi6 * xf / np_f);
            w = w * w;
            const float v = (xf - (float)(fi / 2)) * cutoff;
            const float s = v == 0.f ? 1.f : std::sin(v * kPiF) / (v * kPiF);
            taps[x] = w * s;
            sum += taps[x];
        }
        for (int x = 0; x < fi; ++x) taps[x] = (taps[x] / sum) / (float)(2 * fi);  // FftResampler::new: / (2*fft_size_in)
    }
    // ---- filter spectrum, bins the unit keeps (new_len), rounded to f32 like realfft's Complex<f32>
    const int nl = fi < fo ? fi + 1 : fo;
    const int Ni = 2 * fi;
    std::vector<double> cs((size_t)Ni), sn((size_t)Ni);
    for (int j = 0; j < Ni; ++j) { const double th = 2.0 * M_PI * (double)j / (double)Ni; cs[j] = std::cos(th); sn[j] = std::sin(th); }
    std::vector<float> hr((size_t)nl), hi((size_t)nl);
    for (int k = 0; k < nl; ++k) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < fi; ++n) { const int j = (int)(((long long)k * n) % Ni); sr += (double)taps[n] * cs[j]; si -= (double)taps[n] * sn[j]; }
        hr[k] = (float)sr; hi[k] = (float)si;
    }
    // ---- impulse response of "filter, truncate, inverse transform" on the common time grid
    const long long g = std::gcd(fi, fo);
    const long long N = 2LL * fi * fo / g;
    std::vector<double> cN((size_t)N), sN((size_t)N);
    for (long long j = 0; j < N; ++j) { const double th = 2.0 * M_PI * (double)j / (double)N; cN[j] = std::cos(th); sN[j] = std::sin(th); }
    // only the grid points the matrix touches are evaluated
    std::vector<float> gt((size_t)N, 0.f);
    std::vector<unsigned char> have((size_t)N, 0);
    std::vector<float> G((size_t)fo * kpad, 0.f);
    for (int j = 0; j < fo; ++j)
        for (int n = 0; n < 2 * fi; ++n) {
            long long t = ((long long)j * fi - (long long)(n - fi) * fo) / g;  // exact: both terms are multiples of g
            t %= N; if (t < 0) t += N;
            if (!have[t]) {
                double acc = (double)hr[0];
                for (int k = 1; k < nl; ++k) {
                    const long long idx = ((long long)k * t) % N;
                    acc += 2.0 * ((double)hr[k] * cN[idx] - (double)hi[k] * sN[idx]);
                }
                gt[t] = (float)acc; have[t] = 1;
            }
            G[(size_t)j * kpad + n] = gt[t];
        }
    std::unique_ptr<Resampler> r(new Resampler());
    r->ctx = ctx;
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    if (!r->g2t.reserve(G.size() * sizeof(float))) return nullptr;
    if (!hip_ok(hipMemcpy(r->g2t.p, G.data(), G.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(resampler matrix)")) return nullptr;
    r->dev.fs_in = (int)fs_in; r->dev.fi = fi; r->dev.fo = fo; r->dev.kpad = (int)kpad; r->dev.g2t = r->g2t.as<float>();
    // 48 kHz: tables of the FFT-structured kernel (twiddles from f64, the filter spectrum halved: the kernel's
    // untangling step leaves 2*U).  RP_RESAMPLE_GEMM=1 keeps the matrix kernel (benchmarks / cross-checks).
    const char *force = std::getenv("RP_RESAMPLE_GEMM");
    if (fi == 1440 && fo == 480 && !(force && force[0] == '1')) {
        std::vector<float2> t((size_t)kR48TableLen, make_float2(0.f, 0.f));
        auto w = [](long long e, long long n) { const double th = -2.0 * M_PI * (double)(e % n) / (double)n; return make_float2((float)std::cos(th), (float)std::sin(th)); };
        for (int k = 0; k < 240; ++k) { t[kR48OffTw240 + k] = w(k, 240); t[kR48OffTw480 + k] = w(k, 480); }
        for (int d = 0; d < 6; ++d) for (int q = 0; q < 480; ++q) t[kR48OffTwc + d * 480 + q] = w((long long)d * q, 2880);
        for (int k = 0; k <= 240; ++k) { float2 c = w(k, 960); c.y = -c.y; t[kR48OffW960c + k] = c; }
        for (int k = 0; k < 480; ++k) t[kR48OffHf + k] = make_float2(0.5f * hr[k], 0.5f * hi[k]);
        if (!r->fft48.reserve(t.size() * sizeof(float2))) return nullptr;
        if (!hip_ok(hipMemcpy(r->fft48.p, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice), "hipMemcpy(resampler tables)")) return nullptr;
        r->dev.fft48 = r->fft48.as<float>();
    }
    return r.release();
}

}  // namespace rp
