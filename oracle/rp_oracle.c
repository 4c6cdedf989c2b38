/*
 * rp_oracle.c -- CPU restatement of rustpotter v3.0.2's MFCC + DTW scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rustpotter_amd/ may include, link or
 * call this file.  It is used by tests/, by __graft_entry__.smoke() and by the
 * `cpu_baseline` leg of bench.py, always as the checker / the CPU number that is
 * reported next to the GPU number, never as the product path.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *   G1  the MFCC matrices the reference itself wrote into
 *       tests/resources/{oye_casa_g,alexa}.rpw (tests/wakeword.rs:27-54),
 *   G2  the exact f32 (avg_score, score) pairs asserted in tests/detector.rs:9-87
 *       and the "0 detections" case tests/detector.rs:89-99,
 *   G2b the filter goldens tests/detector.rs:113-159,
 *   G5  the NN score formula value tests/detector.rs:227,
 *   G6  the 48 kHz path: the MFCC matrices of six resampled recordings inside
 *       tests/resources/oye_casa_real.rpw (tests/wakeword.rs:57-71), the
 *       (avg_score, score, counter) triples of tests/detector.rs:163-213, and the three
 *       example wavs the filter tests write from real_sample.wav (band_pass_filter.rs:69-185,
 *       gain_normalizer_filter.rs:81-131): the resampler's output sample by sample,
 *   G7  oye_casa_g_1_f32.wav (encoder.rs:139-183): i16 -> f32 decode, bit-exact.
 * The reference (Rust) cannot be built in this image (no cargo/rustc, crates not
 * vendored).  Third-party arithmetic not present under /root/reference:
 *   rustfft 6.1.0 (Cargo.lock:545): forward unnormalised complex DFT, restated
 *   here as a plain mixed-radix f32 FFT (twiddles computed in f64, stored f32);
 *   rubato 0.14.1 (Cargo.lock:533): FftFixedInOut<f32>, restated from its published
 *   algorithm (its anti-aliasing cutoff formula cross-checked numerically on G6, see
 *   rubato_cutoff).  The wakeword-model goldens on resampled audio
 *   (tests/detector.rs:216-267) are NOT reproducible by any arithmetic other than
 *   rustfft's own (tests/test_oracle_golden.py explains): the NN forward stays
 *   "parity unpinned" by reference goldens and is pinned to this file instead.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference root).  All arithmetic is strict f32, evaluated in the reference's
 * order; build with -ffp-contract=off (see oracle/Makefile).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#define ORC_SAMPLE_RATE 16000 /* src/constants.rs:1 */
#define ORC_FRAME 480         /* 30 ms @16k, src/constants.rs:2, src/detector.rs:101 */
#define ORC_SHIFT 160         /* 10 ms, src/constants.rs:8, src/detector.rs:102-104 */
#define ORC_NBINS 240         /* samples_per_frame / 2, src/mfcc/extractor.rs:28 */
#define ORC_PRE_EMPHASIS 0.97f /* src/constants.rs:9 */
#define ORC_PI 3.14159274101257324f /* std::f32::consts::PI */
#define ORC_MAX_K1 65          /* K+1 upper bound used for stack buffers */

/* ------------------------------------------------------------------ tables */

/* src/mfcc/extractor.rs:115-120 */
void orc_hamming_window(int n, float *out) {
    int ns_minus_1 = n - 1;
    for (int s = 0; s < n; ++s)
        out[s] = 0.54f - (0.46f * cosf(2.f * ORC_PI * ((float)s / (float)ns_minus_1)));
}

/* src/mfcc/extractor.rs:132-134 */
static float frequency_to_mel(int frequency) {
    return 1127.f * logf(1.f + ((float)frequency / 700.0f));
}

/* src/mfcc/extractor.rs:164-198.  out is [ncoef][nbins] row-major, centres is
 * [ncoef+2] (may be NULL). */
void orc_mel_filter_bank(int sample_rate, int nbins, int ncoef, float *out, int *centres_out) {
    float max_mel = floorf(frequency_to_mel(sample_rate / 2));
    float min_mel = floorf(frequency_to_mel(0));
    int *centre = (int *)malloc(sizeof(int) * (size_t)(ncoef + 2));
    memset(out, 0, sizeof(float) * (size_t)ncoef * (size_t)nbins);
    for (int i = 0; i < ncoef + 2; ++i) {
        float f = (float)i * (max_mel - min_mel) / (float)(ncoef + 1) + min_mel;
        float tmp = logf(1.f + 1000.0f / 700.0f) / 1000.0f;
        tmp = (expf(f * tmp) - 1.f) / ((float)sample_rate / 2.f);
        centre[i] = (int)floorf(0.5f + 700.f * (float)nbins * tmp);
        if (centres_out) centres_out[i] = centre[i];
    }
    for (int i = 0; i < ncoef; ++i) {
        int b = centre[i], c = centre[i + 1], e = centre[i + 2];
        int up = c - b, down = e - c;
        for (int k = b; k < c && k < nbins; ++k) out[i * nbins + k] = (float)(k - b) / (float)up;
        for (int k = c; k < e && k < nbins; ++k) out[i * nbins + k] = (float)(e - k) / (float)down;
    }
    free(centre);
}

/* ------------------------------------------------------------------ FFT-480
 * Stands in for rustfft 6.1.0 `plan_fft_forward(480)` (src/mfcc/extractor.rs:102-110):
 * forward, unnormalised, complex.  Generic decimation-in-time mixed radix. */
typedef struct { float re, im; } cpx;

typedef struct {
    int n;
    cpx *tw;      /* W_n^k, k=0..n-1 */
    cpx *scratch; /* n */
} orc_fft;

static void fft_init(orc_fft *f, int n) {
    f->n = n;
    f->tw = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    f->scratch = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        double th = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        f->tw[k].re = (float)cos(th);
        f->tw[k].im = (float)sin(th);
    }
}
static void fft_free(orc_fft *f) { free(f->tw); free(f->scratch); }

static int smallest_factor(int n) {
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    for (int p = 3; p * p <= n; p += 2) if (n % p == 0) return p;
    return n;
}

/* out[0..n) = DFT of in[0], in[stride], ... ; tws = N/n is the twiddle stride. */
static void fft_rec(const orc_fft *f, const cpx *in, cpx *out, int n, int stride, int tws) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = smallest_factor(n), q = n / p;
    for (int r = 0; r < p; ++r) fft_rec(f, in + (size_t)r * stride, out + (size_t)r * q, q, stride * p, tws * p);
    cpx t[16];
    int N = f->n;
    for (int k = 0; k < q; ++k) {
        for (int r = 0; r < p; ++r) {
            cpx v = out[r * q + k];
            cpx w = f->tw[((size_t)r * k * tws) % N];
            t[r].re = v.re * w.re - v.im * w.im;
            t[r].im = v.re * w.im + v.im * w.re;
        }
        for (int j = 0; j < p; ++j) {
            float sr = t[0].re, si = t[0].im;
            for (int r = 1; r < p; ++r) {
                cpx w = f->tw[((size_t)(r * j % p) * q * tws) % N];
                sr += t[r].re * w.re - t[r].im * w.im;
                si += t[r].re * w.im + t[r].im * w.re;
            }
            out[j * q + k].re = sr;
            out[j * q + k].im = si;
        }
    }
}

/* ------------------------------------------------------------ MFCC extractor */
typedef struct orc_mfcc {
    int ncoef;             /* K+1: "num_coefficients", src/mfcc/extractor.rs:48 */
    float *filter_bank;    /* [ncoef][240] */
    float hamming[ORC_FRAME];
    float samples[ORC_FRAME + ORC_SHIFT];
    int nsamples;          /* self.samples.len() */
    orc_fft fft;
    cpx buf[ORC_FRAME], spec[ORC_FRAME];
} orc_mfcc;

/* MfccExtractor::new + set_out_size, src/mfcc/extractor.rs:19-59 (out_size = K). */
orc_mfcc *orc_mfcc_new(int K) {
    orc_mfcc *m = (orc_mfcc *)calloc(1, sizeof(orc_mfcc));
    m->ncoef = K + 1;
    m->filter_bank = (float *)malloc(sizeof(float) * (size_t)m->ncoef * ORC_NBINS);
    orc_mel_filter_bank(ORC_SAMPLE_RATE, ORC_NBINS, m->ncoef, m->filter_bank, NULL);
    orc_hamming_window(ORC_FRAME, m->hamming);
    fft_init(&m->fft, ORC_FRAME);
    m->nsamples = 0;
    return m;
}
void orc_mfcc_free(orc_mfcc *m) {
    if (!m) return;
    fft_free(&m->fft);
    free(m->filter_bank);
    free(m);
}
/* src/mfcc/extractor.rs:66-68 */
void orc_mfcc_reset(orc_mfcc *m) { m->nsamples = 0; }

/* src/mfcc/extractor.rs:80-86,101-163: one 480-sample (already pre-emphasised)
 * frame -> K coefficients. */
static void extract_mfccs(orc_mfcc *m, const float *frame, float *out) {
    int nc = m->ncoef;
    float mag[ORC_NBINS], lg[ORC_MAX_K1], dct[ORC_MAX_K1];
    /* calculate_magnitude_spectrum :101-114 */
    for (int i = 0; i < ORC_FRAME; ++i) { m->buf[i].re = frame[i] * m->hamming[i]; m->buf[i].im = 0.f; }
    fft_rec(&m->fft, m->buf, m->spec, ORC_FRAME, 1, 1);
    for (int i = 0; i < ORC_NBINS; ++i)
        mag[i] = sqrtf((m->spec[i].re * m->spec[i].re) + (m->spec[i].im * m->spec[i].im));
    /* calculate_mel_frequency_cepstrum :135-145, then ln :121-131 */
    for (int i = 0; i < nc; ++i) {
        const float *fb = m->filter_bank + (size_t)i * ORC_NBINS;
        float s = 0.f;
        for (int j = 0; j < ORC_NBINS; ++j) s += mag[j] * mag[j] * fb[j];
        lg[i] = logf(s + FLT_MIN);
    }
    /* discrete_cosine_transform :146-163 */
    float pi_over_n = ORC_PI / (float)nc;
    for (int k = 0; k < nc; ++k) {
        float s = 0.f;
        for (int n = 0; n < nc; ++n) s += lg[n] * cosf(pi_over_n * ((float)n + 0.5f) * (float)k);
        dct[k] = 2.f * s;
    }
    /* drop coefficient 0 :84 */
    for (int k = 1; k < nc; ++k) out[k - 1] = dct[k];
}

/* src/mfcc/extractor.rs:69-79,87-97: one 160-sample shift. Returns 1 if a frame
 * was produced into out[K]. */
static int process_audio_part(orc_mfcc *m, const float *part, float *out) {
    float pre[ORC_SHIFT];
    float tmp_sample = 0.f; /* reset at the start of EVERY shift, :88 */
    for (int i = 0; i < ORC_SHIFT; ++i) {
        float previous = tmp_sample;
        tmp_sample = part[i];
        pre[i] = tmp_sample - ORC_PRE_EMPHASIS * previous;
    }
    if (m->nsamples >= ORC_FRAME) {
        memmove(m->samples, m->samples + ORC_SHIFT, sizeof(float) * (size_t)(m->nsamples - ORC_SHIFT));
        m->nsamples -= ORC_SHIFT;
        memcpy(m->samples + m->nsamples, pre, sizeof(pre));
        m->nsamples += ORC_SHIFT;
        extract_mfccs(m, m->samples, out);
        return 1;
    }
    memcpy(m->samples + m->nsamples, pre, sizeof(pre));
    m->nsamples += ORC_SHIFT;
    return 0;
}

/* MfccExtractor::compute, src/mfcc/extractor.rs:60-65 (chunks_exact(160)).
 * out must hold (n/160)*K floats.  Returns the number of frames produced. */
int orc_mfcc_compute(orc_mfcc *m, const float *samples, int n, float *out) {
    int K = m->ncoef - 1, frames = 0;
    for (int off = 0; off + ORC_SHIFT <= n; off += ORC_SHIFT)
        frames += process_audio_part(m, samples + off, out + (size_t)frames * K);
    return frames;
}

/* Whole-stream helper: feeds the stream in 480-sample chunks exactly like
 * src/mfcc/wav_file_extractor.rs:59-66 / tests/detector.rs:361-368 do (a tail
 * shorter than 480 samples is discarded).  n_frames = 3*floor(N/480) - 3. */
long orc_mfcc_stream(const float *pcm, long N, int K, float *out) {
    orc_mfcc *m = orc_mfcc_new(K);
    long frames = 0;
    for (long off = 0; off + ORC_FRAME <= N; off += ORC_FRAME)
        frames += orc_mfcc_compute(m, pcm + off, ORC_FRAME, out + frames * K);
    orc_mfcc_free(m);
    return frames;
}

/* ------------------------------------------------------------- normalizer */
/* src/mfcc/normalizer.rs:3-31 */
void orc_normalize(const float *in, int n, int K, float *out) {
    float sum[ORC_MAX_K1];
    if (n == 0) return;
    for (int j = 0; j < K; ++j) sum[j] = 0.f;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < K; ++j) { float v = in[(size_t)i * K + j]; sum[j] += v; out[(size_t)i * K + j] = v; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < K; ++j) out[(size_t)i * K + j] -= sum[j] / (float)n;
}

/* ------------------------------------------------------------- comparator */
/* src/mfcc/comparator.rs:28-48 */
static float cosine_similarity(const float *a, const float *b, int K) {
    float dot_ab = 0.f, dot_a = 0.f, dot_b = 0.f;
    for (int d = 0; d < K; ++d) {
        float ca = a[d], cb = b[d];
        dot_ab += ca * cb;
        dot_a += ca * ca;
        dot_b += cb * cb;
    }
    float magnitude = sqrtf(dot_a * dot_b);
    if (magnitude == 0.f) return 0.f;
    return dot_ab / magnitude;
}
/* src/mfcc/comparator.rs:15-17 */
static float calculate_distance(const float *a, const float *b, int K) { return 1.f - cosine_similarity(a, b, K); }

static float min3_fold(float insertion, float deletion, float matches) {
    /* [insertion, deletion, matches].iter().fold(INF, |a,&b| a.min(b)), src/mfcc/dtw.rs:85-87 */
    float a = INFINITY;
    a = fminf(a, insertion); a = fminf(a, deletion); a = fminf(a, matches);
    return a;
}

/* Dtw::compute_optimal_path_with_window, src/mfcc/dtw.rs:56-105.
 * a = first_sequence [m][K] (the template), b = second_sequence [n][K] (the window).
 * Returns final[m-1][n-1] == D[m-1][n]  (dtw.rs:101). */
float orc_dtw_banded(const float *a, int m, const float *b, int n, int K, int w) {
    int diff = m > n ? m - n : n - m;
    int window = w > diff ? w : diff;
    size_t cols = (size_t)n + 1;
    float *D = (float *)malloc(sizeof(float) * (size_t)(m + 1) * cols);
    for (size_t i = 0; i < (size_t)(m + 1) * cols; ++i) D[i] = INFINITY;
    D[0] = 0.f;
    for (int r = 1; r <= m; ++r) {
        int start = (r > window) ? ((r - window) > 1 ? (r - window) : 1) : 1;
        int end = (n + 1 < r + window) ? n + 1 : r + window; /* exclusive */
        for (int c = start; c < end; ++c) {
            float cost = calculate_distance(a + (size_t)(r - 1) * K, b + (size_t)(c - 1) * K, K);
            float mn = min3_fold(D[(size_t)(r - 1) * cols + c], D[(size_t)r * cols + c - 1], D[(size_t)(r - 1) * cols + c - 1]);
            D[(size_t)r * cols + c] = cost + mn;
        }
    }
    float similarity = D[(size_t)(m - 1) * cols + n];
    free(D);
    return similarity;
}

/* Dtw::compute_optimal_path (unbanded), src/mfcc/dtw.rs:11-55; fills the caller's
 * [m][n] matrix and returns D[m-1][n-1].  Used by orc_average. */
static float dtw_full(const float *a, int m, const float *b, int n, int K, float *D) {
    D[0] = calculate_distance(a, b, K);
    for (int r = 1; r < m; ++r) D[(size_t)r * n] = calculate_distance(a + (size_t)r * K, b, K) + D[(size_t)(r - 1) * n];
    for (int c = 1; c < n; ++c) D[c] = calculate_distance(a, b + (size_t)c * K, K) + D[c - 1];
    for (int r = 1; r < m; ++r)
        for (int c = 1; c < n; ++c) {
            float cost = calculate_distance(a + (size_t)r * K, b + (size_t)c * K, K);
            D[(size_t)r * n + c] = cost + min3_fold(D[(size_t)(r - 1) * n + c], D[(size_t)r * n + c - 1], D[(size_t)(r - 1) * n + c - 1]);
        }
    return D[(size_t)(m - 1) * n + n - 1];
}

/* MfccAverager::average for ONE fold step, src/mfcc/averager.rs:7-35 with
 * Dtw::retrieve_optimal_path src/mfcc/dtw.rs:106-138.  origin [m][K] is updated
 * in place from frames [n][K]. */
void orc_average_step(float *origin, int m, const float *frames, int n, int K) {
    float *D = (float *)malloc(sizeof(float) * (size_t)m * (size_t)n);
    dtw_full(origin, m, frames, n, K, D);
    /* path: pushed from the end towards (0,0); the start cell (m-1,n-1) itself is
     * never pushed (dtw.rs:110-135), the vec starts with min(m-1,n-1) zero pairs. */
    int r = m - 1, c = n - 1;
    int cap = m + n + (r < c ? r : c) + 4, np = 0;
    int *px = (int *)malloc(sizeof(int) * (size_t)cap), *py = (int *)malloc(sizeof(int) * (size_t)cap);
    for (int i = 0; i < (r < c ? r : c); ++i) { px[np] = 0; py[np] = 0; ++np; }
    while (r > 0 || c > 0) {
        if (r > 0 && c > 0) {
            float ins = D[(size_t)(r - 1) * n + c], del = D[(size_t)r * n + c - 1], mat = D[(size_t)(r - 1) * n + c - 1];
            float mn = min3_fold(ins, del, mat);
            if (mn == mat) { --r; --c; } else if (mn == ins) { --r; } else if (mn == del) { --c; }
        } else if (r > 0 && c == 0) { --r; } else if (r == 0 && c > 0) { --c; }
        px[np] = r; py[np] = c; ++np;
    }
    /* avgs[x][index] = [origin[x][index], frames[y][index] for each path (x,y) in REVERSED order] */
    float *sum = (float *)malloc(sizeof(float) * (size_t)m * K);
    int *cnt = (int *)calloc((size_t)m, sizeof(int));
    for (int i = 0; i < m * K; ++i) sum[i] = origin[i];
    for (int i = 0; i < m; ++i) cnt[i] = 1;
    for (int i = np - 1; i >= 0; --i) { /* path.reverse() then iterate */
        int x = px[i], y = py[i];
        for (int k = 0; k < K; ++k) sum[(size_t)x * K + k] += frames[(size_t)y * K + k];
        cnt[x] += 1;
    }
    /* note: iter().sum() adds origin first then the pushed values in order -> same as above */
    for (int x = 0; x < m; ++x)
        for (int k = 0; k < K; ++k) origin[(size_t)x * K + k] = sum[(size_t)x * K + k] / (float)cnt[x];
    free(D); free(px); free(py); free(sum); free(cnt);
}

/* MfccComparator::compare + compute_probability, src/mfcc/comparator.rs:18-26 */
float orc_compare(const float *a, int m, const float *b, int n, int K, int band, float score_ref) {
    float cost = orc_dtw_banded(a, m, b, n, K, band);
    float normalized_cost = cost / (float)(m + n);
    return 1.f / (1.f + expf((normalized_cost - score_ref) / score_ref));
}

/* WakewordComparator::cut_and_normalize_frame + score_frame,
 * src/wakewords/comp/wakeword_comp.rs:22-37: window [wn][K] cut to the template's
 * length (keeping the OLDEST frames), mean-normalised, compared. */
float orc_score_window(const float *window, int wn, const float *templ, int tl, int K, int band, float score_ref) {
    int n = wn > tl ? tl : wn;
    float *norm = (float *)malloc(sizeof(float) * (size_t)n * K);
    orc_normalize(window, n, K, norm);
    float s = orc_compare(templ, tl, norm, n, K, band, score_ref);
    free(norm);
    return s;
}

/* score modes, src/config.rs:86-96 (declaration order) */
enum { ORC_AVERAGE = 0, ORC_MAX, ORC_MEDIAN, ORC_P25, ORC_P50, ORC_P75, ORC_P80, ORC_P90, ORC_P95 };

static int cmp_f32_total(const void *x, const void *y) {
    float a = *(const float *)x, b = *(const float *)y;
    return (a > b) - (a < b);
}
/* src/wakewords/comp/wakeword_comp.rs:38-49 */
static float get_percentile(const float *sorted, int n, float percentile) {
    float index = percentile / 100.0f * (float)(n - 1);
    float index_floor = floorf(index);
    if (index_floor == index) return sorted[(int)index];
    int i = (int)index_floor;
    float d = index - index_floor;
    return sorted[i] * (1.0f - d) + sorted[i + 1] * d;
}
/* src/wakewords/comp/wakeword_comp.rs:108-139.  scores are taken in the order
 * given (the reference iterates a HashMap: order unspecified, only matters for
 * the f32 rounding of Average). */
float orc_aggregate(const float *scores, int T, int mode) {
    float tmp[256];
    if (T > 256) T = 256;
    memcpy(tmp, scores, sizeof(float) * (size_t)T);
    if (mode == ORC_AVERAGE) {
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += tmp[i];
        return s / (float)T;
    }
    qsort(tmp, (size_t)T, sizeof(float), cmp_f32_total);
    switch (mode) {
    case ORC_MAX: return tmp[T - 1];
    case ORC_MEDIAN: case ORC_P50: return get_percentile(tmp, T, 50.f);
    case ORC_P25: return get_percentile(tmp, T, 25.f);
    case ORC_P75: return get_percentile(tmp, T, 75.f);
    case ORC_P80: return get_percentile(tmp, T, 80.f);
    case ORC_P90: return get_percentile(tmp, T, 90.f);
    case ORC_P95: return get_percentile(tmp, T, 95.f);
    }
    return 0.f;
}

/* ---------------------------------------------------------------- wakewords */
#define ORC_KIND_REF 0
#define ORC_KIND_NN 1
#define ORC_MAX_T 256

typedef struct {
    int kind;
    /* ref (WakewordComparator, src/wakewords/comp/wakeword_comp.rs:10-20) */
    int T, K;
    int *lens;       /* [T] */
    float **feats;   /* [T] -> [len][K] */
    int avg_len;     /* 0 = None */
    float *avg;      /* [avg_len][K] */
    float threshold, avg_threshold; /* NaN = None */
    float rms_level;
    /* nn (WakewordNN, src/wakewords/nn/wakeword_nn.rs:13-21) */
    int train_size, n_labels, none_index, n_layers;
    int dims[4];     /* layer sizes in->...->labels */
    float *W[3], *B[3];
    int uid;         /* n-th wakeword ever added to its detector: what a detection reports (the reference keeps the
                        wakeword's NAME in the partial detection, so it survives a remove_wakeword) */
} orc_wakeword;

typedef struct {
    int name;         /* wakeword index, or for NN the label index */
    int wakeword;     /* wakeword index */
    float avg_score, score;
    int n_scores;
    float scores[ORC_MAX_T];
    int counter;
    float gain;
} orc_detection;

/* WakewordComparator::run_detection, src/wakewords/comp/wakeword_comp.rs:77-152 */
static int comp_run_detection(const orc_wakeword *w, int wi, const float *window, int wn, float avg_threshold,
                              float threshold, int band, float score_ref, int score_mode, orc_detection *out) {
    if (!isnan(w->avg_threshold)) avg_threshold = w->avg_threshold;
    float avg_score = 0.f;
    if (w->avg_len > 0 && avg_threshold != 0.f) {
        avg_score = orc_score_window(window, wn, w->avg, w->avg_len, w->K, band, score_ref);
        if (avg_score < avg_threshold) return 0;
    }
    if (!isnan(w->threshold)) threshold = w->threshold;
    for (int t = 0; t < w->T; ++t)
        out->scores[t] = orc_score_window(window, wn, w->feats[t], w->lens[t], w->K, band, score_ref);
    out->n_scores = w->T;
    float score = orc_aggregate(out->scores, w->T, score_mode);
    if (score > threshold) {
        out->name = wi; out->wakeword = wi; out->avg_score = avg_score; out->score = score;
        out->counter = 0; out->gain = NAN;
        return 1;
    }
    return 0;
}

/* Linear -> ReLU -> ... -> Linear; candle Linear is x.W^T + b with W [out,in]
 * (src/wakewords/nn/wakeword_nn.rs:305-389).  x [B][dims[0]] -> out [B][dims[n_layers]].
 * Accumulation: bias added after a sequential k-ordered dot product. */
void orc_mlp_forward(const float *x, long B, int n_layers, const int *dims, float *const *W, float *const *Bv, float *out) {
    int maxd = 0;
    for (int l = 0; l <= n_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
    float *h0 = (float *)malloc(sizeof(float) * (size_t)maxd), *h1 = (float *)malloc(sizeof(float) * (size_t)maxd);
    for (long b = 0; b < B; ++b) {
        memcpy(h0, x + (size_t)b * dims[0], sizeof(float) * (size_t)dims[0]);
        for (int l = 0; l < n_layers; ++l) {
            int in = dims[l], on = dims[l + 1];
            for (int o = 0; o < on; ++o) {
                const float *wr = W[l] + (size_t)o * in;
                float s = 0.f;
                for (int i = 0; i < in; ++i) s += h0[i] * wr[i];
                s += Bv[l][o];
                if (l + 1 < n_layers && s < 0.f) s = 0.f;
                h1[o] = s;
            }
            float *t = h0; h0 = h1; h1 = t;
        }
        memcpy(out + (size_t)b * dims[n_layers], h0, sizeof(float) * (size_t)dims[n_layers]);
    }
    free(h0); free(h1);
}

/* Round-to-nearest-even f32 -> bf16 -> f32: the rounding the HIP bf16 MFMA path applies to the
 * layer-1 inputs and weights (no counterpart in the reference, which is f32 throughout). */
static float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) <= 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
/* orc_mlp_forward with layer-1 x and W rounded to bf16, f32 accumulation everywhere. */
void orc_mlp_forward_bf16(const float *x, long B, int n_layers, const int *dims, float *const *W, float *const *Bv, float *out) {
    size_t nx = (size_t)B * dims[0], nw = (size_t)dims[0] * dims[1];
    float *xr = (float *)malloc(sizeof(float) * nx), *wr = (float *)malloc(sizeof(float) * nw);
    for (size_t i = 0; i < nx; ++i) xr[i] = bf16_round(x[i]);
    for (size_t i = 0; i < nw; ++i) wr[i] = bf16_round(W[0][i]);
    float *W2[4];
    for (int l = 0; l < n_layers; ++l) W2[l] = W[l];
    W2[0] = wr;
    orc_mlp_forward(xr, B, n_layers, dims, W2, Bv, out);
    free(xr); free(wr);
}

/* src/wakewords/nn/wakeword_nn.rs:161-163 */
float orc_calc_inverse_similarity(float n1, float n2, float reference) {
    return 1.f - (1.f / (1.f + expf(((n1 - n2) - reference) / reference)));
}

/* WakewordNN::run_detection, src/wakewords/nn/wakeword_nn.rs:39-159 */
static int nn_run_detection(const orc_wakeword *w, int wi, const float *window, int wn, float avg_threshold,
                            float threshold, float score_ref, orc_detection *out) {
    int n = wn > w->train_size ? w->train_size : wn; /* truncate :145 */
    if (n * w->K != w->dims[0]) return 0;            /* candle shape error -> None :107-111 */
    float *norm = (float *)malloc(sizeof(float) * (size_t)n * w->K);
    orc_normalize(window, n, w->K, norm);
    float logits[ORC_MAX_T];
    orc_mlp_forward(norm, 1, w->n_layers, w->dims, w->W, w->B, logits);
    free(norm);
    /* get_label :47-60: max_by total_cmp returns the LAST maximum */
    int best = 0;
    for (int i = 1; i < w->n_labels; ++i) if (!(logits[i] < logits[best])) best = i;
    if (best == w->none_index) return 0;
    float ref = score_ref * 10.f; /* :39 */
    float none_prob = w->none_index >= 0 ? logits[w->none_index] : 0.f;
    float label_prob = logits[best];
    int calc_avg = avg_threshold != 0.f; /* :146 */
    float second = 0.f;
    if (calc_avg) { /* :75-83: max_by(|a,b| b.total_cmp(a)) == minimum-by-reversed => LAST max of reversed = first... */
        /* max_by with reversed comparator returns the element that is "greatest" under
         * reversed order = the MINIMUM value among p != label_prob (last one on ties). */
        int found = 0;
        for (int i = 0; i < w->n_labels; ++i) {
            if (logits[i] == label_prob) continue;
            if (!found || !(logits[i] > second)) { second = logits[i]; found = 1; }
        }
        if (!found) second = 0.f;
    }
    out->name = best; out->wakeword = wi;
    out->avg_score = calc_avg ? orc_calc_inverse_similarity(label_prob, second, ref) : 0.f;
    out->score = orc_calc_inverse_similarity(label_prob, none_prob, ref);
    out->n_scores = w->n_labels;
    for (int i = 0; i < w->n_labels; ++i) out->scores[i] = logits[i];
    out->counter = 0; out->gain = NAN;
    /* validate_scores :113-123 */
    return (out->score >= threshold && out->avg_score >= avg_threshold) ? 1 : 0;
}

/* ------------------------------------------------------------------ NN training */
/* training_loop, src/wakewords/nn/wakeword_model_train.rs:170-222: `epochs` full-batch steps of
 *   logits = model.forward(x); log_sm = log_softmax(logits) (x - max - ln(sum(exp(x - max))), candle_nn::ops);
 *   loss = nll(log_sm, labels) = -mean(log_sm[b][label_b]); SGD: w -= lr * dloss/dw.
 * candle 0.2.2 is not in the reference tree; its autograd is restated as the closed-form gradients of exactly these
 * ops (dlogits = (softmax - onehot)/B, ReLU mask on the forward activations).  W/Bv are updated in place; returns
 * the loss of the last epoch (evaluated before that epoch's update, like the value the reference prints). */
float orc_mlp_train(const float *x, const int *labels, long B, int n_layers, const int *dims, float *const *W, float *const *Bv,
                    float lr, int epochs) {
    float *act[4], *dz[4];
    for (int l = 0; l < n_layers; ++l) {
        act[l] = (float *)malloc(sizeof(float) * (size_t)B * dims[l + 1]);
        dz[l] = (float *)malloc(sizeof(float) * (size_t)B * dims[l + 1]);
    }
    const int C = dims[n_layers];
    float loss = NAN;
    for (int ep = 0; ep < epochs; ++ep) {
        for (int l = 0; l < n_layers; ++l) { /* forward, keeping every layer's output */
            const float *in = l == 0 ? x : act[l - 1];
            int ni = dims[l], no = dims[l + 1];
            for (long b = 0; b < B; ++b)
                for (int o = 0; o < no; ++o) {
                    float s = 0.f;
                    for (int i = 0; i < ni; ++i) s += in[b * ni + i] * W[l][(size_t)o * ni + i];
                    s += Bv[l][o];
                    if (l + 1 < n_layers && s < 0.f) s = 0.f;
                    act[l][b * no + o] = s;
                }
        }
        float lsum = 0.f;
        for (long b = 0; b < B; ++b) {
            const float *lg = act[n_layers - 1] + b * C;
            float mx = lg[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(lg[c] - mx);
            float lse = logf(se);
            for (int c = 0; c < C; ++c) {
                float lsm = (lg[c] - mx) - lse;
                if (c == labels[b]) lsum += -lsm;
                dz[n_layers - 1][b * C + c] = (expf(lsm) - (c == labels[b] ? 1.f : 0.f)) / (float)B;
            }
        }
        loss = lsum / (float)B;
        for (int l = n_layers - 1; l >= 0; --l) {
            const float *in = l == 0 ? x : act[l - 1];
            int ni = dims[l], no = dims[l + 1];
            if (l > 0)
                for (long b = 0; b < B; ++b)
                    for (int i = 0; i < ni; ++i) {
                        float s = 0.f;
                        for (int o = 0; o < no; ++o) s += dz[l][b * no + o] * W[l][(size_t)o * ni + i];
                        dz[l - 1][b * ni + i] = act[l - 1][b * ni + i] > 0.f ? s : 0.f;
                    }
            for (int o = 0; o < no; ++o) {
                for (int i = 0; i < ni; ++i) {
                    float g = 0.f;
                    for (long b = 0; b < B; ++b) g += dz[l][b * no + o] * in[b * ni + i];
                    W[l][(size_t)o * ni + i] = W[l][(size_t)o * ni + i] - g * lr;
                }
                float g = 0.f;
                for (long b = 0; b < B; ++b) g += dz[l][b * no + o];
                Bv[l][o] = Bv[l][o] - g * lr;
            }
        }
    }
    for (int l = 0; l < n_layers; ++l) { free(act[l]); free(dz[l]); }
    return loss;
}

/* ---------------------------------------------------------------------- VAD */
/* src/mfcc/vad.rs:3-50 */
typedef struct { float mode_value; int index; float window[50]; int voice_countdown; } orc_vad;
static void vad_reset(orc_vad *v) { for (int i = 0; i < 50; ++i) v->window[i] = NAN; v->voice_countdown = 0; v->index = 0; }
static int vad_is_voice(orc_vad *v, const float *mfcc, int K) {
    float s = 0.f;
    for (int i = 0; i < K; ++i) s += fabsf(mfcc[i]);
    float value = s / (float)K;
    v->window[v->index] = value;
    v->index = (v->index >= 49) ? 0 : v->index + 1;
    float mn = INFINITY; int any = 0;
    for (int i = 0; i < 50; ++i) if (!isnan(v->window[i])) { if (!any || v->window[i] < mn) mn = v->window[i]; any = 1; }
    mn = fmaxf(mn, 0.01f);
    float th = mn * v->mode_value;
    int n_high = 0;
    for (int i = 0; i < 50; ++i) if (v->window[i] > th) ++n_high;
    if (n_high > 10) v->voice_countdown = 500;
    if (v->voice_countdown > 0) { v->voice_countdown -= 1; return 1; }
    return 0;
}

/* ---------------------------------------------------------- audio filters */
/* src/audio/gain_normalizer_filter.rs:3-80 */
typedef struct {
    int enabled, window_size, fixed; float min_gain, max_gain, rms_level_ref, rms_level_sqrt;
    float *win; int win_len, win_cap;
} orc_gain;
float orc_rms_level(const float *signal, int n) { /* :49-55 */
    float sum_squared = 0.0f;
    for (int i = 0; i < n; ++i) sum_squared += signal[i] * signal[i];
    return sqrtf(sum_squared / (float)n);
}
static float gain_filter(orc_gain *g, float *signal, int n, float rms_level) { /* :14-41 */
    if (!isnan(g->rms_level_ref) && rms_level != 0.f) {
        if (g->win_len == g->win_cap) { g->win_cap = g->win_cap ? g->win_cap * 2 : 64; g->win = (float *)realloc(g->win, sizeof(float) * (size_t)g->win_cap); }
        g->win[g->win_len++] = rms_level;
        if (g->win_len > g->window_size) { memmove(g->win, g->win + 1, sizeof(float) * (size_t)(g->win_len - 1)); g->win_len--; }
        float s = 0.f;
        for (int i = 0; i < g->win_len; ++i) s += g->win[i];
        float frame_rms_level = s / (float)g->win_len;
        float gain = g->rms_level_sqrt / sqrtf(frame_rms_level);
        gain = roundf(gain * 10.f) / 10.f;
        /* f32::clamp(min,max) */
        if (gain < g->min_gain) gain = g->min_gain;
        if (gain > g->max_gain) gain = g->max_gain;
        if (gain != 1.f)
            for (int i = 0; i < n; ++i) { float v = signal[i] * gain; if (v < -1.f) v = -1.f; if (v > 1.f) v = 1.f; signal[i] = v; }
        return gain;
    }
    return 1.f;
}
/* src/audio/band_pass_filter.rs:5-55 */
typedef struct { int enabled; float a0, a1, a2, b1, b2, x1, x2, y1, y2; } orc_bandpass;
static void bandpass_init(orc_bandpass *b, float sample_rate, float low_cutoff, float high_cutoff) {
    float omega_low = 2.0f * ORC_PI * low_cutoff / sample_rate;
    float omega_high = 2.0f * ORC_PI * high_cutoff / sample_rate;
    float cos_omega_low = cosf(omega_low), cos_omega_high = cosf(omega_high);
    float alpha_low = sinf(omega_low) / 2.0f, alpha_high = sinf(omega_high) / 2.0f;
    float a0 = 1.0f / (1.0f + alpha_high - alpha_low);
    b->a0 = a0; b->a1 = -2.0f * cos_omega_low * a0; b->a2 = (1.0f - alpha_high - alpha_low) * a0;
    b->b1 = -2.0f * cos_omega_high * a0; b->b2 = (1.0f - alpha_high + alpha_low) * a0;
    b->x1 = b->x2 = b->y1 = b->y2 = 0.f;
}
static void bandpass_filter(orc_bandpass *b, float *signal, int n) {
    for (int i = 0; i < n; ++i) {
        float x = signal[i];
        float y = b->a0 * x + b->a1 * b->x1 + b->a2 * b->x2 - b->b1 * b->y1 - b->b2 * b->y2;
        signal[i] = y;
        b->x2 = b->x1; b->x1 = x; b->y2 = b->y1; b->y1 = y;
    }
}

/* The audio front-end of process_audio over a whole stream (src/detector.rs:358-371): rms level of the raw
 * chunk, gain normaliser, band-pass; out [N], rms / gains [N/480].  window_size as on_wakeword_change sets it. */
void orc_frontend_stream(const float *pcm, long N, int gain_on, float gain_ref_fixed /*NaN = not fixed*/, float min_gain,
                         float max_gain, float rms_level_ref, int window_size, int bp_on, float low_cutoff, float high_cutoff,
                         float *out, float *rms, float *gains) {
    orc_gain g; memset(&g, 0, sizeof(g));
    g.enabled = gain_on; g.min_gain = min_gain; g.max_gain = max_gain;
    g.fixed = !isnan(gain_ref_fixed);
    g.rms_level_ref = g.fixed ? gain_ref_fixed : rms_level_ref;
    g.rms_level_sqrt = sqrtf(g.rms_level_ref);
    g.window_size = window_size != 0 ? window_size : 1;
    orc_bandpass b; memset(&b, 0, sizeof(b));
    b.enabled = bp_on;
    if (bp_on) bandpass_init(&b, (float)ORC_SAMPLE_RATE, low_cutoff, high_cutoff);
    long nch = N / ORC_FRAME;
    memcpy(out, pcm, sizeof(float) * (size_t)N);
    for (long c = 0; c < nch; ++c) {
        float *buf = out + c * ORC_FRAME;
        float r = orc_rms_level(buf, ORC_FRAME);
        float gain = 1.f;
        if (gain_on) gain = gain_filter(&g, buf, ORC_FRAME, r);
        if (bp_on) bandpass_filter(&b, buf, ORC_FRAME);
        if (rms) rms[c] = r;
        if (gains) gains[c] = gain;
    }
    free(g.win);
}

/* ------------------------------------------------------------------ resampler */
/* rubato 0.14.1 `FftFixedInOut<f32>` (Cargo.lock; not vendored in the reference tree), as the reference drives
 * it: `FftFixedInOut::new(fs_in, 16000, chunk_size_in = 480, 1)` and one `process_into_buffer` per input
 * frame, src/audio/encoder.rs:57-60,72-83.  Restated from the crate's published algorithm (synchro.rs
 * FftFixedInOut::new / FftResampler::{new,resample_unit}, sinc.rs make_sincs, windows.rs blackman_harris):
 *   gcd = gcd(fs_in, fs_out); fft_chunks = ceil(chunk_size_in / (fs_out/gcd));
 *   fft_size_in = fft_chunks*fs_in/gcd, fft_size_out = fft_chunks*fs_out/gcd       (48 kHz: 1440 -> 480,
 *   so a 48 kHz "frame" is 30 ms too and get_samples_per_frame() returns 1440)
 *   cutoff = rubato_cutoff(min(fft_size_in, fft_size_out)) [* fft_size_out/fft_size_in when downsampling]
 *   filter_t[n] = sinc_bh2[n] / (2*fft_size_in), n < fft_size_in, zero padded to 2*fft_size_in, real FFT
 *   unit: zero pad the chunk to 2*fft_size_in, real FFT, multiply the first new_len bins by the filter
 *         spectrum, zero the rest, inverse real FFT of length 2*fft_size_out (unnormalised),
 *         out = first half + overlap; overlap = second half.
 * The window / sinc / filter taps are built in f32 like the crate does; the two transforms are evaluated as
 * f64 DFT sums and rounded to f32 where realfft stores Complex<f32> / f32 (rustfft's own f32 butterflies
 * differ from any other f32 FFT by a few 1e-7 of the signal level; the f64 sums sit in the middle of that
 * cloud).  Pinned by the reference's goldens for 48 kHz input: tests/resources/oye_casa_real.rpw (MFCCs of the
 * six resampled 48 kHz wavs, tests/wakeword.rs:57-71) and tests/detector.rs:163-255. */
typedef struct orc_resampler {
    int fs_in, fs_out, fft_in, fft_out, new_len;
    float *filt_re, *filt_im; /* filter_f, fft_in+1 bins */
    float *overlap;           /* fft_out */
    double *cs_in, *sn_in;    /* cos/sin(2 pi j / (2 fft_in)) */
    double *cs_out, *sn_out;  /* cos/sin(2 pi j / (2 fft_out)) */
    float *in_re, *in_im;     /* input_f after the filter product, new_len bins */
} orc_resampler;

static int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static float rubato_sinc(float value) { /* sinc.rs: sin(pi x)/(pi x) */
    if (value == 0.f) return 1.f;
    return sinf(value * ORC_PI) / (value * ORC_PI);
}

void orc_resampler_free(orc_resampler *r) {
    if (!r) return;
    free(r->filt_re); free(r->filt_im); free(r->overlap); free(r->cs_in); free(r->sn_in); free(r->cs_out); free(r->sn_out);
    free(r->in_re); free(r->in_im); free(r);
}

/* Relative anti-aliasing cutoff of the windowed sinc: rubato's sinc.rs calculate_cutoff(npoints, BlackmanHarris2) =
 * 1 / (k1/n + k2/n^2 + k3/n^3 + 1), used by FftResampler::new with n = the shorter of the two transform lengths.
 * The three constants are quoted from the crate's published source (it is not vendored here).  They were checked
 * independently: fitting the cutoff alone against tests/resources/oye_casa_real.rpw (4 680 MFCC values; the error has
 * a sharp V-shaped minimum) gives 0.9716115 +- 1e-6 for n = 480, the formula gives 0.9716114. */
static float rubato_cutoff(int npoints) {
    /* sinc.rs calculate_cutoff::<f32>(npoints, BlackmanHarris2), evaluated in f32 like the crate does */
    const float k1 = 13.745202940783823f, k2 = 121.73532586374934f, k3 = 5964.163279612051f;
    const float n = (float)npoints;
    return 1.0f / (k1 / n + k2 / (n * n) + k3 / (n * n * n) + 1.0f);
}
float orc_resampler_cutoff(int npoints) { return rubato_cutoff(npoints); }
orc_resampler *orc_resampler_new(int fs_in, int fs_out, int chunk_size_in) {
    if (fs_in <= 0 || fs_out <= 0) return NULL; /* validate_sample_rates -> "Unsupported sample rate, ..." encoder.rs:78 */
    orc_resampler *r = (orc_resampler *)calloc(1, sizeof(orc_resampler));
    int g = gcd_int(fs_in, fs_out);
    int fft_chunks = (int)ceilf((float)chunk_size_in / (float)(fs_out / g)); /* sic: the wanted size is divided by the OUTPUT granule */
    r->fs_in = fs_in; r->fs_out = fs_out;
    r->fft_out = fft_chunks * fs_out / g;
    r->fft_in = fft_chunks * fs_in / g;
    int fi = r->fft_in, fo = r->fft_out;
    r->new_len = fi < fo ? fi + 1 : fo;
    float cutoff = fi > fo ? rubato_cutoff(fo) * (float)fo / (float)fi : rubato_cutoff(fi);
    /* windows.rs blackman_harris (periodic form), squared for BlackmanHarris2 */
    float *win = (float *)malloc(sizeof(float) * (size_t)fi), *y = (float *)malloc(sizeof(float) * (size_t)fi);
    float pi2 = 2.0f * ORC_PI, pi4 = 4.0f * ORC_PI, pi6 = 6.0f * ORC_PI, np_f = (float)fi;
    for (int x = 0; x < fi; ++x) {
        float xf = (float)x;
        float w = 0.35875f - 0.48829f * cosf(pi2 * xf / np_f) + 0.14128f * cosf(pi4 * xf / np_f) - 0.01168f * cosf(pi6 * xf / np_f);
        win[x] = w * w;
    }
    /* sinc.rs make_sincs(npoints = fft_in, factor = 1, cutoff, BlackmanHarris2) */
    float sum = 0.f;
    for (int x = 0; x < fi; ++x) {
        float val = win[x] * rubato_sinc(((float)x - (float)(fi / 2)) * cutoff / 1.0f);
        sum += val;
        y[x] = val;
    }
    sum /= 1.0f;
    double *ft = (double *)calloc(2 * (size_t)fi, sizeof(double));
    for (int n = 0; n < fi; ++n) ft[n] = (double)((y[n] / sum) / (float)(2 * fi));
    int Ni = 2 * fi, No = 2 * fo;
    r->cs_in = (double *)malloc(sizeof(double) * (size_t)Ni); r->sn_in = (double *)malloc(sizeof(double) * (size_t)Ni);
    r->cs_out = (double *)malloc(sizeof(double) * (size_t)No); r->sn_out = (double *)malloc(sizeof(double) * (size_t)No);
    for (int j = 0; j < Ni; ++j) { double th = 2.0 * 3.14159265358979323846 * (double)j / (double)Ni; r->cs_in[j] = cos(th); r->sn_in[j] = sin(th); }
    for (int j = 0; j < No; ++j) { double th = 2.0 * 3.14159265358979323846 * (double)j / (double)No; r->cs_out[j] = cos(th); r->sn_out[j] = sin(th); }
    r->filt_re = (float *)malloc(sizeof(float) * (size_t)(fi + 1)); r->filt_im = (float *)malloc(sizeof(float) * (size_t)(fi + 1));
    for (int k = 0; k <= fi; ++k) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < fi; ++n) { int j = (int)(((long)k * n) % Ni); sr += ft[n] * r->cs_in[j]; si -= ft[n] * r->sn_in[j]; }
        r->filt_re[k] = (float)sr; r->filt_im[k] = (float)si;
    }
    r->overlap = (float *)calloc((size_t)fo, sizeof(float));
    r->in_re = (float *)malloc(sizeof(float) * (size_t)r->new_len); r->in_im = (float *)malloc(sizeof(float) * (size_t)r->new_len);
    free(win); free(y); free(ft);
    return r;
}
int orc_resampler_in_len(const orc_resampler *r) { return r->fft_in; }
int orc_resampler_out_len(const orc_resampler *r) { return r->fft_out; }
void orc_resampler_filter(const orc_resampler *r, float *re, float *im) {
    memcpy(re, r->filt_re, sizeof(float) * (size_t)(r->fft_in + 1)); memcpy(im, r->filt_im, sizeof(float) * (size_t)(r->fft_in + 1));
}

/* FftResampler::resample_unit: in[fft_in] -> out[fft_out] */
void orc_resampler_process(orc_resampler *r, const float *in, float *out) {
    int fi = r->fft_in, fo = r->fft_out, Ni = 2 * fi, No = 2 * fo, nl = r->new_len;
    for (int k = 0; k < nl; ++k) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < fi; ++n) { int j = (int)(((long)k * n) % Ni); sr += (double)in[n] * r->cs_in[j]; si -= (double)in[n] * r->sn_in[j]; }
        float xr = (float)sr, xi = (float)si; /* input_f: Complex<f32> */
        float hr = r->filt_re[k], hi = r->filt_im[k];
        r->in_re[k] = xr * hr - xi * hi;      /* Complex<f32> *= */
        r->in_im[k] = xr * hi + xi * hr;
    }
    /* inverse real FFT, length No, bins nl.. (always including No/2) are zero; Hermitian extension, the imaginary part of bin 0 is ignored */
    for (int n = 0; n < No; ++n) {
        double acc = (double)r->in_re[0];
        for (int k = 1; k < nl; ++k) {
            int j = (int)(((long)k * n) % No);
            acc += 2.0 * ((double)r->in_re[k] * r->cs_out[j] - (double)r->in_im[k] * r->sn_out[j]);
        }
        float v = (float)acc; /* output_buf: f32 */
        if (n < fo) out[n] = v + r->overlap[n];
        else r->overlap[n - fo] = v;
    }
}

/* A whole stream through the encoder the way wav_file_extractor.rs:83-96 does: chunks_exact(input frame) ->
 * resample -> concatenate.  Returns the number of output samples. */
long orc_resample_stream(const float *pcm, long N, int fs_in, float *out) {
    orc_resampler *r = orc_resampler_new(fs_in, ORC_SAMPLE_RATE, ORC_FRAME);
    if (!r) return -1;
    long n_out = 0;
    for (long off = 0; off + r->fft_in <= N; off += r->fft_in) { orc_resampler_process(r, pcm + off, out + n_out); n_out += r->fft_out; }
    orc_resampler_free(r);
    return n_out;
}

/* ----------------------------------------------------------------- detector */
/* Mirrors struct Rustpotter, src/detector.rs:34-92.  Input at another sample rate goes through the
 * orc_resampler above first (orc_detector_process_resampled). */
typedef struct orc_detector {
    float avg_threshold, threshold; int min_scores, eager, score_mode; float score_ref; int band_size;
    int has_vad; orc_vad vad;
    orc_gain gain_f; orc_bandpass bp;
    orc_mfcc *mfcc; int K;
    orc_wakeword *ww; int n_ww;
    float *window; int win_len, win_cap; /* audio_mfcc_window */
    int max_mfcc_frames;
    int has_partial; orc_detection partial;
    int detection_countdown;
    float rms_level, gain;
    float *frames_tmp;
    int next_uid;
} orc_detector;

/* Rustpotter::new, src/detector.rs:95-141.  vad_mode: 0 none, 1 easy, 2 medium, 3 hard. */
orc_detector *orc_detector_new(float avg_threshold, float threshold, int min_scores, int eager, float score_ref,
                               int band_size, int score_mode, int vad_mode,
                               int gain_enabled, float gain_ref /*NaN=None*/, float min_gain, float max_gain,
                               int bp_enabled, float low_cutoff, float high_cutoff) {
    orc_detector *d = (orc_detector *)calloc(1, sizeof(orc_detector));
    d->avg_threshold = avg_threshold; d->threshold = threshold; d->min_scores = min_scores; d->eager = eager;
    d->score_ref = score_ref; d->band_size = band_size; d->score_mode = score_mode;
    d->has_vad = vad_mode != 0;
    d->vad.mode_value = vad_mode == 1 ? 2.f : vad_mode == 2 ? 2.5f : 3.f; /* src/config.rs:140-146 */
    vad_reset(&d->vad);
    d->gain_f.enabled = gain_enabled; d->gain_f.min_gain = min_gain; d->gain_f.max_gain = max_gain;
    d->gain_f.rms_level_ref = gain_ref; d->gain_f.rms_level_sqrt = isnan(gain_ref) ? NAN : sqrtf(gain_ref);
    d->gain_f.fixed = !isnan(gain_ref); d->gain_f.window_size = 1;
    d->bp.enabled = bp_enabled;
    if (bp_enabled) bandpass_init(&d->bp, (float)ORC_SAMPLE_RATE, low_cutoff, high_cutoff);
    d->gain = 1.f; d->rms_level = 0.f;
    return d;
}

/* Rustpotter::reset, src/detector.rs:290-302 */
static void det_reset(orc_detector *d) {
    d->has_partial = 0;
    d->win_len = 0;
    if (d->mfcc) orc_mfcc_reset(d->mfcc);
    if (d->has_vad) vad_reset(&d->vad);
}
void orc_detector_reset(orc_detector *d) { det_reset(d); }

/* Rustpotter::update_detector_config, src/detector.rs:262-280: the new values, a fresh VadDetector, reset().
 * (the wakeword detectors take score_ref / band_size / score_mode from the detector on every call here) */
void orc_detector_update_config(orc_detector *d, float avg_threshold, float threshold, int min_scores, int eager,
                                float score_ref, int band_size, int score_mode, int vad_mode) {
    d->avg_threshold = avg_threshold; d->threshold = threshold; d->min_scores = min_scores; d->eager = eager;
    d->score_ref = score_ref; d->band_size = band_size; d->score_mode = score_mode;
    d->has_vad = vad_mode != 0;
    d->vad.mode_value = vad_mode == 1 ? 2.f : vad_mode == 2 ? 2.5f : 3.f;
    vad_reset(&d->vad);
    det_reset(d);
}

/* Rustpotter::update_filters_config, src/detector.rs:284-288: both filters are rebuilt from the config -- the new
 * GainNormalizerFilter has window_size 1 and, without a fixed gain_ref, NO reference level (NaN: gain stays 1) until
 * the wakeword set changes again (set_rms_level_ref is only called from on_wakeword_change, :328-346) -- then reset(). */
void orc_detector_update_filters(orc_detector *d, int gain_enabled, float gain_ref /*NaN=None*/, float min_gain, float max_gain,
                                 int bp_enabled, float low_cutoff, float high_cutoff) {
    d->gain_f.enabled = gain_enabled; d->gain_f.min_gain = min_gain; d->gain_f.max_gain = max_gain;
    d->gain_f.rms_level_ref = gain_ref; d->gain_f.rms_level_sqrt = isnan(gain_ref) ? NAN : sqrtf(gain_ref);
    d->gain_f.fixed = !isnan(gain_ref); d->gain_f.window_size = 1; d->gain_f.win_len = 0;
    d->bp.enabled = bp_enabled;
    if (bp_enabled) bandpass_init(&d->bp, (float)ORC_SAMPLE_RATE, low_cutoff, high_cutoff);
    det_reset(d);
}

/* on_wakeword_change, src/detector.rs:328-346 */
static void on_wakeword_change(orc_detector *d) {
    int mx = 0; float target_rms = NAN;
    for (int i = 0; i < d->n_ww; ++i) {
        orc_wakeword *w = &d->ww[i];
        int fs = 0;
        if (w->kind == ORC_KIND_REF) { for (int t = 0; t < w->T; ++t) if (w->lens[t] > fs) fs = w->lens[t]; }
        else fs = w->train_size;
        if (fs > mx) mx = fs;
        target_rms = fmaxf(w->rms_level, target_rms); /* f32::max ignores NaN */
    }
    d->max_mfcc_frames = mx;
    if (d->gain_f.enabled) { /* set_rms_level_ref, gain_normalizer_filter.rs:42-48 */
        if (!d->gain_f.fixed) { d->gain_f.rms_level_ref = target_rms; d->gain_f.rms_level_sqrt = sqrtf(target_rms); }
        int ws = d->max_mfcc_frames / 3;
        d->gain_f.window_size = ws != 0 ? ws : 1;
    }
}

static int add_common(orc_detector *d, int K) { /* add_wakeword, src/detector.rs:304-326 */
    if (d->n_ww == 0) {
        det_reset(d);
        orc_mfcc_free(d->mfcc);
        d->mfcc = orc_mfcc_new(K); /* set_out_size */
        d->K = K;
        free(d->frames_tmp); d->frames_tmp = (float *)malloc(sizeof(float) * 3 * (size_t)K);
    } else if (d->K != K) return -1; /* "Usage of wakewords with different mfcc size is not supported..." */
    d->ww = (orc_wakeword *)realloc(d->ww, sizeof(orc_wakeword) * (size_t)(d->n_ww + 1));
    memset(&d->ww[d->n_ww], 0, sizeof(orc_wakeword));
    d->ww[d->n_ww].uid = d->next_uid++;
    return d->n_ww++;
}

/* add_wakeword_ref: templates given in iteration order.  feats = concatenation of
 * the T matrices; avg may be NULL (avg_len 0); threshold/avg_threshold NaN = None. */
int orc_detector_add_ref(orc_detector *d, int T, int K, const int *lens, const float *feats, int avg_len,
                         const float *avg, float threshold, float avg_threshold, float rms_level) {
    int wi = add_common(d, K);
    if (wi < 0) return -1;
    orc_wakeword *w = &d->ww[wi];
    w->kind = ORC_KIND_REF; w->T = T; w->K = K;
    w->lens = (int *)malloc(sizeof(int) * (size_t)T);
    w->feats = (float **)malloc(sizeof(float *) * (size_t)T);
    size_t off = 0;
    for (int t = 0; t < T; ++t) {
        w->lens[t] = lens[t];
        w->feats[t] = (float *)malloc(sizeof(float) * (size_t)lens[t] * K);
        memcpy(w->feats[t], feats + off, sizeof(float) * (size_t)lens[t] * K);
        off += (size_t)lens[t] * K;
    }
    w->avg_len = avg_len;
    if (avg_len > 0) { w->avg = (float *)malloc(sizeof(float) * (size_t)avg_len * K); memcpy(w->avg, avg, sizeof(float) * (size_t)avg_len * K); }
    w->threshold = threshold; w->avg_threshold = avg_threshold; w->rms_level = rms_level;
    on_wakeword_change(d);
    return wi;
}

/* add_wakeword_model: dims [n_layers+1]; weights: for each layer W [out][in] then bias [out]. */
int orc_detector_add_model(orc_detector *d, int train_size, int K, int n_labels, int none_index, int n_layers,
                           const int *dims, const float *weights, float rms_level) {
    int wi = add_common(d, K);
    if (wi < 0) return -1;
    orc_wakeword *w = &d->ww[wi];
    w->kind = ORC_KIND_NN; w->K = K; w->train_size = train_size; w->n_labels = n_labels; w->none_index = none_index;
    w->n_layers = n_layers; w->rms_level = rms_level; w->threshold = NAN; w->avg_threshold = NAN;
    size_t off = 0;
    for (int l = 0; l <= n_layers; ++l) w->dims[l] = dims[l];
    for (int l = 0; l < n_layers; ++l) {
        size_t nw = (size_t)dims[l] * dims[l + 1];
        w->W[l] = (float *)malloc(sizeof(float) * nw); memcpy(w->W[l], weights + off, sizeof(float) * nw); off += nw;
        w->B[l] = (float *)malloc(sizeof(float) * (size_t)dims[l + 1]); memcpy(w->B[l], weights + off, sizeof(float) * (size_t)dims[l + 1]); off += (size_t)dims[l + 1];
    }
    on_wakeword_change(d);
    return wi;
}

static void wakeword_release(orc_wakeword *w) {
    if (w->kind == ORC_KIND_REF) { for (int t = 0; t < w->T; ++t) free(w->feats[t]); free(w->feats); free(w->lens); free(w->avg); }
    else for (int l = 0; l < w->n_layers; ++l) { free(w->W[l]); free(w->B[l]); }
}

/* Rustpotter::remove_wakeword, src/detector.rs:180-189: drop it and on_wakeword_change() -- no reset: the frames already
 * in the window stay, even if the window is now longer than the largest remaining wakeword needs */
int orc_detector_remove(orc_detector *d, int index) {
    if (index < 0 || index >= d->n_ww) return 0;
    wakeword_release(&d->ww[index]);
    memmove(&d->ww[index], &d->ww[index + 1], sizeof(orc_wakeword) * (size_t)(d->n_ww - index - 1));
    d->n_ww -= 1;
    on_wakeword_change(d);
    return 1;
}

void orc_detector_free(orc_detector *d) {
    if (!d) return;
    for (int i = 0; i < d->n_ww; ++i) wakeword_release(&d->ww[i]);
    free(d->ww); free(d->window); free(d->frames_tmp); free(d->gain_f.win);
    orc_mfcc_free(d->mfcc);
    free(d);
}

/* run_wakeword_detectors, src/detector.rs:433-447: best score wins (stable sort
 * desc => first of equal scores in iteration order). */
static int run_wakeword_detectors(orc_detector *d, orc_detection *out) {
    int found = 0; orc_detection tmp;
    for (int i = 0; i < d->n_ww; ++i) {
        orc_wakeword *w = &d->ww[i];
        int ok = w->kind == ORC_KIND_REF
            ? comp_run_detection(w, w->uid, d->window, d->win_len, d->avg_threshold, d->threshold, d->band_size, d->score_ref, d->score_mode, &tmp)
            : nn_run_detection(w, w->uid, d->window, d->win_len, d->avg_threshold, d->threshold, d->score_ref, &tmp);
        if (ok && (!found || tmp.score > out->score)) { *out = tmp; found = 1; }
    }
    return found;
}

/* run_detection, src/detector.rs:398-432 */
static int det_run_detection(orc_detector *d, orc_detection *out) {
    if (d->detection_countdown != 0) d->detection_countdown -= 1;
    if (d->has_partial) {
        int done = d->detection_countdown == 0 ? 1 : (d->eager && d->partial.counter >= d->min_scores); /* :448-454 */
        if (done) {
            orc_detection taken = d->partial; d->has_partial = 0; /* take() */
            if (taken.counter >= d->min_scores) { det_reset(d); *out = taken; return 1; }
        }
    }
    orc_detection det;
    if (run_wakeword_detectors(d, &det)) {
        det.counter = d->has_partial ? d->partial.counter + 1 : 1;
        det.gain = d->gain;
        if (!d->has_partial || d->partial.score < det.score) { d->partial = det; d->has_partial = 1; }
        else d->partial.counter = det.counter;
        d->detection_countdown = d->max_mfcc_frames / 2;
    }
    return 0;
}

/* process_new_mfccs, src/detector.rs:377-397 */
static int process_new_mfccs(orc_detector *d, const float *frame, orc_detection *out) {
    int result = 0, K = d->K;
    int should_run = d->has_partial || (d->has_vad ? vad_is_voice(&d->vad, frame, K) : 1);
    if (d->win_len == d->win_cap) { d->win_cap = d->win_cap ? d->win_cap * 2 : 256; d->window = (float *)realloc(d->window, sizeof(float) * (size_t)d->win_cap * K); }
    memcpy(d->window + (size_t)d->win_len * K, frame, sizeof(float) * (size_t)K);
    d->win_len++;
    if (d->win_len >= d->max_mfcc_frames && should_run) result = det_run_detection(d, out);
    if (d->win_len >= d->max_mfcc_frames && d->win_len > 0) { /* drain(0..1) */
        memmove(d->window, d->window + K, sizeof(float) * (size_t)(d->win_len - 1) * K);
        d->win_len--;
    }
    return result;
}

/* process_audio, src/detector.rs:347-376: one encoded chunk of n f32 samples at 16 kHz (480 for 16 kHz input,
 * the resampler's output length otherwise).  Returns 1 and fills *out on a detection. */
#define ORC_MAX_CHUNK 8192
int orc_detector_process_n(orc_detector *d, const float *samples, int n, orc_detection *out) {
    if (d->n_ww == 0 || n > ORC_MAX_CHUNK) return 0;
    float buf[ORC_MAX_CHUNK];
    float frames[(ORC_MAX_CHUNK / ORC_SHIFT) * ORC_MAX_K1];
    memcpy(buf, samples, sizeof(float) * (size_t)n);
    d->rms_level = orc_rms_level(buf, n);
    if (d->gain_f.enabled) d->gain = gain_filter(&d->gain_f, buf, n, d->rms_level);
    if (d->bp.enabled) bandpass_filter(&d->bp, buf, n);
    int nf = orc_mfcc_compute(d->mfcc, buf, n, frames);
    for (int i = 0; i < nf; ++i) /* find_map: stop at the first Some */
        if (process_new_mfccs(d, frames + (size_t)i * d->K, out)) return 1;
    return 0;
}
int orc_detector_process(orc_detector *d, const float *samples480, orc_detection *out) {
    return orc_detector_process_n(d, samples480, ORC_FRAME, out);
}

/* process_samples for input at another sample rate: one input frame of orc_resampler_in_len samples goes
 * through the resampler first (encoder.rs:41-60), the detector sees its output. */
int orc_detector_process_resampled(orc_detector *d, orc_resampler *r, const float *samples, orc_detection *out) {
    float enc[ORC_MAX_CHUNK];
    if (r->fft_out > ORC_MAX_CHUNK) return 0;
    orc_resampler_process(r, samples, enc);
    return orc_detector_process_n(d, enc, r->fft_out, out);
}

/* process_samples::<i16> path: v as f32 / i16::MAX as f32, src/audio/audio_types.rs:108-117 */
int orc_detector_process_i16(orc_detector *d, const int16_t *samples480, orc_detection *out) {
    float buf[ORC_FRAME];
    for (int i = 0; i < ORC_FRAME; ++i) buf[i] = (float)samples480[i] / 32767.f;
    return orc_detector_process(d, buf, out);
}

/* Getters used by tests (src/detector.rs:212-229). */
int orc_detector_state(const orc_detector *d, int *win_len, int *countdown, int *partial_counter, float *partial_score) {
    *win_len = d->win_len; *countdown = d->detection_countdown;
    *partial_counter = d->has_partial ? d->partial.counter : -1;
    *partial_score = d->has_partial ? d->partial.score : NAN;
    return d->max_mfcc_frames;
}

/* ---------------------------------------------------------- synthetic input */
/* SURVEY.md §8(d) / BASELINE.md §2: x[s][i] = u - 0.5, u = (splitmix64(seed ^ (s<<32 + i)) >> 40) / 2^24 */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_synth_pcm(uint64_t seed, uint64_t stream, long N, float *out) {
    for (long i = 0; i < N; ++i) {
        uint64_t h = splitmix64(seed ^ ((stream << 32) + (uint64_t)i));
        out[i] = (float)(h >> 40) / 16777216.f - 0.5f;
    }
}

/* ---------------------------------------------------- batched reference path
 * Scores every window start s of a stream against T templates exactly the way
 * the detector would (cut to L_t oldest frames, normalise, banded DTW, logistic),
 * i.e. Score[s][t] for s in [0, n_frames - Lmax].  This is the unit of work
 * ("scoring") of BASELINE.md; used for parity tests and the cpu_baseline leg. */
long orc_score_stream(const float *mfcc, long n_frames, int K, int T, const int *lens, const float *feats,
                      int band, float score_ref, int score_mode, float *scores /*[n_win][T]*/, float *agg /*[n_win]*/) {
    int Lmax = 0;
    for (int t = 0; t < T; ++t) if (lens[t] > Lmax) Lmax = lens[t];
    long n_win = n_frames - Lmax + 1;
    if (n_win <= 0) return 0;
    for (long s = 0; s < n_win; ++s) {
        size_t off = 0;
        for (int t = 0; t < T; ++t) {
            scores[(size_t)s * T + t] = orc_score_window(mfcc + (size_t)s * K, Lmax, feats + off, lens[t], K, band, score_ref);
            off += (size_t)lens[t] * K;
        }
        if (agg) agg[s] = orc_aggregate(scores + (size_t)s * T, T, score_mode);
    }
    return n_win;
}

typedef struct {
    uint64_t seed; long s0, s1, N; int K, T; const int *lens; const float *feats; int band; float score_ref; int mode;
    double checksum; long scorings;
} bench_job;

static void *bench_worker(void *arg) {
    bench_job *j = (bench_job *)arg;
    float *pcm = (float *)malloc(sizeof(float) * (size_t)j->N);
    long max_frames = 3 * (j->N / ORC_FRAME);
    float *mfcc = (float *)malloc(sizeof(float) * (size_t)max_frames * j->K);
    float *scores = (float *)malloc(sizeof(float) * (size_t)max_frames * j->T);
    float *agg = (float *)malloc(sizeof(float) * (size_t)max_frames);
    for (long s = j->s0; s < j->s1; ++s) {
        orc_synth_pcm(j->seed, (uint64_t)s, j->N, pcm);
        long nf = orc_mfcc_stream(pcm, j->N, j->K, mfcc);
        long nw = orc_score_stream(mfcc, nf, j->K, j->T, j->lens, j->feats, j->band, j->score_ref, j->mode, scores, agg);
        for (long w = 0; w < nw; ++w) j->checksum += agg[w];
        j->scorings += nw;
    }
    free(pcm); free(mfcc); free(scores); free(agg);
    return NULL;
}

/* CPU baseline: S synthetic streams of N samples against T templates, `threads`
 * pthreads (streams partitioned contiguously).  Returns wall seconds; writes the
 * number of scorings and a checksum (sum of aggregated scores). */
double orc_bench(uint64_t seed, long S, long N, int K, int T, const int *lens, const float *feats, int band,
                 float score_ref, int mode, int threads, long *scorings_out, double *checksum_out) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; bench_job jobs[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < threads; ++i) {
        bench_job *j = &jobs[i];
        j->seed = seed; j->s0 = S * i / threads; j->s1 = S * (i + 1) / threads; j->N = N; j->K = K; j->T = T;
        j->lens = lens; j->feats = feats; j->band = band; j->score_ref = score_ref; j->mode = mode;
        j->checksum = 0.0; j->scorings = 0;
        pthread_create(&th[i], NULL, bench_worker, j);
    }
    long sc = 0; double cs = 0.0;
    for (int i = 0; i < threads; ++i) { pthread_join(th[i], NULL); sc += jobs[i].scorings; cs += jobs[i].checksum; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (scorings_out) *scorings_out = sc;
    if (checksum_out) *checksum_out = cs;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

int orc_sizeof_detection(void) { return (int)sizeof(orc_detection); }
