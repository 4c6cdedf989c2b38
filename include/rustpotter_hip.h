/*
 * rustpotter_hip.h -- C ABI of librustpotter_hip.so, an MI355X (gfx950) HIP
 * implementation of rustpotter v3.0.2's MFCC + DTW wakeword scoring path.
 *
 * The reference has no C ABI of its own; its public surface is the
 * Rust API re-exported in src/lib.rs:8-21.  Every entry point below cites the
 * reference item it replaces (paths relative to the reference root).  A Rust
 * crate binds this header with `extern "C"` (bindings/rustpotter_hip.rs,
 * INTEGRATION.md) and re-exposes the reference's method names unchanged.
 *
 * Conventions (SURVEY.md §8b "Ownership/Errors/Threading"):
 *  - all functions return plain C types; status returns are int: >=0 ok, <0 error;
 *    the message of the last error of the calling thread is rp_last_error()
 *    (the reference returns Result<_, String>).  A NULL handle or a NULL required pointer is
 *    refused with -1 ("null handle" / "null argument"); getters return 0, *_free(NULL) is a no-op.
 *  - the caller owns every input buffer for the duration of the call only;
 *    wakewords are copied into the handle (src/wakewords/comp/wakeword_comp.rs:55-63).
 *  - strings/arrays inside rp_detection are owned by the handle and stay valid
 *    until the next call on that handle.
 *  - handles are NOT thread-safe (reference: every method takes &mut self); distinct
 *    handles are independent.
 *  - nothing in this library falls back to the CPU: without a usable HIP device
 *    rp_new / rp_ctx_new fail with an error.
 */
#ifndef RUSTPOTTER_HIP_H
#define RUSTPOTTER_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ config */
/* src/audio/audio_types.rs:4-9 */
typedef enum { RP_SAMPLE_I8 = 0, RP_SAMPLE_I16 = 1, RP_SAMPLE_I32 = 2, RP_SAMPLE_F32 = 3 } rp_sample_format;
/* src/audio/audio_types.rs:52-56 */
typedef enum { RP_ENDIAN_BIG = 0, RP_ENDIAN_LITTLE = 1, RP_ENDIAN_NATIVE = 2 } rp_endianness;
/* src/config.rs:86-96 (declaration order) */
typedef enum {
    RP_SCORE_AVERAGE = 0, RP_SCORE_MAX = 1, RP_SCORE_MEDIAN = 2, RP_SCORE_P25 = 3, RP_SCORE_P50 = 4,
    RP_SCORE_P75 = 5, RP_SCORE_P80 = 6, RP_SCORE_P90 = 7, RP_SCORE_P95 = 8
} rp_score_mode;
/* src/config.rs:134-138; RP_VAD_NONE == Option::None */
typedef enum { RP_VAD_NONE = 0, RP_VAD_EASY = 1, RP_VAD_MEDIUM = 2, RP_VAD_HARD = 3 } rp_vad_mode;

/* src/config.rs:10-19 AudioFmt */
typedef struct {
    size_t sample_rate;
    rp_sample_format sample_format;
    uint16_t channels;
    rp_endianness endianness;
} rp_audio_fmt;

/* src/config.rs:172-191 DetectorConfig */
typedef struct {
    float avg_threshold;
    float threshold;
    size_t min_scores;
    bool eager;
    float score_ref;
    uint16_t band_size;
    rp_score_mode score_mode;
    rp_vad_mode vad_mode;
} rp_detector_config;

/* src/config.rs:32-42 GainNormalizationConfig (gain_ref: Option<f32>) */
typedef struct {
    bool enabled;
    bool has_gain_ref;
    float gain_ref;
    float min_gain;
    float max_gain;
} rp_gain_normalization_config;

/* src/config.rs:55-62 BandPassConfig */
typedef struct {
    bool enabled;
    float low_cutoff;
    float high_cutoff;
} rp_band_pass_config;

/* src/config.rs:75-82 FiltersConfig */
typedef struct {
    rp_gain_normalization_config gain_normalizer;
    rp_band_pass_config band_pass;
} rp_filters_config;

/* src/config.rs:212-219 RustpotterConfig */
typedef struct {
    rp_audio_fmt fmt;
    rp_detector_config detector;
    rp_filters_config filters;
} rp_config;

/* RustpotterConfig::default(), src/config.rs:20-29,43-52,63-71,192-207 */
void rp_config_default(rp_config *out);

/* --------------------------------------------------------------- detection */
/* src/detector.rs:488-501 RustpotterDetection.  `scores` is the HashMap<String,f32>
 * flattened to parallel arrays (template file names for references, label names
 * for models). */
typedef struct {
    const char *name;
    float avg_score;
    float score;
    size_t n_scores;
    const char *const *score_names;
    const float *scores;
    size_t counter;
    float gain;
} rp_detection;

/* ------------------------------------------- single-stream `Rustpotter` mirror */
typedef struct rp_detector rp_detector;

/* Rustpotter::new, src/detector.rs:95-141.  config->fmt.sample_rate may be any rate AudioEncoder::new
 * (src/audio/encoder.rs:63-83) accepts whose 16 kHz output frame is 480 or 640 samples -- every standard
 * audio rate: 8 / 16 / 32 / 48 / 96 kHz and the 44.1 kHz family give 30 ms input frames (3 MFCC frames per
 * call), 11.025 / 22.05 kHz give 40 ms frames (640 encoded samples, 4 MFCC frames per call).  Input that is
 * not 16 kHz goes through the device resampler (rubato FftFixedInOut restated);
 * rp_get_samples_per_frame() is then the resampler's input frame x channels (1 440 for 48 kHz mono).
 * Errors: "Unsupported sample rate, unable to initialize the resampler" (src/audio/encoder.rs:78) for rate 0
 * and for rates whose output frame has no device kernel (rp_resampler_frame_lengths tells). */
int rp_new(const rp_config *config, rp_detector **out);
void rp_free(rp_detector *d);

/* add_wakeword_from_buffer / _from_file, src/detector.rs:152-176 (WakewordV2 ->
 * WakewordRef -> WakewordModel fall-through, decided on the CBOR map keys). */
int rp_add_wakeword_from_buffer(rp_detector *d, const char *key, const uint8_t *buffer, size_t len);
int rp_add_wakeword_from_file(rp_detector *d, const char *key, const char *path);
/* remove_wakeword / remove_wakewords, src/detector.rs:180-202 */
bool rp_remove_wakeword(rp_detector *d, const char *key);
bool rp_remove_wakewords(rp_detector *d);

/* src/detector.rs:204-229 */
size_t rp_get_samples_per_frame(const rp_detector *d);
size_t rp_get_bytes_per_frame(const rp_detector *d);
/* returns 1 and fills *out if a partial detection exists, else 0 */
int rp_get_partial_detection(const rp_detector *d, rp_detection *out);
float rp_get_rms_level(const rp_detector *d);
float rp_get_gain(const rp_detector *d);
float rp_get_rms_level_ref(const rp_detector *d);

/* process_bytes, src/detector.rs:234-240; process_samples::<T>, :245-254.
 * Return 1 (detection written to *out), 0 (None: also for a wrong buffer length or
 * no wakewords, exactly like the reference), <0 device error. */
int rp_process_bytes(rp_detector *d, const uint8_t *audio_bytes, size_t len, rp_detection *out);
int rp_process_samples_i8(rp_detector *d, const int8_t *samples, size_t n, rp_detection *out);
int rp_process_samples_i16(rp_detector *d, const int16_t *samples, size_t n, rp_detection *out);
int rp_process_samples_i32(rp_detector *d, const int32_t *samples, size_t n, rp_detection *out);
int rp_process_samples_f32(rp_detector *d, const float *samples, size_t n, rp_detection *out);

/* update_config / update_detector_config / update_filters_config / reset,
 * src/detector.rs:257-302 */
int rp_update_config(rp_detector *d, const rp_config *config);
int rp_update_detector_config(rp_detector *d, const rp_detector_config *config);
int rp_update_filters_config(rp_detector *d, const rp_filters_config *config);
void rp_reset(rp_detector *d);

/* Result<_, String> error text of the calling thread's last failing call */
const char *rp_last_error(void);

/* ------------------------------------------------- batched operator level
 * What the HIP kernels sit behind: the reference's per-frame operators applied to
 * S independent streams at once.  All array arguments are DEVICE pointers unless
 * the context was created with RP_CTX_HOST_POINTERS (then the library stages
 * them through its own device buffers). */
typedef struct rp_ctx rp_ctx;
typedef struct rp_templates rp_templates;

/* RP_CTX_FULL_SCORES: rp_batch_detect* compare every window with every sample template to the end even when the
 * averaged-template gate (avg_threshold != 0, wakeword_comp.rs:85-93) would skip them or the running cost already rules a
 * detection out -- the behaviour of the per-window score outputs, forced for calls that do not ask for them (same
 * detections either way; used to time the paths against each other). */
enum {
    RP_CTX_DEVICE_POINTERS = 0, RP_CTX_HOST_POINTERS = 1, RP_CTX_FULL_SCORES = 2,
    /* the arithmetic of the context's DTW scoring (RP_ARITH_* below; neither bit: RP_ARITH_F32_MATRIX) */
    RP_CTX_ARITH_STRICT_F32 = 4, RP_CTX_ARITH_FAST_SPLIT = 8,
    /* with RP_CTX_ARITH_FAST_SPLIT: references whose templates differ in length also go to the matrix cores (dtw_ragged_kernel) */
    RP_CTX_RAGGED_MATRIX = 16
};

/* device: HIP device ordinal.  Fails (<0) if no HIP device is usable, or when both RP_CTX_ARITH_* bits are set. */
int rp_ctx_new(int device, int flags, rp_ctx **out);
/* The arithmetic of the cosine products of the DTW cost (replaces the implicit "f32" of src/mfcc/comparator.rs:28-48, whose products and
 * sums are f32 -- the config-is-a-struct convention of src/config.rs:172-219, not an environment variable).  The window mean, the norms, the
 * min-plus recurrence, the score and every index are f32 / integer arithmetic in every mode; the modes differ in where the five products of
 * a band cell are formed and how many bits of them are kept:
 *   RP_ARITH_F32_MATRIX (default)  matrix cores, f32-grade: both operands as three bf16 parts (exact: 3 x 8 = an f32's 24 significant bits),
 *       six of the nine partial products accumulated in f32 -- what is dropped is below 2^-22 of a product (2^-25.7 rms; an f32 multiply
 *       rounds by up to 2^-24, 2^-25.3 rms): mfcc_size 5 in chunks of 3..8 templates of one length, mfcc_size 13 / 16 in chunks of up to
 *       four.  Template sets with no such kernel (lengths that occur once or twice, other frame sizes, a band other than 3..5) run the
 *       f32 vector kernels.
 *   RP_ARITH_STRICT_F32            f32 vector FMAs for every product (the "register" kernels), nothing on the matrix cores.
 *   RP_ARITH_FAST_SPLIT            matrix cores with two f16 parts per operand (22 significant bits, one partial product dropped): the
 *       fastest form, NARROWER than the reference's f32 products; scores stay within the 1e-5 parity gate for score_ref >= 0.05.
 *       ragged_matrix != 0 additionally sends references of unequal template lengths to dtw_ragged_kernel (same two-part products).
 * May be changed between calls (a template set serves every mode); rp_ctx_dtw_kernels reports what ran.  Returns 0, <0 on a bad value. */
enum { RP_ARITH_F32_MATRIX = 0, RP_ARITH_STRICT_F32 = 1, RP_ARITH_FAST_SPLIT = 2 };
int rp_ctx_set_arithmetic(rp_ctx *ctx, int arith, int ragged_matrix);
/* the current RP_ARITH_* (ragged_matrix, if not NULL, receives the flag) */
int rp_ctx_arithmetic(rp_ctx *ctx, int *ragged_matrix);
void rp_ctx_free(rp_ctx *ctx);
/* Run subsequent launches on an externally owned hipStream_t (e.g. the caller's
 * torch stream); NULL = the context's own stream. */
int rp_ctx_set_stream(rp_ctx *ctx, void *hip_stream);
int rp_ctx_synchronize(rp_ctx *ctx);
/* Diagnostics of the cosine distance (replaces nothing; src/mfcc/comparator.rs:28-48 is what it watches): the reference
 * divides by sqrt(dot_a * dot_b) in f32 and returns similarity 0 when that is 0.  The device kernels compare unit-length
 * vectors and hand every (window, templates) pair that met a squared norm outside 2^-60 .. 2^30 (frames) / 2^60 (template
 * rows) -- where the f32 product can underflow, lose bits or overflow -- to a reference-shaped kernel.  *pairs = the number
 * of pairs rescored that way by this context's DTW calls since it was made (waits for the context's stream); 0 for audio
 * in any ordinary range. */
int rp_ctx_dtw_ref_pairs(rp_ctx *ctx, uint64_t *pairs);
/* Diagnostics of the DTW dispatch (replaces nothing; src/mfcc/dtw.rs:56-105 is what every family computes): the kernel families this
 * context's DTW calls launched since the last call of this function, as a mask of RP_DTW_KERNEL_* (then cleared).  Tests and bench.py use
 * it to name the kernel a measurement belongs to. */
enum {
    RP_DTW_KERNEL_MFMA = 1,      /* dtw_mfma_kernel: chunks of 3..8 same-length templates, mfcc_size 5, cosines on the matrix cores */
    RP_DTW_KERNEL_MFMA_WIDE = 2, /* dtw_mfma_wide3_kernel (three bf16 parts) / dtw_mfma_wide_kernel (two f16 parts): mfcc_size 13 / 16 */
    RP_DTW_KERNEL_RAGGED = 4,    /* dtw_ragged_kernel: templates of unequal length on the matrix cores, mfcc_size 5 */
    RP_DTW_KERNEL_REGISTER = 8,  /* dtw_band_kernel / dtw_band2_kernel / dtw_band_wide_kernel: f32 vector arithmetic throughout */
    RP_DTW_KERNEL_GENERIC = 16,  /* dtw_generic_kernel */
    RP_DTW_KERNEL_SINGLE = 32,   /* dtw_single_kernel (a handful of windows of one stream) */
    RP_DTW_KERNEL_REF_ALL = 64,  /* dtw_ref_kernel over every window (a template row outside the norm range) */
    RP_DTW_KERNEL_MFMA_GROUP = 128, /* dtw_mfma_group_kernel: several chunks of one template length share a column's operand (same bits as MFMA) */
    /* the product arithmetic of the matrix-core launches among the above */
    RP_DTW_PRODUCTS_BF16X3 = 256,  /* three bf16 parts per operand: f32-grade (RP_ARITH_F32_MATRIX) */
    RP_DTW_PRODUCTS_F16X2 = 512    /* two f16 parts per operand: 22 bits (RP_ARITH_FAST_SPLIT) */
};
int rp_ctx_dtw_kernels(rp_ctx *ctx);
/* Which build this library is (replaces nothing): the target architecture and the compiler flags it differs by from the
 * product's Makefile defaults -- "gfx950" for the product, "gfx950 +-DRP_..." for an experiment build (tools/ab.sh builds
 * those into rustpotter_amd/variants/, never over the product).  bench.py prints it on its JSON line. */
const char *rp_build_info(void);
/* Diagnostics of the wakeword-model forward (replaces nothing; src/wakewords/nn/wakeword_nn.rs:101-106 is what it computes): the
 * kernel(s) the last rp_mlp_forward_batch / rp_mlp_forward_windows / rp_batch_detect_model of this context ran and their operand format, e.g. "mlp_stream_kernel<f16x2 splits> +
 * mlp_mfma_kernel<f32> on listed rows".  The string belongs to the context and is valid until its next forward; "" before the first. */
const char *rp_ctx_last_mlp_kernel(rp_ctx *ctx);

/* Number of MFCC frames MfccExtractor::compute yields for a stream of n_samples fed
 * in 480-sample chunks from a fresh extractor: 3*floor(n/480) - 3
 * (src/mfcc/extractor.rs:60-79; the frame made of the first three shifts is never
 * emitted). */
size_t rp_mfcc_num_frames(size_t n_samples);

/* MfccExtractor::compute over whole streams, src/mfcc/extractor.rs:60-163.
 * pcm  [S][pcm_stride] f32 16 kHz mono (n_samples valid per stream)
 * mfcc [S][n_frames][K] f32, n_frames = rp_mfcc_num_frames(n_samples); K = mfcc_size
 * (set_out_size(K): K+1 filters/coefficients, coefficient 0 dropped). */
int rp_mfcc_batch(rp_ctx *ctx, const float *pcm, size_t S, size_t n_samples, size_t pcm_stride, int K, float *mfcc);

/* Same with the PCM in any of the reference's `Sample` formats (src/audio/audio_types.rs:59-137), host byte
 * order, decoded inside the kernel as `v as f32 / T::MAX as f32`: pcm [S][pcm_stride] of int8 / int16 /
 * int32 / float.  Halves (int16) the bytes the path reads from HBM. */
int rp_mfcc_batch_fmt(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                      int K, float *mfcc);

/* WakewordRef::new_from_sample_buffers (src/wakewords/comp/wakeword_ref_build.rs:9-41) followed by
 * WakewordSave::save_to_buffer (src/wakewords/wakeword_file.rs:22-26): builds a wakeword reference from
 * n 16 kHz wav buffers (PCM 8/16/32-bit or f32; first channel) -- MFCCs from the HIP kernel, whole-matrix
 * normalisation, DTW-aligned average template, rms level -- and returns it serialised as .rpw (CBOR) bytes
 * in *out_rpw (free with rp_buffer_free).  threshold / avg_threshold: NULL = None.  rms_from_files != 0
 * takes the MEDIAN sample level like new_from_sample_files (:80-81) instead of the maximum.
 * The bytes can be written to a file or handed to rp_add_wakeword_from_buffer. */
int rp_wakeword_ref_build(rp_ctx *ctx, const char *name, const float *threshold, const float *avg_threshold, size_t n,
                          const char *const *sample_names, const uint8_t *const *wav_buffers, const size_t *wav_lens,
                          uint16_t mfcc_size, int rms_from_files, uint8_t **out_rpw, size_t *out_len);
void rp_buffer_free(uint8_t *buffer);

/* WakewordModelTrain::train_from_buffers (src/wakewords/nn/wakeword_model_train.rs:44-168) followed by
 * save_to_buffer: trains a wakeword model on the device from labelled wav samples and returns it as .rpw bytes.
 * A sample's label is the lower-cased text between '[' and ']' in its name ("none" without one).  Features =
 * the whole-file-normalised MFCC matrix of every sample (any sample rate, like rp_wakeword_ref_build), flattened,
 * zero-padded / truncated to the longest training sample; network shapes of ModelType Tiny / Small / Medium / Large
 * (src/wakewords/nn/wakeword_nn.rs:305-389); `epochs` full-batch steps of log_softmax + nll + SGD(learning_rate).
 * prev_model (NULL = none): an existing model .rpw to continue from -- its labels, type, mfcc_size and train_size
 * win over the options, as in the reference.  A fresh model's weights are drawn like candle_nn::linear does
 * (weights N(0, 2/fan_in), biases U(+-1/sqrt(fan_in))) from `seed` (the reference uses the thread RNG).
 * final_loss: the loss of the last epoch; test_accuracy: share of test samples whose arg-max label is right
 * (test_model :251-272); either may be NULL.  Errors as the reference: "No training data provided", "No test data
 * provided", "Your training data need to contain at least two labels", "Forbidden label '...'...". */
typedef enum { RP_MODEL_TINY = 0, RP_MODEL_SMALL = 1, RP_MODEL_MEDIUM = 2, RP_MODEL_LARGE = 3 } rp_model_type;
typedef struct {
    int m_type;            /* rp_model_type */
    float learning_rate;
    size_t epochs, test_epochs;   /* test_epochs only paces the reference's progress printing; unused */
    uint16_t mfcc_size;
    uint64_t seed;
} rp_train_options;
int rp_wakeword_model_train(rp_ctx *ctx, const rp_train_options *options, size_t n_train, const char *const *train_names,
                            const uint8_t *const *train_wavs, const size_t *train_lens, size_t n_test,
                            const char *const *test_names, const uint8_t *const *test_wavs, const size_t *test_lens,
                            const uint8_t *prev_model, size_t prev_model_len, uint8_t **out_rpw, size_t *out_len,
                            float *final_loss, float *test_accuracy);

/* The audio front-end of Rustpotter::process_audio for whole streams (src/detector.rs:358-371): sample
 * decode, GainNormalizerFilter (src/audio/gain_normalizer_filter.rs:14-55: per 480-sample chunk, gain =
 * round(10*sqrt(ref)/sqrt(mean of the last `window_size` chunk RMS values))/10, clamped, samples clamped to
 * +-1) and BandPassFilter (src/audio/band_pass_filter.rs:19-55: biquad with state carried along the
 * stream).  Neither filter depends on detections (reset() keeps them), so the result is what the
 * detector would have fed its MFCC extractor.  rms_level_ref: what on_wakeword_change sets (the largest
 * wakeword rms_level) -- ignored when filters->gain_normalizer.has_gain_ref; window_size: max_mfcc_frames / 3
 * (src/detector.rs:337).  pcm_out [S][out_stride] f32; rms / gains [S][n_samples/480] (either may be NULL):
 * get_rms_level() and get_gain() of every chunk. */
int rp_frontend_batch(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                      const rp_filters_config *filters, float rms_level_ref, size_t window_size, float *pcm_out,
                      size_t out_stride, float *rms, float *gains);

/* A wakeword reference resident on the device: T templates [len_t][K] (already
 * mean-normalised, as stored in a .rpw: src/wakewords/wakeword_ref.rs:12-20) given
 * as HOST arrays; avg may be NULL. */
int rp_templates_new(rp_ctx *ctx, int T, int K, const int *lens, const float *feats, int avg_len, const float *avg,
                     rp_templates **out);
void rp_templates_free(rp_templates *t);
int rp_templates_max_len(const rp_templates *t);

/* WakewordComparator::run_detection scoring for every window start of every stream,
 * src/wakewords/comp/wakeword_comp.rs:22-37,77-139 + src/mfcc/comparator.rs +
 * src/mfcc/dtw.rs:56-105 + src/mfcc/normalizer.rs.  n_win = n_frames - max_len + 1.
 * scores [S][n_win][T]; avg [S][n_win] (NULL or ignored when the template set has no
 * avg template / with_avg == 0); agg [S][n_win] = score_mode aggregate.  band_size 0 is
 * legal as in the reference (u16): no cell lies in the band and every score is 0. */
int rp_dtw_score_batch(rp_ctx *ctx, const float *mfcc, size_t S, size_t n_frames, const rp_templates *t,
                       float score_ref, int band_size, rp_score_mode score_mode, int with_avg,
                       float *scores, float *avg, float *agg);

/* One detection found by rp_detect_scan */
typedef struct {
    int32_t stream;
    int32_t frame;       /* index of the MFCC frame whose processing emitted the detection */
    int32_t window;      /* window start (frame index) of the best partial = row of `scores` */
    int32_t counter;     /* RustpotterDetection::counter */
    float avg_score;
    float score;
} rp_batch_detection;

/* Rustpotter::process_new_mfccs / run_detection / reset state machine,
 * src/detector.rs:290-302,377-454, run over precomputed window scores.  With config->vad_mode set,
 * the VadDetector gate (src/mfcc/vad.rs:11-36, :379-383) is evaluated per stream from `mfcc`
 * ([S][n_frames][K], required then; may be NULL otherwise).
 * det [S][max_det] (device or host per ctx flag), n_det [S] (may exceed max_det: only the first max_det detections of a
 * stream are stored; slots behind a stream's detections are zero). */
int rp_detect_scan(rp_ctx *ctx, const float *agg, const float *avg, size_t S, size_t n_frames, int max_len,
                   const rp_detector_config *config, int avg_enabled, const float *mfcc, int K,
                   rp_batch_detection *det, int32_t *n_det, int max_det);

/* The whole path for S independent streams in one call = S x (Rustpotter::new + add_wakeword +
 * process_samples over the stream in 480-sample chunks), src/detector.rs:347-454: MFCC -> window
 * scores -> score_mode aggregate -> detection state machine; intermediate arrays live in buffers
 * owned by the context (grown on demand, reused across calls).  pcm [S][pcm_stride] f32 16 kHz mono;
 * det [S][max_det], n_det [S] as in rp_detect_scan.  Optional outputs (NULL to skip): scores
 * [S][n_win][T] and agg [S][n_win] with n_win = rp_mfcc_num_frames(n_samples) - max_len + 1.
 * The wakeword's own threshold / avg_threshold overrides (Option<f32> in the .rpw) are passed in
 * `config` by the caller; avg gate as in WakewordComparator::run_detection :83-93: with an averaged template and
 * avg_threshold != 0, windows whose avg_score is below the threshold are NOT compared with the sample templates
 * (one DTW instead of T+1) unless `scores` / `agg` are requested -- those arrays hold every window -- or the
 * context has RP_CTX_FULL_SCORES; the detections are the same on either path.  A call that asks for neither array is
 * "detect-only": in ScoreMode::Max it may also stop DTWs whose running cost already rules out a score above
 * `threshold` (cell costs are >= 0, so the cost can only grow) -- only where a whole wave of 64 windows is past that bound,
 * so every window that can fire keeps exact scores and the detections do not change. */
int rp_batch_detect(rp_ctx *ctx, const float *pcm, size_t S, size_t n_samples, size_t pcm_stride, const rp_templates *t,
                    const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det, int max_det,
                    float *scores, float *agg);

/* rp_batch_detect with the PCM in any sample format (see rp_mfcc_batch_fmt). */
int rp_batch_detect_fmt(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                        const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                        int max_det, float *scores, float *agg);

/* rp_batch_detect_fmt for streams that live in HOST memory, pipelined (the serving shape of `process_samples` over many
 * streams, src/detector.rs:234-254: the audio arrives in host buffers): the S streams are taken in blocks of `block_streams`
 * (0 = 8 192); block k+1's host-to-device copy runs on the context's copy stream while block k's kernels run on its launch
 * stream, two device blocks are cycled, the detections of every block are copied back as it finishes.  pcm [S][pcm_stride],
 * det [S][max_det] and n_det [S] are HOST arrays whatever the context's RP_CTX_HOST_POINTERS flag says; detection records carry
 * global stream ids.  The copies overlap the kernels when the host memory is page-locked (hipHostMalloc / hipHostRegister by the
 * caller: the link then runs at its rate and binds -- 862 B per scoring for f32, half for i16); pageable memory works, copy by
 * copy.  Device memory needed: two blocks of PCM + one block's intermediates, so S may exceed what fits in HBM at once.
 * seconds (NULL to skip): wall time of the call. */
int rp_batch_detect_ingest(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                           const rp_templates *t, const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det,
                           int max_det, size_t block_streams, double *seconds);

/* Multi-GPU form of rp_batch_detect_fmt: independent streams shard across the GPUs of a node, nothing is
 * exchanged but the final per-stream results.  The caller owns one context per device (rp_ctx_new(device g)) with the
 * wakeword replicated on each (rp_templates_new on every context; <= 128 KB).  Shard g = S[g] streams whose PCM
 * pcm[g] lives on ctxs[g]'s device (or in host memory if the contexts have RP_CTX_HOST_POINTERS); global stream ids are
 * assigned in shard order (shard g holds streams sum(S[0..g)) .. ).  One host thread per shard drives its device and
 * stream -- the reference's `Rustpotter: Send`, one detector per thread -- and every shard's detections are gathered
 * into ONE block: det [sum S][max_det] / n_det [sum S], `stream` fields holding global ids, in host memory
 * (RP_CTX_HOST_POINTERS) or on ctxs[0]'s device (peer copies over xGMI).  Returns when the gathered block is complete.
 * Same detections as one rp_batch_detect_fmt call over the concatenated streams. */
int rp_batch_detect_sharded(rp_ctx *const *ctxs, const rp_templates *const *t, int n_shards, const void *const *pcm,
                            rp_sample_format fmt, const size_t *S, size_t n_samples, size_t pcm_stride,
                            const rp_detector_config *config, rp_batch_detection *det, int32_t *n_det, int max_det);
/* How the calling thread's last rp_batch_detect_sharded gathered the shards' results (replaces nothing; SURVEY 8e's final gather):
 * per shard whether its device writes into the gathering device directly (peer access over xGMI) or the runtime stages the copy.
 * Valid until the thread's next sharded call; "" before the first. */
const char *rp_sharded_gather_info(void);

/* rp_batch_detect for a detector that holds SEVERAL wakewords (run_wakeword_detectors, src/detector.rs:433-447: every
 * wakeword whose own thresholds pass proposes a detection for the frame, the best score wins; max_mfcc_frames is the
 * longest template over all wakewords and each wakeword scores the oldest frames of that window,
 * wakeword_comp.rs:22-27).  t [n_wakewords] (1..8, all with the same mfcc_size, else the reference's "Usage of
 * wakewords with different mfcc size is not supported, ignoring wakeword"); thresholds / avg_thresholds
 * [n_wakewords]: the wakeword's own Option<f32> overrides, NaN = the value in `config` (either array may be NULL).
 * det_wakeword [S][max_det] (NULL to skip): index of the wakeword a detection belongs to. */
int rp_batch_detect_multi(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                          size_t n_wakewords, const rp_templates *const *t, const rp_detector_config *config,
                          const float *thresholds, const float *avg_thresholds, rp_batch_detection *det,
                          int32_t *det_wakeword, int32_t *n_det, int max_det);

/* Sample-rate conversion in front of the path: AudioEncoder::new / reencode_to_mono_with_sample_rate
 * (src/audio/encoder.rs:41-60,63-83), i.e. rubato's FftFixedInOut<f32>::new(sample_rate, 16000, 480, 1) and one
 * process_into_buffer per input frame.  rp_resampler_frame_lengths gives AudioEncoder's
 * get_input_frame_length() (per channel) and the number of 16 kHz samples one input frame yields (48 kHz:
 * 1440 -> 480).  Returns 0, or -1 for a rate the resampler cannot be built for ("Unsupported sample rate,
 * unable to initialize the resampler", encoder.rs:78). */
int rp_resampler_frame_lengths(size_t sample_rate, size_t *in_len, size_t *out_len);
/* Whole streams: pcm [S][pcm_stride] in `fmt`, `channels` interleaved channels (the first one is used,
 * encoder.rs:42-48), n_samples frames per stream at `sample_rate`; chunks_exact(in_len) frames are converted
 * (a shorter tail is dropped, src/mfcc/wav_file_extractor.rs:83-96), the stream starts from silence.
 * out [S][out_stride] receives (n_samples / in_len) * out_len f32 samples at 16 kHz per stream. */
int rp_resample_batch(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, int channels, size_t sample_rate, size_t S,
                      size_t n_samples, size_t pcm_stride, float *out, size_t out_stride);

/* Live streams: S detectors that each receive n_chunks 30 ms chunks per call -- the batched form of
 * calling Rustpotter::process_samples (src/detector.rs:347-376) once per chunk on S independent
 * instances that share one wakeword and one DetectorConfig.  Everything the reference keeps between
 * calls stays on the device: the previous chunk (MfccExtractor's sample history, src/mfcc/extractor.rs),
 * the last max_len-1 MFCC frames (Rustpotter::audio_mfcc_window), the partial detection / countdown
 * (src/detector.rs:62-79) and the VadDetector window.  Frames are numbered as in rp_batch_detect
 * (frame f is the f-th MFCC frame since the batch was created), so feeding a stream in pieces gives
 * the detections of rp_batch_detect over the concatenation. */
/* Lifetime: a stream batch borrows `ctx` and `t` -- both must outlive it (free the batch first).  A call that fails
 * (-1) leaves the batch unusable: part of its device state may already have advanced, so every later
 * rp_stream_batch_process on it fails with "stream batch is in a failed state"; free it and create a new one. */
typedef struct rp_stream_batch rp_stream_batch;
int rp_stream_batch_new(rp_ctx *ctx, const rp_templates *t, const rp_detector_config *config, size_t S,
                        size_t max_chunks_per_call, rp_stream_batch **out);
void rp_stream_batch_free(rp_stream_batch *b);
/* pcm [S][pcm_stride] holds n_chunks * rp_stream_batch_samples_per_chunk() new samples per stream (480 per chunk by
 * default; 1 <= n_chunks <= max_chunks_per_call).
 * det [S][max_det], n_det [S]: detections emitted during these chunks (n_det may exceed max_det; only
 * the first max_det are stored).  agg (NULL to skip) [S][3*n_chunks]: aggregate score of the window
 * ending at each new frame (garbage for windows that reach before frame 0).  Without `agg` (and without
 * RP_CTX_FULL_SCORES) windows below avg_threshold are not compared with the sample templates, as in rp_batch_detect. */
int rp_stream_batch_process(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                            rp_batch_detection *det, int32_t *n_det, int max_det, float *agg);
/* The AudioFmt of the streams (RustpotterConfig.fmt: sample_rate, channels; src/config.rs:10-29), to be set before
 * the first rp_stream_batch_process -- default 16 kHz mono.  Input that is not 16 kHz is converted per call
 * exactly as one Rustpotter per stream would (the first channel of every frame, rubato-style resampling with the
 * previous input frame of every stream kept on the device; the resampler is not touched by resets).  A chunk is
 * then rp_stream_batch_samples_per_chunk() samples (= get_samples_per_frame(): 1 440 for 48 kHz mono).  For the
 * 11.025 / 22.05 kHz family a chunk is a 40 ms frame (640 encoded samples): a stream then gains four MFCC frames per
 * chunk (agg has 4 * n_chunks columns) and a detection drops the rest of its 40 ms frame, as in the reference. */
int rp_stream_batch_set_input(rp_stream_batch *b, size_t sample_rate, int channels);
size_t rp_stream_batch_samples_per_chunk(const rp_stream_batch *b);
/* Rustpotter::reset (src/detector.rs:290-302) of one stream, or of all when stream < 0. */
int rp_stream_batch_reset(rp_stream_batch *b, long long stream);
/* chunks consumed so far (per stream) */
size_t rp_stream_batch_chunks_seen(const rp_stream_batch *b);

/* Live-stream batches whose detectors hold SEVERAL wakewords and / or wakeword MODELS (add_wakeword*, run_wakeword_detectors:
 * src/detector.rs:304-346,433-447): every wakeword whose own thresholds pass proposes a detection for the frame and the best
 * score wins; the window is as long as the longest wakeword and each wakeword scores its oldest frames
 * (wakeword_comp.rs:22-27, wakeword_nn.rs:137: truncate).  One entry per wakeword: exactly one of `templates` / `model`;
 * threshold / avg_threshold: a reference's own Option<f32> overrides, NaN = the value in `config`; none_index / precision as in
 * rp_batch_detect_model.  All wakewords share mfcc_size (else "Usage of wakewords with different mfcc size is not supported,
 * ignoring wakeword").  The batch borrows the context and every templates / model handle. */
typedef struct rp_model rp_model;
typedef struct {
    const rp_templates *templates;
    const rp_model *model;
    int none_index;
    int precision;
    float threshold;
    float avg_threshold;
} rp_wakeword_spec;
int rp_stream_batch_new_multi(rp_ctx *ctx, size_t n_wakewords, const rp_wakeword_spec *wakewords, int mfcc_size,
                              const rp_detector_config *config, size_t S, size_t max_chunks_per_call, rp_stream_batch **out);
/* rp_stream_batch_process that also tells which wakeword fired: det_wakeword [S][max_det] = index into `wakewords`,
 * det_label [S][max_det] = the label index when that wakeword is a model, else -1 (either may be NULL).  Works on every
 * stream batch (a batch of rp_stream_batch_new reports wakeword 0, label -1).  Fed the same audio piece by piece it gives
 * the detections of rp_batch_detect_multi / rp_batch_detect_model over the concatenation. */
int rp_stream_batch_process_multi(rp_stream_batch *b, const void *pcm, rp_sample_format fmt, size_t n_chunks, size_t pcm_stride,
                                  rp_batch_detection *det, int32_t *det_wakeword, int32_t *det_label, int32_t *n_det, int max_det);

/* A wakeword model (src/wakewords/wakeword_model.rs:11-18) resident on the device.  weights are
 * HOST arrays W_l [dims[l+1]][dims[l]] (candle Linear: x.W^T + b), biases b_l [dims[l+1]]; 1..3 layers. */
int rp_model_new(rp_ctx *ctx, int n_layers, const int *dims, const float *const *weights, const float *const *biases,
                 rp_model **out);
void rp_model_free(rp_model *m);

enum { RP_MLP_F32 = 0, RP_MLP_BF16 = 1, RP_MLP_F32_STRICT = 2, RP_MLP_F32_FAST = 3 };
/* WakewordNN forward (ModelImpl::forward: Linear -> ReLU -> ... -> Linear, raw logits),
 * src/wakewords/nn/wakeword_nn.rs:101-106,305-389: x [B][dims[0]] (the flattened, mean-normalised
 * window, :139-149,268-273) -> logits [B][dims[n_layers]].  Layer 1 runs on the matrix cores:
 * RP_MLP_F32 = f32-grade layer 1: inputs and weights as three bf16 parts each (exact: 3 x 8 = an f32's 24 significant bits), six of the
 *   nine partial products of a multiplication accumulated in f32 -- what is dropped is below 2^-22 of a product, 2^-25.7 rms (an f32
 *   multiply rounds by up to 2^-24); bf16 has the f32 exponent range, so there is no out-of-range row and no second pass; finite input
 *   never gives NaN logits.  (A feature that is +-inf splits into inf + NaN: that row's logits are NaN, where candle's f32 Linear would
 *   answer +-inf, NaN or -- every such product cut by the ReLU -- finite values; NaN features give NaN in both.)  A model whose three-part
 *   weight groups do not fit the LDS runs the f32 matrix instructions.
 * RP_MLP_F32_STRICT = the f32 matrix instructions for every row (each output a k-ordered fmaf chain, as candle's f32 Linear up to
 *   summation order).
 * RP_MLP_F32_FAST = two f16 parts per operand (22 significant bits, one partial product dropped: NARROWER than the reference's f32 products;
 *   logits within 1e-5), a row that holds a feature beyond the f16 range, |x| > 65 504, is computed by the f32 matrix instructions in a
 *   second short pass; the form the whole-stream window kernels of rp_batch_detect_model use at the matrix cores' full rate.
 * RP_MLP_BF16 = inputs rounded to bf16, f32 accumulate. */

int rp_mlp_forward_batch(rp_ctx *ctx, const rp_model *model, const float *x, size_t B, int precision, float *logits);

/* The same forward over EVERY window of S streams' MFCC rows, what WakewordNN::run_detection computes frame after frame
 * (src/wakewords/nn/wakeword_nn.rs:101-159: the window of train_size = dims[0] / mfcc_size frames, mean-normalised by
 * MfccNormalizer::normalize, flattened, through the model): mfcc [S][n_frames][mfcc_size] -> logits [S][n_win][dims[n_layers]],
 * n_win = n_frames - train_size + 1 (0 rows when a stream is shorter than a window).  The windows are never materialised: a
 * stream's frames are staged once and the window mean is taken out after layer 1 (W.(f - mu) = W.f - sum_k mu[k] wsum[k]);
 * RP_MLP_BF16 is accepted and computes as RP_MLP_F32 here (bf16 is an INPUT format of rp_mlp_forward_batch's dense rows).
 * rp_batch_detect_model is this call between rp_mfcc_batch and the score / detection passes. */
int rp_mlp_forward_windows(rp_ctx *ctx, const rp_model *model, const float *mfcc, size_t S, size_t n_frames, int mfcc_size,
                           int precision, float *logits);

/* rp_batch_detect for a wakeword MODEL (WakewordNN::run_detection, src/wakewords/nn/wakeword_nn.rs:39-159, inside the
 * detection state machine): MFCC -> windows of train_size = dims[0] / mfcc_size frames, mean-normalised -> MLP forward
 * -> per window the arg-max label (the last maximum; `none_index` = index of the "none" label, -1 without one),
 * score = 1 - 1/(1 + exp(((label - none) - 10 score_ref) / (10 score_ref))), avg_score likewise against the smallest
 * other logit when config->avg_threshold != 0, kept when score >= threshold && avg_score >= avg_threshold -> state
 * machine.  precision: RP_MLP_F32 / RP_MLP_F32_STRICT / RP_MLP_BF16.  det_label [S][max_det] (NULL to skip): label index of a detection. */
int rp_batch_detect_model(rp_ctx *ctx, const void *pcm, rp_sample_format fmt, size_t S, size_t n_samples, size_t pcm_stride,
                          const rp_model *model, int mfcc_size, int none_index, const rp_detector_config *config, int precision,
                          rp_batch_detection *det, int32_t *det_label, int32_t *n_det, int max_det);

/* Synthetic benchmark input of BASELINE.md §2, generated on the device:
 * pcm[s][i] = (splitmix64(seed ^ ((first_stream+s)<<32 + i)) >> 40) / 2^24 - 0.5 */
int rp_synth_pcm_batch(rp_ctx *ctx, uint64_t seed, uint64_t first_stream, size_t S, size_t n_samples, size_t pcm_stride,
                       float *pcm);

/* Average duration in ms of the launches of one kernel since the last reset, timed
 * with hipEvents on the launch stream (used by bench.py's roofline block).
 * kernel: 0 mfcc, 1 dtw, 2 aggregate, 3 scan, 4 mlp, 5 resample. */
int rp_ctx_timing_enable(rp_ctx *ctx, int enable);
int rp_ctx_timing_read(rp_ctx *ctx, int kernel, double *avg_ms, int *launches);
int rp_ctx_timing_reset(rp_ctx *ctx);

const char *rp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RUSTPOTTER_HIP_H */
