// rp_host.h -- host side of librustpotter_hip: constant tables, .rpw reader, device
// context, and the C++ mirror of the reference's `Rustpotter` (src/detector.rs).
#pragma once
#include <map>
#include <tuple>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rustpotter_hip.h"
#include "rp_kernels.h"

namespace rp {

// ---------------------------------------------------------------- constant tables
struct HostTables {
    int K1 = 0;
    std::vector<float> hamming, fb, dct;
    std::vector<float2> tw240, tw480;
    std::vector<int> centres;
};
// MfccExtractor::new / set_out_size(K) tables, src/mfcc/extractor.rs:19-59,115-120,164-198
HostTables build_tables(int K);

// ------------------------------------------------------------------- .rpw reader
struct WakewordRefData {  // src/wakewords/wakeword_ref.rs:12-20 (+ wakeword_v2.rs:8-16)
    std::string name;
    std::vector<std::string> tnames;  // file order
    std::vector<int> lens;
    std::vector<std::vector<float>> feats;  // [T] -> [len*K]
    bool has_avg = false;
    int avg_len = 0;
    std::vector<float> avg;
    bool has_threshold = false, has_avg_threshold = false;
    float threshold = 0.f, avg_threshold = 0.f;
    float rms_level = 0.f;
    int mfcc_size = 0;
};
struct WakewordModelData {  // src/wakewords/wakeword_model.rs:11-18,68-72
    std::vector<std::string> labels;
    size_t train_size = 0;
    int mfcc_size = 0;
    std::string m_type;
    std::map<std::string, std::pair<std::vector<size_t>, std::vector<float>>> weights;  // name -> (dims, data)
    float rms_level = 0.f;
};
enum class RpwKind { Ref, Model };
// WakewordV2 -> WakewordRef -> WakewordModel fall-through of src/detector.rs:152-176
bool parse_rpw(const uint8_t *buf, size_t len, RpwKind *kind, WakewordRefData *ref, WakewordModelData *model,
               std::string *err);

// ---------------------------------------------------------------- device context
struct Ctx;
// builder / writer (rp_builder.cpp)
bool compute_wav_mfccs(Ctx *ctx, const uint8_t *buf, size_t len, int K, std::vector<float> *mfcc, int *frames, float *rms_level);
bool build_wakeword_ref(Ctx *ctx, const std::string &name, const float *threshold, const float *avg_threshold, size_t n,
                        const char *const *sample_names, const uint8_t *const *wavs, const size_t *wav_lens, int mfcc_size,
                        bool rms_median, WakewordRefData *out);
std::vector<uint8_t> serialize_wakeword_ref(const WakewordRefData &r);
std::vector<uint8_t> serialize_wakeword_model(const WakewordModelData &m);
// trainer (rp_train.cpp)
bool model_dims(int m_type, size_t input_len, int mfcc_size, size_t n_labels, std::vector<int> *dims);
bool train_wakeword_model(Ctx *ctx, const rp_train_options &opt, size_t n_train, const char *const *train_names,
                          const uint8_t *const *train_wavs, const size_t *train_lens, size_t n_test, const char *const *test_names,
                          const uint8_t *const *test_wavs, const size_t *test_lens, const WakewordModelData *prev,
                          WakewordModelData *out, float *final_loss, float *test_accuracy);

void set_last_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what);

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf();
    bool reserve(size_t bytes);  // grows, contents NOT preserved
    template <class T> T *as() const { return static_cast<T *>(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// page-locked host memory the device reads / writes in place (the single-stream path's chunk and results)
struct PinBuf {
    void *p = nullptr;      // host pointer
    void *dev = nullptr;    // the same memory as the device sees it
    size_t cap = 0;
    ~PinBuf();
    bool reserve(size_t bytes);  // grows, contents NOT preserved
    template <class T> T *as() const { return static_cast<T *>(p); }
    template <class T> T *dev_as() const { return static_cast<T *>(dev); }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
};

struct Ctx {
    int device = 0;
    int flags = 0;
    DtwArith arith;   // the DTW launchers' arithmetic (RP_CTX_ARITH_* at creation, rp_ctx_set_arithmetic); every template set of the context points here
    int n_cu = 256;  // compute units of the device (persistent kernels launch one workgroup per CU)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::map<int, MfccTablesDev> tables;  // by K
    std::map<size_t, std::unique_ptr<struct Resampler>> resamplers;  // by input sample rate
    DevBuf ws_resample;
    // timing
    bool timing = false;
    struct Timed { hipEvent_t a, b; int kernel; };
    std::vector<Timed> pending;
    double sum_ms[kKernelCount] = {0};
    int count[kKernelCount] = {0};
    // staging for RP_CTX_HOST_POINTERS
    DevBuf stage_in, stage_out, stage_out2, stage_out3;
    // intermediates of rp_batch_detect
    DevBuf ws_mfcc, ws_scores, ws_agg, ws_avg, ws_vad, ws_ring, ws_rms, ws_gain, ws_list, ws_hot;
    DevBuf ws_dtw;  // [2 * kDtwSchedChunks | 2 + 2 * kDtwFixCap] uint32: tile counters and fix list of the DTW launchers, zero between calls
    // rp_batch_detect_ingest: copy stream, two device blocks of PCM, per block "copy landed" / "kernels done with it" events
    hipStream_t copy_stream = nullptr;
    hipEvent_t ingest_landed[2] = {nullptr, nullptr}, ingest_freed[2] = {nullptr, nullptr};
    DevBuf ws_ingest;
    bool ingest_ready();   // creates the stream and events on first use
    // per-stream "a window of this stream can fire" flags between the aggregate pass and the scan: zero between calls (the scan clears
    // what it reads; a call that fails in between leaves spurious 1s, which only cost the next scan its shortcut), zeroed here when the
    // buffer grows
    uint32_t *hot_flags(size_t S);
    // the list of rows the wakeword-model forward computes again with the f32 matrix instructions (kMlpF16x2, rp_kernels.h): [2 + B]
    // words, the first two zero between calls
    DevBuf ws_mlp_redo;
    uint32_t *mlp_redo(size_t B);
    std::string last_mlp_kernel;   // what the last dense-row forward ran (rp_ctx_last_mlp_kernel)
    mutable uint32_t dtw_ran = 0;  // kDtwRan* of the DTW launches since rp_ctx_dtw_kernels last read it
    DtwWork dtw_work() const { return DtwWork{ws_dtw.as<uint32_t>(), ws_dtw.as<uint32_t>() + 2 * kDtwSchedChunks, &dtw_ran}; }
    // ... plus the blocks dtw_ragged_kernel needs for a call over S streams x n_win windows (whole-stream batches; a failed reservation
    // only means that kernel is not taken)
    DevBuf ws_rag;
    DtwWork dtw_work_for(size_t S, size_t rows);

    static Ctx *create(int device, int flags);
    ~Ctx();
    const MfccTablesDev *tables_for(int K);
    const Resampler *resampler_for(size_t fs_in);  // plans are cached per input rate
    void time_begin(int kernel);
    void time_end();
    void time_collect();
};

// AudioEncoder's resampler (src/audio/encoder.rs:72-83) as a device-resident linear map, rp_resampler.cpp
bool resampler_frame_lengths(size_t fs_in, size_t *in_len, size_t *out_len);
struct Resampler {
    Ctx *ctx = nullptr;
    ResamplerDev dev;
    DevBuf g2t, fft48;
    static Resampler *create(Ctx *ctx, size_t fs_in);
};

struct Templates {
    Ctx *ctx = nullptr;
    TemplatesDev dev;
    static Templates *create(Ctx *ctx, int T, int K, const int *lens, const float *feats, int avg_len,
                             const float *avg);
    ~Templates();
};

struct Model {
    Ctx *ctx = nullptr;
    MlpDev dev;
    bool mfma_ok = false;  // layer-1 shape supported by the MFMA kernels
    std::vector<float *> W, B;  // plain per-layer device copies for the generic kernel
    std::vector<int> dims;
    std::vector<float> w1_host;                     // layer-1 weights [dims[1]][dims[0]]
    std::map<int, std::unique_ptr<DevBuf>> wsums;   // per mfcc_size K: [16*nt][K], sum over frames of the layer-1 weights
    const float *wsum_for(int K);                   // (takes the window mean out after layer 1, launch_mlp_mfma_windows)
    std::unique_ptr<DevBuf> stream_img[6];          // per precision (kMlpF32 / kMlpBf16 / kMlpF16x2 / kMlpBf16x3): layer-1 weights in the stream kernel's fragment order
    int stream_ksteps = 0;                          // k-steps of 32 the images hold (zero padded past dims[0])
    // plan of launch_mlp_stream for rows starting at x (x decides the phases); false: use launch_mlp_mfma
    bool stream_plan(const float *x, size_t B, int precision, MlpStreamPlan *plan);
    // weights/biases: HOST arrays, W_l [dims[l+1]][dims[l]], b_l [dims[l+1]]
    static Model *create(Ctx *ctx, int n_layers, const int *dims, const float *const *weights, const float *const *biases);
    ~Model();
};

// ------------------------------------------------------------ `Rustpotter` mirror
struct Detection {  // src/detector.rs:488-501
    std::string name;
    float avg_score = 0.f, score = 0.f;
    std::vector<std::string> score_names;
    std::vector<float> scores;
    size_t counter = 0;
    float gain = 0.f;
};

class Rustpotter {
public:
    static Rustpotter *create(const rp_config &cfg);
    ~Rustpotter();
    bool add_wakeword_from_buffer(const std::string &key, const uint8_t *buf, size_t len);
    bool add_wakeword_from_file(const std::string &key, const std::string &path);
    bool remove_wakeword(const std::string &key);
    bool remove_wakewords();
    size_t get_samples_per_frame() const { return in_len_ * (size_t)fmt_.channels; }  // AudioEncoder::get_input_frame_length
    size_t get_bytes_per_frame() const;
    const Detection *get_partial_detection() const { return has_partial_ ? &partial_ : nullptr; }
    float get_rms_level() const { return rms_level_; }
    float get_gain() const { return gain_; }
    float get_rms_level_ref() const;
    // returns 1 detection, 0 none, <0 error
    int process_bytes(const uint8_t *bytes, size_t len, Detection *out);
    template <class T> int process_samples(const T *samples, size_t n, Detection *out);
    void update_detector_config(const rp_detector_config &c);
    void update_filters_config(const rp_filters_config &c);
    void reset();

private:
    Rustpotter() = default;
    int encode_and_process(float *mono, Detection *out);   // resample (if any) -> process_audio
    int process_audio(float *buf, size_t n, Detection *out);
    void on_wakeword_change();
    bool add_wakeword_ref(const std::string &key, WakewordRefData &&ref);
    bool add_wakeword_model(const std::string &key, WakewordModelData &&model);
    bool prepare_first(int K);
    bool run_detection(int frame_slot, Detection *out);

    struct Wakeword;
    bool gate_first_ = false;   // process_audio: score the averaged template before the others (the gate rejected the last chunk)
    std::unique_ptr<Ctx> ctx_;
    rp_audio_fmt fmt_{};
    rp_detector_config det_{};
    rp_filters_config filt_{};
    std::vector<std::pair<std::string, std::unique_ptr<Wakeword>>> wakewords_;  // insertion order
    int K_ = 0;
    // AudioEncoder (src/audio/encoder.rs): frame lengths and, for input that is not 16 kHz, the resampler
    // plan with the previous input frame kept on the device
    size_t in_len_ = 480, out_len_ = 480;
    const Resampler *rs_ = nullptr;
    PinBuf rs_x_, enc_;          // [2*in_len] previous | current input frame, [out_len] the encoded (16 kHz) chunk:
                                 // page-locked host memory the resampler kernel reads / writes in place
    // extractor state (src/mfcc/extractor.rs:14, :66-79): the last two 10 ms shifts and how many are buffered
    size_t shifts_seen_ = 0;     // capped at 3: a frame is emitted from the 4th shift on
    PinBuf up_;                  // [160 pad | 2 buffered shifts | new shifts]: the MFCC kernel reads it in place
    // audio_mfcc_window (src/detector.rs:69): device history + explicit length
    DevBuf hist_;                // [hist_cap][K]
    size_t hist_cap_ = 0, n_hist_ = 0, win_len_ = 0, max_mfcc_frames_ = 0;
    DevBuf nn_x_, nn_s0_, nn_s1_;
    PinBuf result_;              // [frames | per wakeword score blocks]: written by the kernels, read by the host
    // detection state (src/detector.rs:71-79)
    bool has_partial_ = false;
    Detection partial_;
    size_t countdown_ = 0;
    float rms_level_ = 0.f, gain_ = 1.f;
    // VAD (src/mfcc/vad.rs)
    struct Vad { float mode_value = 2.f; size_t index = 0; float window[50]; size_t voice_countdown = 0; void reset(); bool is_voice(const float *mfcc, int K); };
    bool has_vad_ = false;
    Vad vad_;
    // filters (src/audio/gain_normalizer_filter.rs, band_pass_filter.rs)
    struct Gain { bool enabled = false, fixed = false; size_t window_size = 1; float min_gain = 0.1f, max_gain = 1.f, rms_level_ref = 0.f, rms_level_sqrt = 0.f; std::vector<float> win; };
    struct BandPass { bool enabled = false; float a0, a1, a2, b1, b2, x1, x2, y1, y2; };
    Gain gainf_;
    BandPass bp_;
    void configure_filters();
};

}  // namespace rp
