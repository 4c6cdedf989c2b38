//! rustpotter_hip.rs -- Rust binding of include/rustpotter_hip.h.
//!
//! NOT COMPILED in the authoring image (no rustc/cargo there).  It is the source a rustpotter
//! maintainer drops into the crate as `src/hip.rs` behind a cargo feature `hip`, so that
//! `Rustpotter` keeps its public method names and argument types while every 30 ms chunk is scored by
//! librustpotter_hip.so.  build.rs: `println!("cargo:rustc-link-lib=dylib=rustpotter_hip");`
//!
//! What stands in for a compiler here: tests/test_rust_binding.py parses this file and the header and compares every
//! function (name, argument count, argument and return types), every `#[repr(C)]` struct (field order and types) and
//! every constant with its C declaration, so the two cannot drift apart silently.
//!
//! Replaces: src/detector.rs:95-302 (the public methods of `Rustpotter`), src/mfcc/extractor.rs:60-68
//! (`MfccExtractor::compute` -> `HipContext::mfcc_batch`), src/wakewords/wakeword_detector.rs:3-14
//! (`WakewordDetector::run_detection` -> `HipContext::dtw_score_batch` / `mlp_forward_batch`), src/detector.rs:377-454
//! (the detection state machine -> `HipContext::detect_scan`, `batch_detect*`, `StreamBatch`).
#![allow(non_camel_case_types)]
use std::collections::HashMap;
use std::ffi::{c_void, CStr, CString};
use std::os::raw::{c_char, c_int};

use crate::{
    AudioFmt, BandPassConfig, DetectorConfig, Endianness, FiltersConfig, GainNormalizationConfig, RustpotterConfig, SampleFormat,
    ScoreMode, VADMode, WakewordModel, WakewordRef, WakewordSave,
};

// ------------------------------------------------------------------------------------------------ C declarations
// enums of the header are plain `int`s on this side
pub const RP_SAMPLE_I8: c_int = 0;
pub const RP_SAMPLE_I16: c_int = 1;
pub const RP_SAMPLE_I32: c_int = 2;
pub const RP_SAMPLE_F32: c_int = 3;
pub const RP_ENDIAN_BIG: c_int = 0;
pub const RP_ENDIAN_LITTLE: c_int = 1;
pub const RP_ENDIAN_NATIVE: c_int = 2;
pub const RP_SCORE_AVERAGE: c_int = 0;
pub const RP_SCORE_MAX: c_int = 1;
pub const RP_SCORE_MEDIAN: c_int = 2;
pub const RP_SCORE_P25: c_int = 3;
pub const RP_SCORE_P50: c_int = 4;
pub const RP_SCORE_P75: c_int = 5;
pub const RP_SCORE_P80: c_int = 6;
pub const RP_SCORE_P90: c_int = 7;
pub const RP_SCORE_P95: c_int = 8;
pub const RP_VAD_NONE: c_int = 0;
pub const RP_VAD_EASY: c_int = 1;
pub const RP_VAD_MEDIUM: c_int = 2;
pub const RP_VAD_HARD: c_int = 3;
pub const RP_MODEL_TINY: c_int = 0;
pub const RP_MODEL_SMALL: c_int = 1;
pub const RP_MODEL_MEDIUM: c_int = 2;
pub const RP_MODEL_LARGE: c_int = 3;
pub const RP_CTX_DEVICE_POINTERS: c_int = 0;
pub const RP_CTX_HOST_POINTERS: c_int = 1;
/// compare every window with every sample template even where the averaged-template gate would skip them
pub const RP_CTX_FULL_SCORES: c_int = 2;
/// arithmetic of the DTW cost's cosine products (neither bit: `RP_ARITH_F32_MATRIX`)
pub const RP_CTX_ARITH_STRICT_F32: c_int = 4;
pub const RP_CTX_ARITH_FAST_SPLIT: c_int = 8;
pub const RP_CTX_RAGGED_MATRIX: c_int = 16;
/// matrix cores, both operands as three bf16 parts (exact), f32 accumulate: f32-grade (default)
pub const RP_ARITH_F32_MATRIX: c_int = 0;
/// f32 vector FMAs for every product
pub const RP_ARITH_STRICT_F32: c_int = 1;
/// matrix cores, two f16 parts per operand (22 bits): fastest, narrower than the reference's f32 products
pub const RP_ARITH_FAST_SPLIT: c_int = 2;
pub const RP_DTW_KERNEL_MFMA: c_int = 1;
pub const RP_DTW_KERNEL_MFMA_WIDE: c_int = 2;
pub const RP_DTW_KERNEL_RAGGED: c_int = 4;
pub const RP_DTW_KERNEL_REGISTER: c_int = 8;
pub const RP_DTW_KERNEL_GENERIC: c_int = 16;
pub const RP_DTW_KERNEL_SINGLE: c_int = 32;
pub const RP_DTW_KERNEL_REF_ALL: c_int = 64;
pub const RP_DTW_KERNEL_MFMA_GROUP: c_int = 128;
pub const RP_DTW_PRODUCTS_BF16X3: c_int = 256;
pub const RP_DTW_PRODUCTS_F16X2: c_int = 512;
pub const RP_MLP_F32: c_int = 0;
pub const RP_MLP_BF16: c_int = 1;
pub const RP_MLP_F32_STRICT: c_int = 2;
/// two f16 parts per operand (22 bits): narrower than the reference's f32 products, the fastest whole-stream model detector
pub const RP_MLP_F32_FAST: c_int = 3;

#[repr(C)] #[derive(Clone, Copy)] pub struct rp_audio_fmt { pub sample_rate: usize, pub sample_format: c_int, pub channels: u16, pub endianness: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_detector_config { pub avg_threshold: f32, pub threshold: f32, pub min_scores: usize, pub eager: bool, pub score_ref: f32, pub band_size: u16, pub score_mode: c_int, pub vad_mode: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_gain_normalization_config { pub enabled: bool, pub has_gain_ref: bool, pub gain_ref: f32, pub min_gain: f32, pub max_gain: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_band_pass_config { pub enabled: bool, pub low_cutoff: f32, pub high_cutoff: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_filters_config { pub gain_normalizer: rp_gain_normalization_config, pub band_pass: rp_band_pass_config }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_config { pub fmt: rp_audio_fmt, pub detector: rp_detector_config, pub filters: rp_filters_config }
#[repr(C)] pub struct rp_detection { pub name: *const c_char, pub avg_score: f32, pub score: f32, pub n_scores: usize, pub score_names: *const *const c_char, pub scores: *const f32, pub counter: usize, pub gain: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_train_options { pub m_type: c_int, pub learning_rate: f32, pub epochs: usize, pub test_epochs: usize, pub mfcc_size: u16, pub seed: u64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_wakeword_spec { pub templates: *const rp_templates, pub model: *const rp_model, pub none_index: c_int, pub precision: c_int, pub threshold: f32, pub avg_threshold: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug, PartialEq)] pub struct rp_batch_detection { pub stream: i32, pub frame: i32, pub window: i32, pub counter: i32, pub avg_score: f32, pub score: f32 }
pub enum rp_detector {}
pub enum rp_ctx {}
pub enum rp_templates {}
pub enum rp_stream_batch {}
pub enum rp_model {}

// every function of include/rustpotter_hip.h, in the header's order
extern "C" {
    pub fn rp_config_default(out: *mut rp_config);
    pub fn rp_new(config: *const rp_config, out: *mut *mut rp_detector) -> c_int;
    pub fn rp_free(d: *mut rp_detector);
    pub fn rp_add_wakeword_from_buffer(d: *mut rp_detector, key: *const c_char, buffer: *const u8, len: usize) -> c_int;
    pub fn rp_add_wakeword_from_file(d: *mut rp_detector, key: *const c_char, path: *const c_char) -> c_int;
    pub fn rp_remove_wakeword(d: *mut rp_detector, key: *const c_char) -> bool;
    pub fn rp_remove_wakewords(d: *mut rp_detector) -> bool;
    pub fn rp_get_samples_per_frame(d: *const rp_detector) -> usize;
    pub fn rp_get_bytes_per_frame(d: *const rp_detector) -> usize;
    pub fn rp_get_partial_detection(d: *const rp_detector, out: *mut rp_detection) -> c_int;
    pub fn rp_get_rms_level(d: *const rp_detector) -> f32;
    pub fn rp_get_gain(d: *const rp_detector) -> f32;
    pub fn rp_get_rms_level_ref(d: *const rp_detector) -> f32;
    pub fn rp_process_bytes(d: *mut rp_detector, audio_bytes: *const u8, len: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i8(d: *mut rp_detector, samples: *const i8, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i16(d: *mut rp_detector, samples: *const i16, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i32(d: *mut rp_detector, samples: *const i32, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_f32(d: *mut rp_detector, samples: *const f32, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_update_config(d: *mut rp_detector, config: *const rp_config) -> c_int;
    pub fn rp_update_detector_config(d: *mut rp_detector, config: *const rp_detector_config) -> c_int;
    pub fn rp_update_filters_config(d: *mut rp_detector, config: *const rp_filters_config) -> c_int;
    pub fn rp_reset(d: *mut rp_detector);
    pub fn rp_last_error() -> *const c_char;

    pub fn rp_ctx_new(device: c_int, flags: c_int, out: *mut *mut rp_ctx) -> c_int;
    pub fn rp_ctx_free(ctx: *mut rp_ctx);
    pub fn rp_ctx_set_stream(ctx: *mut rp_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rp_ctx_synchronize(ctx: *mut rp_ctx) -> c_int;
    pub fn rp_ctx_dtw_ref_pairs(ctx: *mut rp_ctx, pairs: *mut u64) -> c_int;
    pub fn rp_ctx_dtw_kernels(ctx: *mut rp_ctx) -> c_int;
    pub fn rp_ctx_set_arithmetic(ctx: *mut rp_ctx, arith: c_int, ragged_matrix: c_int) -> c_int;
    pub fn rp_ctx_arithmetic(ctx: *mut rp_ctx, ragged_matrix: *mut c_int) -> c_int;
    pub fn rp_ctx_last_mlp_kernel(ctx: *mut rp_ctx) -> *const c_char;
    pub fn rp_build_info() -> *const c_char;
    pub fn rp_sharded_gather_info() -> *const c_char;
    pub fn rp_mfcc_num_frames(n_samples: usize) -> usize;
    pub fn rp_mfcc_batch(ctx: *mut rp_ctx, pcm: *const f32, S: usize, n_samples: usize, pcm_stride: usize, K: c_int, mfcc: *mut f32) -> c_int;
    pub fn rp_mfcc_batch_fmt(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize, K: c_int,
                             mfcc: *mut f32) -> c_int;
    pub fn rp_wakeword_ref_build(ctx: *mut rp_ctx, name: *const c_char, threshold: *const f32, avg_threshold: *const f32, n: usize,
                                 sample_names: *const *const c_char, wav_buffers: *const *const u8, wav_lens: *const usize,
                                 mfcc_size: u16, rms_from_files: c_int, out_rpw: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn rp_buffer_free(buffer: *mut u8);
    pub fn rp_wakeword_model_train(ctx: *mut rp_ctx, options: *const rp_train_options, n_train: usize, train_names: *const *const c_char,
                                   train_wavs: *const *const u8, train_lens: *const usize, n_test: usize, test_names: *const *const c_char,
                                   test_wavs: *const *const u8, test_lens: *const usize, prev_model: *const u8, prev_model_len: usize,
                                   out_rpw: *mut *mut u8, out_len: *mut usize, final_loss: *mut f32, test_accuracy: *mut f32) -> c_int;
    pub fn rp_frontend_batch(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize,
                             filters: *const rp_filters_config, rms_level_ref: f32, window_size: usize, pcm_out: *mut f32,
                             out_stride: usize, rms: *mut f32, gains: *mut f32) -> c_int;
    pub fn rp_templates_new(ctx: *mut rp_ctx, T: c_int, K: c_int, lens: *const c_int, feats: *const f32, avg_len: c_int, avg: *const f32,
                            out: *mut *mut rp_templates) -> c_int;
    pub fn rp_templates_free(t: *mut rp_templates);
    pub fn rp_templates_max_len(t: *const rp_templates) -> c_int;
    pub fn rp_dtw_score_batch(ctx: *mut rp_ctx, mfcc: *const f32, S: usize, n_frames: usize, t: *const rp_templates, score_ref: f32,
                              band_size: c_int, score_mode: c_int, with_avg: c_int, scores: *mut f32, avg: *mut f32, agg: *mut f32) -> c_int;
    pub fn rp_detect_scan(ctx: *mut rp_ctx, agg: *const f32, avg: *const f32, S: usize, n_frames: usize, max_len: c_int,
                          config: *const rp_detector_config, avg_enabled: c_int, mfcc: *const f32, K: c_int,
                          det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int) -> c_int;
    pub fn rp_batch_detect(ctx: *mut rp_ctx, pcm: *const f32, S: usize, n_samples: usize, pcm_stride: usize, t: *const rp_templates,
                           config: *const rp_detector_config, det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int,
                           scores: *mut f32, agg: *mut f32) -> c_int;
    pub fn rp_batch_detect_fmt(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize,
                               t: *const rp_templates, config: *const rp_detector_config, det: *mut rp_batch_detection, n_det: *mut i32,
                               max_det: c_int, scores: *mut f32, agg: *mut f32) -> c_int;
    pub fn rp_batch_detect_ingest(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize,
                                  t: *const rp_templates, config: *const rp_detector_config, det: *mut rp_batch_detection, n_det: *mut i32,
                                  max_det: c_int, block_streams: usize, seconds: *mut f64) -> c_int;
    /// ctxs[g] / t[g]: one context and one replica of the wakeword per device; pcm[g]: the S[g] streams of shard g
    /// ([S[g]][pcm_stride], on ctxs[g]'s device or in host memory per the contexts' flag); det / n_det: ONE gathered
    /// block for all sum(S) streams, `stream` = global id in shard order.  One host thread per shard inside the call.
    pub fn rp_batch_detect_sharded(ctxs: *const *mut rp_ctx, t: *const *const rp_templates, n_shards: c_int, pcm: *const *const c_void,
                                   fmt: c_int, S: *const usize, n_samples: usize, pcm_stride: usize, config: *const rp_detector_config,
                                   det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int) -> c_int;
    pub fn rp_batch_detect_multi(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize,
                                 n_wakewords: usize, t: *const *const rp_templates, config: *const rp_detector_config,
                                 thresholds: *const f32, avg_thresholds: *const f32, det: *mut rp_batch_detection,
                                 det_wakeword: *mut i32, n_det: *mut i32, max_det: c_int) -> c_int;
    pub fn rp_resampler_frame_lengths(sample_rate: usize, in_len: *mut usize, out_len: *mut usize) -> c_int;
    pub fn rp_resample_batch(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, channels: c_int, sample_rate: usize, S: usize,
                             n_samples: usize, pcm_stride: usize, out: *mut f32, out_stride: usize) -> c_int;
    pub fn rp_stream_batch_new(ctx: *mut rp_ctx, t: *const rp_templates, config: *const rp_detector_config, S: usize,
                               max_chunks_per_call: usize, out: *mut *mut rp_stream_batch) -> c_int;
    pub fn rp_stream_batch_free(b: *mut rp_stream_batch);
    /// pcm: S rows of n_chunks chunks (host pointers with RP_CTX_HOST_POINTERS, else device pointers); fmt = rp_sample_format
    pub fn rp_stream_batch_process(b: *mut rp_stream_batch, pcm: *const c_void, fmt: c_int, n_chunks: usize, pcm_stride: usize,
                                   det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int, agg: *mut f32) -> c_int;
    pub fn rp_stream_batch_set_input(b: *mut rp_stream_batch, sample_rate: usize, channels: c_int) -> c_int;
    pub fn rp_stream_batch_samples_per_chunk(b: *const rp_stream_batch) -> usize;
    pub fn rp_stream_batch_reset(b: *mut rp_stream_batch, stream: i64) -> c_int;
    pub fn rp_stream_batch_chunks_seen(b: *const rp_stream_batch) -> usize;
    pub fn rp_stream_batch_new_multi(ctx: *mut rp_ctx, n_wakewords: usize, wakewords: *const rp_wakeword_spec, mfcc_size: c_int,
                                     config: *const rp_detector_config, S: usize, max_chunks_per_call: usize, out: *mut *mut rp_stream_batch) -> c_int;
    pub fn rp_stream_batch_process_multi(b: *mut rp_stream_batch, pcm: *const c_void, fmt: c_int, n_chunks: usize, pcm_stride: usize,
                                         det: *mut rp_batch_detection, det_wakeword: *mut i32, det_label: *mut i32, n_det: *mut i32, max_det: c_int) -> c_int;
    pub fn rp_model_new(ctx: *mut rp_ctx, n_layers: c_int, dims: *const c_int, weights: *const *const f32, biases: *const *const f32,
                        out: *mut *mut rp_model) -> c_int;
    pub fn rp_model_free(m: *mut rp_model);
    pub fn rp_mlp_forward_batch(ctx: *mut rp_ctx, model: *const rp_model, x: *const f32, B: usize, precision: c_int, logits: *mut f32) -> c_int;
    pub fn rp_mlp_forward_windows(ctx: *mut rp_ctx, model: *const rp_model, mfcc: *const f32, S: usize, n_frames: usize, mfcc_size: c_int, precision: c_int, logits: *mut f32) -> c_int;
    pub fn rp_batch_detect_model(ctx: *mut rp_ctx, pcm: *const c_void, fmt: c_int, S: usize, n_samples: usize, pcm_stride: usize,
                                 model: *const rp_model, mfcc_size: c_int, none_index: c_int, config: *const rp_detector_config,
                                 precision: c_int, det: *mut rp_batch_detection, det_label: *mut i32, n_det: *mut i32, max_det: c_int) -> c_int;
    pub fn rp_synth_pcm_batch(ctx: *mut rp_ctx, seed: u64, first_stream: u64, S: usize, n_samples: usize, pcm_stride: usize, pcm: *mut f32) -> c_int;
    pub fn rp_ctx_timing_enable(ctx: *mut rp_ctx, enable: c_int) -> c_int;
    pub fn rp_ctx_timing_read(ctx: *mut rp_ctx, kernel: c_int, avg_ms: *mut f64, launches: *mut c_int) -> c_int;
    pub fn rp_ctx_timing_reset(ctx: *mut rp_ctx) -> c_int;
    pub fn rp_version() -> *const c_char;
}

// ---------------------------------------------------------------- RustpotterConfig (src/config.rs:10-219) -> rp_config
impl From<&AudioFmt> for rp_audio_fmt {
    fn from(f: &AudioFmt) -> Self {
        rp_audio_fmt {
            sample_rate: f.sample_rate,
            sample_format: match f.sample_format { SampleFormat::I8 => RP_SAMPLE_I8, SampleFormat::I16 => RP_SAMPLE_I16, SampleFormat::I32 => RP_SAMPLE_I32, SampleFormat::F32 => RP_SAMPLE_F32 },
            channels: f.channels,
            endianness: match f.endianness { Endianness::Big => RP_ENDIAN_BIG, Endianness::Little => RP_ENDIAN_LITTLE, Endianness::Native => RP_ENDIAN_NATIVE },
        }
    }
}
pub fn score_mode_to_c(m: ScoreMode) -> c_int {
    match m {
        ScoreMode::Average => RP_SCORE_AVERAGE, ScoreMode::Max => RP_SCORE_MAX, ScoreMode::Median => RP_SCORE_MEDIAN,
        ScoreMode::P25 => RP_SCORE_P25, ScoreMode::P50 => RP_SCORE_P50, ScoreMode::P75 => RP_SCORE_P75,
        ScoreMode::P80 => RP_SCORE_P80, ScoreMode::P90 => RP_SCORE_P90, ScoreMode::P95 => RP_SCORE_P95,
    }
}
impl From<&DetectorConfig> for rp_detector_config {
    /// `record_path` (feature `record`) has no counterpart: recording detections to wav files is outside the scoring path.
    fn from(d: &DetectorConfig) -> Self {
        rp_detector_config {
            avg_threshold: d.avg_threshold, threshold: d.threshold, min_scores: d.min_scores, eager: d.eager, score_ref: d.score_ref,
            band_size: d.band_size, score_mode: score_mode_to_c(d.score_mode),
            vad_mode: match d.vad_mode { None => RP_VAD_NONE, Some(VADMode::Easy) => RP_VAD_EASY, Some(VADMode::Medium) => RP_VAD_MEDIUM, Some(VADMode::Hard) => RP_VAD_HARD },
        }
    }
}
impl From<&GainNormalizationConfig> for rp_gain_normalization_config {
    fn from(g: &GainNormalizationConfig) -> Self {
        rp_gain_normalization_config { enabled: g.enabled, has_gain_ref: g.gain_ref.is_some(), gain_ref: g.gain_ref.unwrap_or(0.0), min_gain: g.min_gain, max_gain: g.max_gain }
    }
}
impl From<&BandPassConfig> for rp_band_pass_config {
    fn from(b: &BandPassConfig) -> Self { rp_band_pass_config { enabled: b.enabled, low_cutoff: b.low_cutoff, high_cutoff: b.high_cutoff } }
}
impl From<&FiltersConfig> for rp_filters_config {
    fn from(f: &FiltersConfig) -> Self { rp_filters_config { gain_normalizer: (&f.gain_normalizer).into(), band_pass: (&f.band_pass).into() } }
}
impl From<&RustpotterConfig> for rp_config {
    fn from(c: &RustpotterConfig) -> Self { rp_config { fmt: (&c.fmt).into(), detector: (&c.detector).into(), filters: (&c.filters).into() } }
}

// --------------------------------------------------------------------------- the drop-in `Rustpotter` (src/detector.rs)
/// Same fields as the reference's `RustpotterDetection` (src/detector.rs:488-501).
pub struct RustpotterDetection { pub name: String, pub avg_score: f32, pub score: f32, pub scores: HashMap<String, f32>, pub counter: usize, pub gain: f32 }

/// `Sample` trait of the reference (src/audio/audio_types.rs:59-68) narrowed to the FFI entry point per type.
pub trait Sample: Copy { unsafe fn process(d: *mut rp_detector, s: &[Self], out: *mut rp_detection) -> c_int; }
impl Sample for i8 { unsafe fn process(d: *mut rp_detector, s: &[i8], o: *mut rp_detection) -> c_int { rp_process_samples_i8(d, s.as_ptr(), s.len(), o) } }
impl Sample for i16 { unsafe fn process(d: *mut rp_detector, s: &[i16], o: *mut rp_detection) -> c_int { rp_process_samples_i16(d, s.as_ptr(), s.len(), o) } }
impl Sample for i32 { unsafe fn process(d: *mut rp_detector, s: &[i32], o: *mut rp_detection) -> c_int { rp_process_samples_i32(d, s.as_ptr(), s.len(), o) } }
impl Sample for f32 { unsafe fn process(d: *mut rp_detector, s: &[f32], o: *mut rp_detection) -> c_int { rp_process_samples_f32(d, s.as_ptr(), s.len(), o) } }

fn last_error() -> String { unsafe { CStr::from_ptr(rp_last_error()).to_string_lossy().into_owned() } }
fn status(r: c_int) -> Result<(), String> { if r < 0 { Err(last_error()) } else { Ok(()) } }
unsafe fn owned(d: &rp_detection) -> RustpotterDetection {
    let mut scores = HashMap::new();
    for i in 0..d.n_scores { scores.insert(CStr::from_ptr(*d.score_names.add(i)).to_string_lossy().into_owned(), *d.scores.add(i)); }
    RustpotterDetection { name: CStr::from_ptr(d.name).to_string_lossy().into_owned(), avg_score: d.avg_score, score: d.score, scores, counter: d.counter, gain: d.gain }
}

/// Drop-in for `rustpotter::Rustpotter`: identical method names, argument types and return types.
/// `partial` mirrors the handle's partial detection so that `get_partial_detection` can hand out a reference like the
/// reference does (`Option<&RustpotterDetection>`, src/detector.rs:212); it is refreshed by every `&mut self` method that can change it.
pub struct Rustpotter { h: *mut rp_detector, partial: Option<RustpotterDetection> }
unsafe impl Send for Rustpotter {}   // the reference's detector is Send, never Sync (all methods take &mut self)
impl Drop for Rustpotter { fn drop(&mut self) { unsafe { rp_free(self.h) } } }
impl Rustpotter {
    /// src/detector.rs:95
    pub fn new(config: &RustpotterConfig) -> Result<Rustpotter, String> {
        let c: rp_config = config.into();
        let mut h = std::ptr::null_mut();
        if unsafe { rp_new(&c, &mut h) } < 0 { Err(last_error()) } else { Ok(Rustpotter { h, partial: None }) }
    }
    fn refresh_partial(&mut self) {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        self.partial = unsafe { if rp_get_partial_detection(self.h, d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } };
    }
    /// src/detector.rs:144 -- the in-memory struct travels as its own `.rpw` serialisation (WakewordSave::save_to_buffer)
    pub fn add_wakeword_ref(&mut self, key: &str, wakeword: WakewordRef) -> Result<(), String> {
        self.add_wakeword_from_buffer(key, &wakeword.save_to_buffer()?)
    }
    /// src/detector.rs:148
    pub fn add_wakeword_model(&mut self, key: &str, wakeword: WakewordModel) -> Result<(), String> {
        self.add_wakeword_from_buffer(key, &wakeword.save_to_buffer()?)
    }
    /// src/detector.rs:152
    pub fn add_wakeword_from_buffer(&mut self, key: &str, buffer: &[u8]) -> Result<(), String> {
        let k = CString::new(key).map_err(|e| e.to_string())?;
        let r = status(unsafe { rp_add_wakeword_from_buffer(self.h, k.as_ptr(), buffer.as_ptr(), buffer.len()) });
        self.refresh_partial();
        r
    }
    /// src/detector.rs:165
    pub fn add_wakeword_from_file(&mut self, key: &str, path: &str) -> Result<(), String> {
        let (k, p) = (CString::new(key).map_err(|e| e.to_string())?, CString::new(path).map_err(|e| e.to_string())?);
        let r = status(unsafe { rp_add_wakeword_from_file(self.h, k.as_ptr(), p.as_ptr()) });
        self.refresh_partial();
        r
    }
    pub fn remove_wakeword(&mut self, key: &str) -> bool { CString::new(key).map(|k| unsafe { rp_remove_wakeword(self.h, k.as_ptr()) }).unwrap_or(false) }
    pub fn remove_wakewords(&mut self) -> bool { unsafe { rp_remove_wakewords(self.h) } }
    pub fn get_samples_per_frame(&self) -> usize { unsafe { rp_get_samples_per_frame(self.h) } }
    pub fn get_bytes_per_frame(&self) -> usize { unsafe { rp_get_bytes_per_frame(self.h) } }
    pub fn get_partial_detection(&self) -> Option<&RustpotterDetection> { self.partial.as_ref() }
    pub fn get_rms_level(&self) -> f32 { unsafe { rp_get_rms_level(self.h) } }
    pub fn get_gain(&self) -> f32 { unsafe { rp_get_gain(self.h) } }
    pub fn get_rms_level_ref(&self) -> f32 { unsafe { rp_get_rms_level_ref(self.h) } }
    /// src/detector.rs:234
    pub fn process_bytes(&mut self, audio_bytes: &[u8]) -> Option<RustpotterDetection> {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        let r = unsafe { if rp_process_bytes(self.h, audio_bytes.as_ptr(), audio_bytes.len(), d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } };
        self.refresh_partial();
        r
    }
    /// src/detector.rs:245
    pub fn process_samples<T: Sample>(&mut self, audio_samples: Vec<T>) -> Option<RustpotterDetection> {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        let r = unsafe { if T::process(self.h, &audio_samples, d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } };
        self.refresh_partial();
        r
    }
    /// src/detector.rs:257
    pub fn update_config(&mut self, config: &RustpotterConfig) { let c: rp_config = config.into(); unsafe { rp_update_config(self.h, &c); } self.refresh_partial(); }
    /// src/detector.rs:263
    pub fn update_detector_config(&mut self, config: &DetectorConfig) { let c: rp_detector_config = config.into(); unsafe { rp_update_detector_config(self.h, &c); } self.refresh_partial(); }
    /// src/detector.rs:283
    pub fn update_filters_config(&mut self, config: &FiltersConfig) { let c: rp_filters_config = config.into(); unsafe { rp_update_filters_config(self.h, &c); } self.refresh_partial(); }
    /// src/detector.rs:290
    pub fn reset(&mut self) { unsafe { rp_reset(self.h) } self.partial = None; }
}

/// RustpotterConfig::default() as the library sees it (src/config.rs:20-29,43-52,63-71,192-207)
pub fn default_c_config() -> rp_config { let mut c = std::mem::MaybeUninit::<rp_config>::uninit(); unsafe { rp_config_default(c.as_mut_ptr()); c.assume_init() } }
pub fn version() -> String { unsafe { CStr::from_ptr(rp_version()).to_string_lossy().into_owned() } }
/// target architecture and non-default compiler flags of the loaded library ("gfx950" for the product build)
pub fn build_info() -> String { unsafe { CStr::from_ptr(rp_build_info()).to_string_lossy().into_owned() } }
/// how this thread's last `batch_detect_sharded` gathered its shards (peer access over xGMI or staged copies)
pub fn sharded_gather_info() -> String { unsafe { CStr::from_ptr(rp_sharded_gather_info()).to_string_lossy().into_owned() } }
/// frames `MfccExtractor::compute` yields for n_samples fed in 480-sample chunks (src/mfcc/extractor.rs:60-79)
pub fn mfcc_num_frames(n_samples: usize) -> usize { unsafe { rp_mfcc_num_frames(n_samples) } }
/// `AudioEncoder::get_input_frame_length()` and the 16 kHz samples one input frame yields (src/audio/encoder.rs:63-83)
pub fn resampler_frame_lengths(sample_rate: usize) -> Result<(usize, usize), String> {
    let (mut i, mut o) = (0usize, 0usize);
    status(unsafe { rp_resampler_frame_lengths(sample_rate, &mut i, &mut o) })?;
    Ok((i, o))
}

// ------------------------------------------------------------------ the operator seam: S independent streams per call
/// A wakeword reference on the device (`WakewordComparator`'s copy of the templates, src/wakewords/comp/wakeword_comp.rs:55-63).
pub struct Templates { h: *mut rp_templates, pub n_templates: usize, pub mfcc_size: usize }
impl Drop for Templates { fn drop(&mut self) { unsafe { rp_templates_free(self.h) } } }
impl Templates {
    pub fn max_len(&self) -> usize { unsafe { rp_templates_max_len(self.h) as usize } }
}
/// A wakeword model on the device (`WakewordNN`, src/wakewords/nn/wakeword_nn.rs:17-37).
pub struct Model { h: *mut rp_model, pub dims: Vec<c_int> }
impl Drop for Model { fn drop(&mut self) { unsafe { rp_model_free(self.h) } } }

/// scores [S][n_win][T], avg [S][n_win] (empty without an averaged template), agg [S][n_win]
pub struct WindowScores { pub n_win: usize, pub scores: Vec<f32>, pub avg: Vec<f32>, pub agg: Vec<f32> }
/// detections of every stream, in stream order
pub type Detections = Vec<Vec<rp_batch_detection>>;

fn split_detections(det: Vec<rp_batch_detection>, n_det: Vec<i32>, max_det: usize) -> Detections {
    (0..n_det.len()).map(|s| det[s * max_det..s * max_det + (n_det[s].max(0) as usize).min(max_det)].to_vec()).collect()
}

/// One device + one HIP stream; arrays are host slices (RP_CTX_HOST_POINTERS: the library stages them through its own
/// device buffers).  A service that keeps its audio on the device creates the context with RP_CTX_DEVICE_POINTERS and
/// calls the `rp_*` functions with device pointers directly.  `Send`, not `Sync` -- one call at a time per context.
pub struct HipContext { h: *mut rp_ctx }
unsafe impl Send for HipContext {}
impl Drop for HipContext { fn drop(&mut self) { unsafe { rp_ctx_free(self.h) } } }
impl HipContext {
    pub fn new(device: c_int) -> Result<HipContext, String> { HipContext::with_flags(device, RP_CTX_HOST_POINTERS) }
    pub fn with_flags(device: c_int, flags: c_int) -> Result<HipContext, String> {
        let mut h = std::ptr::null_mut();
        status(unsafe { rp_ctx_new(device, flags, &mut h) })?;
        Ok(HipContext { h })
    }
    pub fn raw(&self) -> *mut rp_ctx { self.h }
    /// run the launches on a caller-owned hipStream_t (NULL = the context's own)
    pub fn set_stream(&self, hip_stream: *mut c_void) -> Result<(), String> { status(unsafe { rp_ctx_set_stream(self.h, hip_stream) }) }
    pub fn synchronize(&self) -> Result<(), String> { status(unsafe { rp_ctx_synchronize(self.h) }) }
    /// kernel(s) and operand format of the last dense-row wakeword-model forward
    pub fn last_mlp_kernel(&self) -> String {
        let p = unsafe { rp_ctx_last_mlp_kernel(self.h) };
        if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
    }
    /// mask of `RP_DTW_KERNEL_*` | `RP_DTW_PRODUCTS_*`: the DTW kernel families launched since the last call and their product arithmetic
    pub fn dtw_kernels(&self) -> u32 { unsafe { rp_ctx_dtw_kernels(self.h) as u32 } }
    /// `RP_ARITH_*` for the calls that follow (the reference has one arithmetic, f32: `src/mfcc/comparator.rs:28-48`)
    pub fn set_arithmetic(&self, arith: c_int, ragged_matrix: bool) -> Result<(), String> {
        status(unsafe { rp_ctx_set_arithmetic(self.h, arith, ragged_matrix as c_int) })
    }
    /// the current `RP_ARITH_*` and the ragged-matrix flag
    pub fn arithmetic(&self) -> (c_int, bool) {
        let mut r: c_int = 0;
        let a = unsafe { rp_ctx_arithmetic(self.h, &mut r) };
        (a, r != 0)
    }
    /// (window, templates) pairs rescored with the reference-shaped cosine (`sqrt(dot_a * dot_b)`, src/mfcc/comparator.rs:28-48) so far
    pub fn dtw_ref_pairs(&self) -> Result<u64, String> {
        let mut v: u64 = 0;
        status(unsafe { rp_ctx_dtw_ref_pairs(self.h, &mut v) })?;
        Ok(v)
    }

    /// `MfccExtractor::compute` (src/mfcc/extractor.rs:60-163) over `n_streams` whole streams of `n_samples` f32 samples:
    /// returns `[n_streams][mfcc_num_frames(n_samples)][mfcc_size]`.
    pub fn mfcc_batch(&self, pcm: &[f32], n_streams: usize, n_samples: usize, mfcc_size: u16) -> Result<Vec<f32>, String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let mut out = vec![0f32; n_streams * mfcc_num_frames(n_samples) * mfcc_size as usize];
        status(unsafe { rp_mfcc_batch(self.h, pcm.as_ptr(), n_streams, n_samples, n_samples, mfcc_size as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// the same with the samples in one of the reference's `Sample` formats (`v as f32 / T::MAX as f32` inside the kernel)
    pub fn mfcc_batch_fmt<T: Sample>(&self, pcm: &[T], fmt: SampleFormat, n_streams: usize, n_samples: usize, mfcc_size: u16) -> Result<Vec<f32>, String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let f = rp_audio_fmt::from(&AudioFmt { sample_rate: 16000, sample_format: fmt, channels: 1, endianness: Endianness::Native }).sample_format;
        let mut out = vec![0f32; n_streams * mfcc_num_frames(n_samples) * mfcc_size as usize];
        status(unsafe { rp_mfcc_batch_fmt(self.h, pcm.as_ptr() as *const c_void, f, n_streams, n_samples, n_samples, mfcc_size as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// Upload a wakeword reference: `templates[t]` = `[len_t][K]` as `WakewordRef::samples_features` holds them.
    pub fn templates(&self, templates: &[Vec<Vec<f32>>], avg: Option<&Vec<Vec<f32>>>) -> Result<Templates, String> {
        let k = templates.first().and_then(|t| t.first()).map_or(0, |f| f.len());
        let lens: Vec<c_int> = templates.iter().map(|t| t.len() as c_int).collect();
        let feats: Vec<f32> = templates.iter().flat_map(|t| t.iter().flatten().copied()).collect();
        let avg_flat: Vec<f32> = avg.map_or(Vec::new(), |a| a.iter().flatten().copied().collect());
        let mut h = std::ptr::null_mut();
        status(unsafe {
            rp_templates_new(self.h, lens.len() as c_int, k as c_int, lens.as_ptr(), feats.as_ptr(), avg.map_or(0, |a| a.len() as c_int),
                             if avg.is_some() { avg_flat.as_ptr() } else { std::ptr::null() }, &mut h)
        })?;
        Ok(Templates { h, n_templates: lens.len(), mfcc_size: k })
    }
    pub fn templates_from_ref(&self, w: &WakewordRef) -> Result<Templates, String> {
        let t: Vec<Vec<Vec<f32>>> = w.samples_features.values().cloned().collect();
        self.templates(&t, w.avg_features.as_ref())
    }
    /// `WakewordDetector::run_detection` scoring (src/wakewords/comp/wakeword_comp.rs:77-139) for every window start of
    /// every stream: per-template scores, the averaged-template score and the `score_mode` aggregate.
    pub fn dtw_score_batch(&self, mfcc: &[f32], n_streams: usize, n_frames: usize, t: &Templates, score_ref: f32, band_size: u16,
                           score_mode: ScoreMode, with_avg: bool) -> Result<WindowScores, String> {
        assert!(mfcc.len() >= n_streams * n_frames * t.mfcc_size);
        let n_win = (n_frames + 1).saturating_sub(t.max_len());
        let mut r = WindowScores { n_win, scores: vec![0f32; n_streams * n_win * t.n_templates], avg: vec![0f32; if with_avg { n_streams * n_win } else { 0 }],
                                   agg: vec![0f32; n_streams * n_win] };
        status(unsafe {
            rp_dtw_score_batch(self.h, mfcc.as_ptr(), n_streams, n_frames, t.h, score_ref, band_size as c_int, score_mode_to_c(score_mode),
                               with_avg as c_int, r.scores.as_mut_ptr(), if with_avg { r.avg.as_mut_ptr() } else { std::ptr::null_mut() }, r.agg.as_mut_ptr())
        })?;
        Ok(r)
    }
    /// `Rustpotter::process_new_mfccs` / `run_detection` / `reset` (src/detector.rs:290-302,377-454) over precomputed window scores.
    pub fn detect_scan(&self, w: &WindowScores, n_streams: usize, n_frames: usize, max_len: usize, config: &DetectorConfig,
                       mfcc: Option<(&[f32], u16)>, max_det: usize) -> Result<Detections, String> {
        let c: rp_detector_config = config.into();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        let (mp, k) = mfcc.map_or((std::ptr::null(), 0), |(m, k)| (m.as_ptr(), k as c_int));
        status(unsafe {
            rp_detect_scan(self.h, w.agg.as_ptr(), if w.avg.is_empty() { std::ptr::null() } else { w.avg.as_ptr() }, n_streams, n_frames,
                           max_len as c_int, &c, !w.avg.is_empty() as c_int, mp, k, det.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int)
        })?;
        Ok(split_detections(det, n_det, max_det))
    }
    /// The whole path for `n_streams` streams in one call (= one `Rustpotter` per stream fed chunk by chunk).
    pub fn batch_detect(&self, pcm: &[f32], n_streams: usize, n_samples: usize, t: &Templates, config: &DetectorConfig, max_det: usize) -> Result<Detections, String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let c: rp_detector_config = config.into();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        status(unsafe {
            rp_batch_detect(self.h, pcm.as_ptr(), n_streams, n_samples, n_samples, t.h, &c, det.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int,
                            std::ptr::null_mut(), std::ptr::null_mut())
        })?;
        Ok(split_detections(det, n_det, max_det))
    }
    pub fn batch_detect_i16(&self, pcm: &[i16], n_streams: usize, n_samples: usize, t: &Templates, config: &DetectorConfig, max_det: usize) -> Result<Detections, String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let c: rp_detector_config = config.into();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        status(unsafe {
            rp_batch_detect_fmt(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_I16, n_streams, n_samples, n_samples, t.h, &c, det.as_mut_ptr(),
                                n_det.as_mut_ptr(), max_det as c_int, std::ptr::null_mut(), std::ptr::null_mut())
        })?;
        Ok(split_detections(det, n_det, max_det))
    }
    /// `batch_detect_i16` for streams in HOST memory, pipelined: blocks of `block_streams` streams (0 = 8 192), the next block's copy
    /// under this block's kernels (the copies overlap when `pcm` is page-locked).  Returns the detections and the call's wall seconds.
    pub fn batch_detect_ingest_i16(&self, pcm: &[i16], n_streams: usize, n_samples: usize, t: &Templates, config: &DetectorConfig, max_det: usize,
                                   block_streams: usize) -> Result<(Detections, f64), String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let c: rp_detector_config = config.into();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        let mut seconds = 0f64;
        status(unsafe {
            rp_batch_detect_ingest(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_I16, n_streams, n_samples, n_samples, t.h, &c, det.as_mut_ptr(),
                                   n_det.as_mut_ptr(), max_det as c_int, block_streams, &mut seconds)
        })?;
        Ok((split_detections(det, n_det, max_det), seconds))
    }
    /// A detector holding several wakewords (`run_wakeword_detectors`, src/detector.rs:433-447); `overrides[w]` = the wakeword's own
    /// `(threshold, avg_threshold)` options.  Returns the detections and, per detection, the index of the wakeword that fired.
    pub fn batch_detect_multi(&self, pcm: &[f32], n_streams: usize, n_samples: usize, wakewords: &[&Templates], overrides: &[(Option<f32>, Option<f32>)],
                              config: &DetectorConfig, max_det: usize) -> Result<(Detections, Vec<Vec<i32>>), String> {
        assert!(pcm.len() >= n_streams * n_samples && overrides.len() == wakewords.len());
        let c: rp_detector_config = config.into();
        let ts: Vec<*const rp_templates> = wakewords.iter().map(|t| t.h as *const rp_templates).collect();
        let thr: Vec<f32> = overrides.iter().map(|o| o.0.unwrap_or(f32::NAN)).collect();
        let athr: Vec<f32> = overrides.iter().map(|o| o.1.unwrap_or(f32::NAN)).collect();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut which = vec![0i32; n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        status(unsafe {
            rp_batch_detect_multi(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, n_streams, n_samples, n_samples, ts.len(), ts.as_ptr(), &c,
                                  thr.as_ptr(), athr.as_ptr(), det.as_mut_ptr(), which.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int)
        })?;
        let w = (0..n_streams).map(|s| which[s * max_det..s * max_det + (n_det[s].max(0) as usize).min(max_det)].to_vec()).collect();
        Ok((split_detections(det, n_det, max_det), w))
    }
    /// Upload a wakeword model: `weights[l]` = `[dims[l+1]][dims[l]]` row-major (candle `Linear`), `biases[l]` = `[dims[l+1]]`.
    pub fn model(&self, dims: &[c_int], weights: &[Vec<f32>], biases: &[Vec<f32>]) -> Result<Model, String> {
        assert!(dims.len() == weights.len() + 1 && weights.len() == biases.len());
        let wp: Vec<*const f32> = weights.iter().map(|w| w.as_ptr()).collect();
        let bp: Vec<*const f32> = biases.iter().map(|b| b.as_ptr()).collect();
        let mut h = std::ptr::null_mut();
        status(unsafe { rp_model_new(self.h, weights.len() as c_int, dims.as_ptr(), wp.as_ptr(), bp.as_ptr(), &mut h) })?;
        Ok(Model { h, dims: dims.to_vec() })
    }
    /// `ModelImpl::forward` (src/wakewords/nn/wakeword_nn.rs:101-106,305-389): x `[rows][dims[0]]` -> logits `[rows][labels]`.
    pub fn mlp_forward_batch(&self, m: &Model, x: &[f32], rows: usize, bf16: bool) -> Result<Vec<f32>, String> {
        assert!(x.len() >= rows * m.dims[0] as usize);
        let mut out = vec![0f32; rows * *m.dims.last().unwrap() as usize];
        status(unsafe { rp_mlp_forward_batch(self.h, m.h, x.as_ptr(), rows, if bf16 { RP_MLP_BF16 } else { RP_MLP_F32 }, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// The forward over every window of `train_size = dims[0] / mfcc_size` frames of each stream's MFCC rows (what `WakewordNN::run_detection`
    /// computes frame after frame, src/wakewords/nn/wakeword_nn.rs:101-159): mfcc `[streams][n_frames][mfcc_size]` -> logits `[streams][n_win][labels]`.
    pub fn mlp_forward_windows(&self, m: &Model, mfcc: &[f32], n_streams: usize, n_frames: usize, mfcc_size: u16) -> Result<Vec<f32>, String> {
        assert!(mfcc.len() >= n_streams * n_frames * mfcc_size as usize);
        let train_size = m.dims[0] as usize / mfcc_size as usize;
        let n_win = if n_frames >= train_size { n_frames - train_size + 1 } else { 0 };
        let mut out = vec![0f32; n_streams * n_win * *m.dims.last().unwrap() as usize];
        status(unsafe { rp_mlp_forward_windows(self.h, m.h, mfcc.as_ptr(), n_streams, n_frames, mfcc_size as c_int, RP_MLP_F32, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `WakewordNN::run_detection` inside the detection state machine, for whole streams; returns detections and their label indices.
    pub fn batch_detect_model(&self, pcm: &[f32], n_streams: usize, n_samples: usize, m: &Model, mfcc_size: u16, none_index: Option<usize>,
                              config: &DetectorConfig, bf16: bool, max_det: usize) -> Result<(Detections, Vec<Vec<i32>>), String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let c: rp_detector_config = config.into();
        let mut det = vec![rp_batch_detection::default(); n_streams * max_det];
        let mut label = vec![0i32; n_streams * max_det];
        let mut n_det = vec![0i32; n_streams];
        status(unsafe {
            rp_batch_detect_model(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, n_streams, n_samples, n_samples, m.h, mfcc_size as c_int,
                                  none_index.map_or(-1, |i| i as c_int), &c, if bf16 { RP_MLP_BF16 } else { RP_MLP_F32 }, det.as_mut_ptr(),
                                  label.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int)
        })?;
        let l = (0..n_streams).map(|s| label[s * max_det..s * max_det + (n_det[s].max(0) as usize).min(max_det)].to_vec()).collect();
        Ok((split_detections(det, n_det, max_det), l))
    }
    /// The audio front-end of `Rustpotter::process_audio` (gain normaliser + band-pass, src/detector.rs:358-371) for whole streams:
    /// returns (filtered audio, per-chunk rms levels, per-chunk gains).
    pub fn frontend_batch(&self, pcm: &[f32], n_streams: usize, n_samples: usize, filters: &FiltersConfig, rms_level_ref: f32,
                          window_size: usize) -> Result<(Vec<f32>, Vec<f32>, Vec<f32>), String> {
        assert!(pcm.len() >= n_streams * n_samples);
        let f: rp_filters_config = filters.into();
        let nch = n_samples / 480;
        let (mut out, mut rms, mut gains) = (vec![0f32; n_streams * n_samples], vec![0f32; n_streams * nch], vec![0f32; n_streams * nch]);
        status(unsafe {
            rp_frontend_batch(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, n_streams, n_samples, n_samples, &f, rms_level_ref, window_size,
                              out.as_mut_ptr(), n_samples, rms.as_mut_ptr(), gains.as_mut_ptr())
        })?;
        Ok((out, rms, gains))
    }
    /// `AudioEncoder::reencode_to_mono_with_sample_rate` (src/audio/encoder.rs:41-60) for whole streams: f32 input at `sample_rate`
    /// with `channels` interleaved channels -> 16 kHz mono.
    pub fn resample_batch(&self, pcm: &[f32], channels: u16, sample_rate: usize, n_streams: usize, n_samples: usize) -> Result<Vec<f32>, String> {
        assert!(pcm.len() >= n_streams * n_samples * channels as usize);
        let (fi, fo) = resampler_frame_lengths(sample_rate)?;
        let per = (n_samples / fi) * fo;
        let mut out = vec![0f32; n_streams * per];
        status(unsafe {
            rp_resample_batch(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, channels as c_int, sample_rate, n_streams, n_samples,
                              n_samples * channels as usize, out.as_mut_ptr(), per)
        })?;
        Ok(out)
    }
    /// the benchmark's synthetic input
    pub fn synth_pcm_batch(&self, seed: u64, first_stream: u64, n_streams: usize, n_samples: usize) -> Result<Vec<f32>, String> {
        let mut out = vec![0f32; n_streams * n_samples];
        status(unsafe { rp_synth_pcm_batch(self.h, seed, first_stream, n_streams, n_samples, n_samples, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// per-kernel launch timing (0 mfcc, 1 dtw, 2 aggregate, 3 scan, 4 mlp, 5 resample): average ms and launches since the last reset
    pub fn timing_enable(&self, on: bool) -> Result<(), String> { status(unsafe { rp_ctx_timing_enable(self.h, on as c_int) }) }
    pub fn timing_reset(&self) -> Result<(), String> { status(unsafe { rp_ctx_timing_reset(self.h) }) }
    pub fn timing_read(&self, kernel: c_int) -> Result<(f64, c_int), String> {
        let (mut ms, mut n) = (0f64, 0 as c_int);
        status(unsafe { rp_ctx_timing_read(self.h, kernel, &mut ms, &mut n) })?;
        Ok((ms, n))
    }
}

/// S live streams that each receive a few 30 ms chunks per call: the batched form of S `Rustpotter` handles
/// (INTEGRATION.md §4).  Borrows the context and the templates, like the C handle does.
pub struct StreamBatch<'a> { h: *mut rp_stream_batch, n_streams: usize, _ctx: &'a HipContext, _t: std::marker::PhantomData<&'a Templates> }
/// One wakeword of a detector that holds several (`Rustpotter::add_wakeword_ref` / `add_wakeword_model`, src/detector.rs:144-150)
pub enum BatchWakeword<'a> {
    /// a reference with its own `threshold` / `avg_threshold` options (`WakewordRef::threshold`, `::avg_threshold`)
    Ref { templates: &'a Templates, threshold: Option<f32>, avg_threshold: Option<f32> },
    /// a model: index of the "none" label (if any) and whether layer 1 runs with bf16 inputs
    Model { model: &'a Model, none_index: Option<usize>, bf16: bool },
}
impl<'a> Drop for StreamBatch<'a> { fn drop(&mut self) { unsafe { rp_stream_batch_free(self.h) } } }
impl<'a> StreamBatch<'a> {
    pub fn new(ctx: &'a HipContext, t: &'a Templates, config: &DetectorConfig, n_streams: usize, max_chunks_per_call: usize) -> Result<StreamBatch<'a>, String> {
        let c: rp_detector_config = config.into();
        let mut h = std::ptr::null_mut();
        status(unsafe { rp_stream_batch_new(ctx.h, t.h, &c, n_streams, max_chunks_per_call, &mut h) })?;
        Ok(StreamBatch { h, n_streams, _ctx: ctx, _t: std::marker::PhantomData })
    }
    /// S detectors that each hold `wakewords` (references and / or models sharing `mfcc_size`): the best score of the wakewords
    /// whose own thresholds pass wins a frame (`run_wakeword_detectors`, src/detector.rs:433-447)
    pub fn new_multi(ctx: &'a HipContext, wakewords: &[BatchWakeword<'a>], mfcc_size: u16, config: &DetectorConfig, n_streams: usize,
                     max_chunks_per_call: usize) -> Result<StreamBatch<'a>, String> {
        let c: rp_detector_config = config.into();
        let specs: Vec<rp_wakeword_spec> = wakewords.iter().map(|w| match w {
            BatchWakeword::Ref { templates, threshold, avg_threshold } => rp_wakeword_spec {
                templates: templates.h as *const rp_templates, model: std::ptr::null(), none_index: -1, precision: RP_MLP_F32,
                threshold: threshold.unwrap_or(f32::NAN), avg_threshold: avg_threshold.unwrap_or(f32::NAN) },
            BatchWakeword::Model { model, none_index, bf16 } => rp_wakeword_spec {
                templates: std::ptr::null(), model: model.h as *const rp_model, none_index: none_index.map_or(-1, |i| i as c_int),
                precision: if *bf16 { RP_MLP_BF16 } else { RP_MLP_F32 }, threshold: f32::NAN, avg_threshold: f32::NAN },
        }).collect();
        let mut h = std::ptr::null_mut();
        status(unsafe { rp_stream_batch_new_multi(ctx.h, specs.len(), specs.as_ptr(), mfcc_size as c_int, &c, n_streams, max_chunks_per_call, &mut h) })?;
        Ok(StreamBatch { h, n_streams, _ctx: ctx, _t: std::marker::PhantomData })
    }
    /// `process` that also tells which wakeword fired and, for a model, which label: (detections, wakeword indices, label indices or -1)
    pub fn process_multi(&mut self, pcm: &[f32], n_chunks: usize, max_det: usize) -> Result<(Detections, Vec<Vec<i32>>, Vec<Vec<i32>>), String> {
        let stride = n_chunks * self.samples_per_chunk();
        assert!(pcm.len() >= self.n_streams * stride);
        let mut det = vec![rp_batch_detection::default(); self.n_streams * max_det];
        let (mut which, mut label) = (vec![0i32; self.n_streams * max_det], vec![0i32; self.n_streams * max_det]);
        let mut n_det = vec![0i32; self.n_streams];
        status(unsafe {
            rp_stream_batch_process_multi(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, n_chunks, stride, det.as_mut_ptr(), which.as_mut_ptr(),
                                          label.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int)
        })?;
        let cut = |v: &Vec<i32>| -> Vec<Vec<i32>> { (0..self.n_streams).map(|s| v[s * max_det..s * max_det + (n_det[s].max(0) as usize).min(max_det)].to_vec()).collect() };
        let (w, l) = (cut(&which), cut(&label));
        Ok((split_detections(det, n_det, max_det), w, l))
    }
    /// `RustpotterConfig.fmt` of the streams (sample rate, channels); before the first `process`
    pub fn set_input(&mut self, sample_rate: usize, channels: u16) -> Result<(), String> { status(unsafe { rp_stream_batch_set_input(self.h, sample_rate, channels as c_int) }) }
    pub fn samples_per_chunk(&self) -> usize { unsafe { rp_stream_batch_samples_per_chunk(self.h) } }
    pub fn chunks_seen(&self) -> usize { unsafe { rp_stream_batch_chunks_seen(self.h) } }
    /// `process_samples` once per chunk on every stream: pcm `[S][n_chunks * samples_per_chunk()]` f32
    pub fn process(&mut self, pcm: &[f32], n_chunks: usize, max_det: usize) -> Result<Detections, String> {
        let stride = n_chunks * self.samples_per_chunk();
        assert!(pcm.len() >= self.n_streams * stride);
        let mut det = vec![rp_batch_detection::default(); self.n_streams * max_det];
        let mut n_det = vec![0i32; self.n_streams];
        status(unsafe {
            rp_stream_batch_process(self.h, pcm.as_ptr() as *const c_void, RP_SAMPLE_F32, n_chunks, stride, det.as_mut_ptr(), n_det.as_mut_ptr(),
                                    max_det as c_int, std::ptr::null_mut())
        })?;
        Ok(split_detections(det, n_det, max_det))
    }
    /// `Rustpotter::reset` of one stream (`None` = all)
    pub fn reset(&mut self, stream: Option<usize>) -> Result<(), String> { status(unsafe { rp_stream_batch_reset(self.h, stream.map_or(-1, |s| s as i64)) }) }
}

// ---- multi-GPU: independent streams sharded over the GPUs of a node ----
/// All streams of `shards` (one Vec of equally long f32 streams per GPU, host memory) through the whole path on
/// `devices[g]`; returns the detections of every stream in shard order.  The wakeword is given as its templates
/// ([T] x [len][K], as `WakewordRef::samples_features` holds them) plus the optional averaged template.
pub fn batch_detect_sharded(devices: &[c_int], templates: &[Vec<Vec<f32>>], avg: Option<&Vec<Vec<f32>>>, config: &DetectorConfig,
                            shards: &[Vec<Vec<f32>>], max_det: usize) -> Result<Detections, String> {
    assert_eq!(devices.len(), shards.len());
    let c: rp_detector_config = config.into();
    let n_samples = shards.iter().flat_map(|s| s.iter()).map(|x| x.len()).next().unwrap_or(0);
    assert!(shards.iter().flat_map(|s| s.iter()).all(|x| x.len() == n_samples), "every stream must hold the same number of samples");
    let ctxs: Vec<HipContext> = devices.iter().map(|&d| HipContext::new(d)).collect::<Result<_, _>>()?;
    let tms: Vec<Templates> = ctxs.iter().map(|c| c.templates(templates, avg)).collect::<Result<_, _>>()?;
    let cp: Vec<*mut rp_ctx> = ctxs.iter().map(|c| c.h).collect();
    let tp: Vec<*const rp_templates> = tms.iter().map(|t| t.h as *const rp_templates).collect();
    let flat: Vec<Vec<f32>> = shards.iter().map(|s| s.iter().flatten().copied().collect()).collect();
    let ptrs: Vec<*const c_void> = flat.iter().map(|f| f.as_ptr() as *const c_void).collect();
    let counts: Vec<usize> = shards.iter().map(|s| s.len()).collect();
    let total: usize = counts.iter().sum();
    let mut det = vec![rp_batch_detection::default(); total * max_det];
    let mut n_det = vec![0i32; total];
    status(unsafe {
        rp_batch_detect_sharded(cp.as_ptr(), tp.as_ptr(), devices.len() as c_int, ptrs.as_ptr(), RP_SAMPLE_F32, counts.as_ptr(), n_samples,
                                n_samples, &c, det.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int)
    })?;
    Ok(split_detections(det, n_det, max_det))
}

// ---- offline tooling ---------------------------------------------------------------------------------------------
fn take_buffer(r: c_int, out: *mut u8, out_len: usize) -> Result<Vec<u8>, String> {
    let res = if r < 0 { Err(last_error()) } else { Ok(unsafe { std::slice::from_raw_parts(out, out_len) }.to_vec()) };
    if !out.is_null() { unsafe { rp_buffer_free(out) }; }
    res
}
/// `WakewordRefBuildFromBuffers::new_from_sample_buffers` + `WakewordSave::save_to_buffer`
/// (src/wakewords/comp/wakeword_ref_build.rs:9-41, src/wakewords/wakeword_file.rs:22-26): returns `.rpw` bytes.
pub fn wakeword_ref_from_sample_buffers(name: &str, threshold: Option<f32>, avg_threshold: Option<f32>,
                                        samples: &HashMap<String, Vec<u8>>, mfcc_size: u16) -> Result<Vec<u8>, String> {
    let cname = CString::new(name).map_err(|e| e.to_string())?;
    let names: Vec<CString> = samples.keys().map(|k| CString::new(k.as_str()).unwrap()).collect();
    let name_ptrs: Vec<*const c_char> = names.iter().map(|n| n.as_ptr()).collect();
    let bufs: Vec<*const u8> = samples.values().map(|v| v.as_ptr()).collect();
    let lens: Vec<usize> = samples.values().map(|v| v.len()).collect();
    let ctx = HipContext::with_flags(0, RP_CTX_DEVICE_POINTERS)?;
    let (mut out, mut out_len) = (std::ptr::null_mut(), 0usize);
    let r = unsafe {
        rp_wakeword_ref_build(ctx.h, cname.as_ptr(), threshold.as_ref().map_or(std::ptr::null(), |t| t as *const f32),
                              avg_threshold.as_ref().map_or(std::ptr::null(), |t| t as *const f32), names.len(),
                              name_ptrs.as_ptr(), bufs.as_ptr(), lens.as_ptr(), mfcc_size, 0, &mut out, &mut out_len)
    };
    take_buffer(r, out, out_len)
}
/// `WakewordModelTrain::train_from_buffers` + `save_to_buffer` (src/wakewords/nn/wakeword_model_train.rs:44-168):
/// `m_type` 0..3 = ModelType::Tiny..Large; returns (`.rpw` bytes, last loss, test accuracy).
pub fn wakeword_model_train_from_buffers(m_type: c_int, train: &HashMap<String, Vec<u8>>, test: &HashMap<String, Vec<u8>>,
                                         learning_rate: f64, epochs: usize, test_epochs: usize, mfcc_size: u16,
                                         wakeword_model: Option<&[u8]>) -> Result<(Vec<u8>, f32, f32), String> {
    fn pack(d: &HashMap<String, Vec<u8>>) -> (Vec<CString>, Vec<*const u8>, Vec<usize>) {
        (d.keys().map(|k| CString::new(k.as_str()).unwrap()).collect(), d.values().map(|v| v.as_ptr()).collect(), d.values().map(|v| v.len()).collect())
    }
    let (trn, trb, trl) = pack(train);
    let (ten, teb, tel) = pack(test);
    let trp: Vec<*const c_char> = trn.iter().map(|n| n.as_ptr()).collect();
    let tep: Vec<*const c_char> = ten.iter().map(|n| n.as_ptr()).collect();
    let opt = rp_train_options { m_type, learning_rate: learning_rate as f32, epochs, test_epochs, mfcc_size, seed: 1 };
    let ctx = HipContext::with_flags(0, RP_CTX_DEVICE_POINTERS)?;
    let (mut out, mut out_len, mut loss, mut acc) = (std::ptr::null_mut(), 0usize, 0f32, 0f32);
    let (pm, pl) = wakeword_model.map_or((std::ptr::null(), 0), |m| (m.as_ptr(), m.len()));
    let r = unsafe {
        rp_wakeword_model_train(ctx.h, &opt, trp.len(), trp.as_ptr(), trb.as_ptr(), trl.as_ptr(), tep.len(), tep.as_ptr(), teb.as_ptr(),
                                tel.as_ptr(), pm, pl, &mut out, &mut out_len, &mut loss, &mut acc)
    };
    take_buffer(r, out, out_len).map(|b| (b, loss, acc))
}
