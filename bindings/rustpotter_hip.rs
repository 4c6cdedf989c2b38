//! rustpotter_hip.rs -- Rust binding of include/rustpotter_hip.h.
//!
//! NOT COMPILED in the authoring image (no rustc/cargo there).  It is the
//! source a rustpotter maintainer drops into the crate (e.g. `src/hip.rs`, behind a cargo
//! feature `hip`) so that `Rustpotter` keeps its public method names while every 30 ms chunk
//! is scored by librustpotter_hip.so.  build.rs: `println!("cargo:rustc-link-lib=dylib=rustpotter_hip");`
//!
//! Replaces: src/detector.rs:95-302 (public methods), src/mfcc/extractor.rs (MfccExtractor::compute),
//! src/wakewords/wakeword_detector.rs:3-14 (WakewordDetector::run_detection).
#![allow(non_camel_case_types)]
use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};

#[repr(C)] #[derive(Clone, Copy)] pub struct rp_audio_fmt { pub sample_rate: usize, pub sample_format: c_int, pub channels: u16, pub endianness: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_detector_config { pub avg_threshold: f32, pub threshold: f32, pub min_scores: usize, pub eager: bool, pub score_ref: f32, pub band_size: u16, pub score_mode: c_int, pub vad_mode: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_gain_normalization_config { pub enabled: bool, pub has_gain_ref: bool, pub gain_ref: f32, pub min_gain: f32, pub max_gain: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_band_pass_config { pub enabled: bool, pub low_cutoff: f32, pub high_cutoff: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_filters_config { pub gain_normalizer: rp_gain_normalization_config, pub band_pass: rp_band_pass_config }
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_config { pub fmt: rp_audio_fmt, pub detector: rp_detector_config, pub filters: rp_filters_config }
#[repr(C)] pub struct rp_detection { pub name: *const c_char, pub avg_score: f32, pub score: f32, pub n_scores: usize, pub score_names: *const *const c_char, pub scores: *const f32, pub counter: usize, pub gain: f32 }
pub enum rp_detector {}

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
    pub fn rp_process_bytes(d: *mut rp_detector, bytes: *const u8, len: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i8(d: *mut rp_detector, s: *const i8, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i16(d: *mut rp_detector, s: *const i16, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_i32(d: *mut rp_detector, s: *const i32, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_process_samples_f32(d: *mut rp_detector, s: *const f32, n: usize, out: *mut rp_detection) -> c_int;
    pub fn rp_update_config(d: *mut rp_detector, c: *const rp_config) -> c_int;
    pub fn rp_update_detector_config(d: *mut rp_detector, c: *const rp_detector_config) -> c_int;
    pub fn rp_update_filters_config(d: *mut rp_detector, c: *const rp_filters_config) -> c_int;
    pub fn rp_reset(d: *mut rp_detector);
    pub fn rp_last_error() -> *const c_char;
}

/// Same fields as the reference's `RustpotterDetection` (src/detector.rs:488-501).
pub struct RustpotterDetection { pub name: String, pub avg_score: f32, pub score: f32, pub scores: HashMap<String, f32>, pub counter: usize, pub gain: f32 }

/// `Sample` trait of the reference (src/audio/audio_types.rs:59-68) narrowed to the FFI entry point per type.
pub trait Sample: Copy { unsafe fn process(d: *mut rp_detector, s: &[Self], out: *mut rp_detection) -> c_int; }
impl Sample for i8 { unsafe fn process(d: *mut rp_detector, s: &[i8], o: *mut rp_detection) -> c_int { rp_process_samples_i8(d, s.as_ptr(), s.len(), o) } }
impl Sample for i16 { unsafe fn process(d: *mut rp_detector, s: &[i16], o: *mut rp_detection) -> c_int { rp_process_samples_i16(d, s.as_ptr(), s.len(), o) } }
impl Sample for i32 { unsafe fn process(d: *mut rp_detector, s: &[i32], o: *mut rp_detection) -> c_int { rp_process_samples_i32(d, s.as_ptr(), s.len(), o) } }
impl Sample for f32 { unsafe fn process(d: *mut rp_detector, s: &[f32], o: *mut rp_detection) -> c_int { rp_process_samples_f32(d, s.as_ptr(), s.len(), o) } }

fn last_error() -> String { unsafe { CStr::from_ptr(rp_last_error()).to_string_lossy().into_owned() } }
unsafe fn owned(d: &rp_detection) -> RustpotterDetection {
    let mut scores = HashMap::new();
    for i in 0..d.n_scores { scores.insert(CStr::from_ptr(*d.score_names.add(i)).to_string_lossy().into_owned(), *d.scores.add(i)); }
    RustpotterDetection { name: CStr::from_ptr(d.name).to_string_lossy().into_owned(), avg_score: d.avg_score, score: d.score, scores, counter: d.counter, gain: d.gain }
}

/// Drop-in for `rustpotter::Rustpotter`: identical method names and return types.
pub struct Rustpotter { h: *mut rp_detector }
unsafe impl Send for Rustpotter {}   // the reference's detector is Send, never Sync (all methods take &mut self)
impl Drop for Rustpotter { fn drop(&mut self) { unsafe { rp_free(self.h) } } }
impl Rustpotter {
    pub fn new(config: &rp_config) -> Result<Rustpotter, String> {
        let mut h = std::ptr::null_mut();
        if unsafe { rp_new(config, &mut h) } < 0 { Err(last_error()) } else { Ok(Rustpotter { h }) }
    }
    pub fn add_wakeword_from_buffer(&mut self, key: &str, buffer: &[u8]) -> Result<(), String> {
        let k = CString::new(key).map_err(|e| e.to_string())?;
        if unsafe { rp_add_wakeword_from_buffer(self.h, k.as_ptr(), buffer.as_ptr(), buffer.len()) } < 0 { Err(last_error()) } else { Ok(()) }
    }
    pub fn add_wakeword_from_file(&mut self, key: &str, path: &str) -> Result<(), String> {
        let (k, p) = (CString::new(key).map_err(|e| e.to_string())?, CString::new(path).map_err(|e| e.to_string())?);
        if unsafe { rp_add_wakeword_from_file(self.h, k.as_ptr(), p.as_ptr()) } < 0 { Err(last_error()) } else { Ok(()) }
    }
    pub fn remove_wakeword(&mut self, key: &str) -> bool { CString::new(key).map(|k| unsafe { rp_remove_wakeword(self.h, k.as_ptr()) }).unwrap_or(false) }
    pub fn remove_wakewords(&mut self) -> bool { unsafe { rp_remove_wakewords(self.h) } }
    pub fn get_samples_per_frame(&self) -> usize { unsafe { rp_get_samples_per_frame(self.h) } }
    pub fn get_bytes_per_frame(&self) -> usize { unsafe { rp_get_bytes_per_frame(self.h) } }
    pub fn get_rms_level(&self) -> f32 { unsafe { rp_get_rms_level(self.h) } }
    pub fn get_gain(&self) -> f32 { unsafe { rp_get_gain(self.h) } }
    pub fn get_rms_level_ref(&self) -> f32 { unsafe { rp_get_rms_level_ref(self.h) } }
    pub fn get_partial_detection(&self) -> Option<RustpotterDetection> {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        unsafe { if rp_get_partial_detection(self.h, d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } }
    }
    pub fn process_bytes(&mut self, audio_bytes: &[u8]) -> Option<RustpotterDetection> {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        unsafe { if rp_process_bytes(self.h, audio_bytes.as_ptr(), audio_bytes.len(), d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } }
    }
    pub fn process_samples<T: Sample>(&mut self, audio_samples: Vec<T>) -> Option<RustpotterDetection> {
        let mut d = std::mem::MaybeUninit::<rp_detection>::uninit();
        unsafe { if T::process(self.h, &audio_samples, d.as_mut_ptr()) == 1 { Some(owned(&d.assume_init())) } else { None } }
    }
    pub fn update_config(&mut self, c: &rp_config) { unsafe { rp_update_config(self.h, c); } }
    pub fn update_detector_config(&mut self, c: &rp_detector_config) { unsafe { rp_update_detector_config(self.h, c); } }
    pub fn update_filters_config(&mut self, c: &rp_filters_config) { unsafe { rp_update_filters_config(self.h, c); } }
    pub fn reset(&mut self) { unsafe { rp_reset(self.h) } }
}

// ---- offline tooling: WakewordRef::new_from_sample_buffers(..).save_to_buffer() on the device -------------
pub enum rp_ctx {}
extern "C" {
    pub fn rp_ctx_new(device: c_int, flags: c_int, out: *mut *mut rp_ctx) -> c_int;
    pub fn rp_ctx_free(ctx: *mut rp_ctx);
    pub fn rp_wakeword_ref_build(ctx: *mut rp_ctx, name: *const c_char, threshold: *const f32, avg_threshold: *const f32, n: usize,
                                 sample_names: *const *const c_char, wav_buffers: *const *const u8, wav_lens: *const usize,
                                 mfcc_size: u16, rms_from_files: c_int, out_rpw: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn rp_buffer_free(buffer: *mut u8);
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
    unsafe {
        let mut ctx = std::ptr::null_mut();
        if rp_ctx_new(0, 0, &mut ctx) < 0 { return Err(last_error()); }
        let (mut out, mut out_len) = (std::ptr::null_mut(), 0usize);
        let r = rp_wakeword_ref_build(ctx, cname.as_ptr(), threshold.as_ref().map_or(std::ptr::null(), |t| t as *const f32),
                                      avg_threshold.as_ref().map_or(std::ptr::null(), |t| t as *const f32), names.len(),
                                      name_ptrs.as_ptr(), bufs.as_ptr(), lens.as_ptr(), mfcc_size, 0, &mut out, &mut out_len);
        let res = if r < 0 { Err(last_error()) } else { Ok(std::slice::from_raw_parts(out, out_len).to_vec()) };
        if !out.is_null() { rp_buffer_free(out); }
        rp_ctx_free(ctx);
        res
    }
}

// ---- offline tooling: WakewordModel::train_from_buffers(..).save_to_buffer() on the device -----------------
#[repr(C)] #[derive(Clone, Copy)] pub struct rp_train_options { pub m_type: c_int, pub learning_rate: f32, pub epochs: usize, pub test_epochs: usize, pub mfcc_size: u16, pub seed: u64 }
extern "C" {
    pub fn rp_wakeword_model_train(ctx: *mut rp_ctx, options: *const rp_train_options, n_train: usize, train_names: *const *const c_char,
                                   train_wavs: *const *const u8, train_lens: *const usize, n_test: usize, test_names: *const *const c_char,
                                   test_wavs: *const *const u8, test_lens: *const usize, prev_model: *const u8, prev_model_len: usize,
                                   out_rpw: *mut *mut u8, out_len: *mut usize, final_loss: *mut f32, test_accuracy: *mut f32) -> c_int;
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
    unsafe {
        let mut ctx = std::ptr::null_mut();
        if rp_ctx_new(0, 0, &mut ctx) < 0 { return Err(last_error()); }
        let (mut out, mut out_len, mut loss, mut acc) = (std::ptr::null_mut(), 0usize, 0f32, 0f32);
        let (pm, pl) = wakeword_model.map_or((std::ptr::null(), 0), |m| (m.as_ptr(), m.len()));
        let r = rp_wakeword_model_train(ctx, &opt, trp.len(), trp.as_ptr(), trb.as_ptr(), trl.as_ptr(), tep.len(), tep.as_ptr(), teb.as_ptr(),
                                        tel.as_ptr(), pm, pl, &mut out, &mut out_len, &mut loss, &mut acc);
        let res = if r < 0 { Err(last_error()) } else { Ok((std::slice::from_raw_parts(out, out_len).to_vec(), loss, acc)) };
        if !out.is_null() { rp_buffer_free(out); }
        rp_ctx_free(ctx);
        res
    }
}

// ---- server side: S live streams per call instead of one `Rustpotter` per stream ----
pub enum rp_templates {}
pub enum rp_stream_batch {}
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct rp_batch_detection { pub stream: i32, pub frame: i32, pub window: i32, pub counter: i32, pub avg_score: f32, pub score: f32 }
extern "C" {
    pub fn rp_templates_new(ctx: *mut rp_ctx, t: c_int, k: c_int, lens: *const c_int, feats: *const f32, avg_len: c_int, avg: *const f32,
                            out: *mut *mut rp_templates) -> c_int;
    pub fn rp_templates_free(t: *mut rp_templates);
    pub fn rp_stream_batch_new(ctx: *mut rp_ctx, t: *const rp_templates, config: *const rp_detector_config, s: usize,
                               max_chunks_per_call: usize, out: *mut *mut rp_stream_batch) -> c_int;
    pub fn rp_stream_batch_free(b: *mut rp_stream_batch);
    pub fn rp_stream_batch_set_input(b: *mut rp_stream_batch, sample_rate: usize, channels: c_int) -> c_int;
    pub fn rp_stream_batch_samples_per_chunk(b: *const rp_stream_batch) -> usize;
    /// pcm: S rows of n_chunks chunks (host pointers with RP_CTX_HOST_POINTERS, else device pointers); fmt = rp_sample_format
    pub fn rp_stream_batch_process(b: *mut rp_stream_batch, pcm: *const std::ffi::c_void, fmt: c_int, n_chunks: usize, pcm_stride: usize,
                                   det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int, agg: *mut f32) -> c_int;
    pub fn rp_stream_batch_reset(b: *mut rp_stream_batch, stream: i64) -> c_int;
    pub fn rp_stream_batch_chunks_seen(b: *const rp_stream_batch) -> usize;
}

// ---- multi-GPU: independent streams sharded over the GPUs of a node ----
pub const RP_CTX_DEVICE_POINTERS: c_int = 0;
pub const RP_CTX_HOST_POINTERS: c_int = 1;
/// compare every window with every sample template even where the averaged-template gate would skip them
pub const RP_CTX_FULL_SCORES: c_int = 2;
extern "C" {
    pub fn rp_batch_detect_fmt(ctx: *mut rp_ctx, pcm: *const std::ffi::c_void, fmt: c_int, s: usize, n_samples: usize, pcm_stride: usize,
                               t: *const rp_templates, config: *const rp_detector_config, det: *mut rp_batch_detection, n_det: *mut i32,
                               max_det: c_int, scores: *mut f32, agg: *mut f32) -> c_int;
    /// ctxs[g] / t[g]: one context and one replica of the wakeword per device; pcm[g]: the S[g] streams of shard g
    /// ([S[g]][pcm_stride], on ctxs[g]'s device or in host memory per the contexts' flag); det / n_det: ONE gathered
    /// block for all sum(S) streams, `stream` = global id in shard order.  One host thread per shard inside the call.
    pub fn rp_batch_detect_sharded(ctxs: *const *mut rp_ctx, t: *const *const rp_templates, n_shards: c_int,
                                   pcm: *const *const std::ffi::c_void, fmt: c_int, s: *const usize, n_samples: usize, pcm_stride: usize,
                                   config: *const rp_detector_config, det: *mut rp_batch_detection, n_det: *mut i32, max_det: c_int) -> c_int;
}

/// All streams of `shards` (one Vec of equally long f32 streams per GPU, host memory) through the whole path on
/// `devices[g]`; returns the detections of every stream in shard order.  The wakeword is given as its templates
/// ([T] x [len][K], as `WakewordRef::samples_features` holds them) plus the optional averaged template.
pub fn batch_detect_sharded(devices: &[c_int], templates: &[Vec<Vec<f32>>], avg: Option<&Vec<Vec<f32>>>, config: &rp_detector_config,
                            shards: &[Vec<Vec<f32>>], max_det: usize) -> Result<Vec<Vec<rp_batch_detection>>, String> {
    assert_eq!(devices.len(), shards.len());
    let k = templates[0][0].len();
    let lens: Vec<c_int> = templates.iter().map(|t| t.len() as c_int).collect();
    let feats: Vec<f32> = templates.iter().flat_map(|t| t.iter().flatten().copied()).collect();
    let avg_flat: Option<Vec<f32>> = avg.map(|a| a.iter().flatten().copied().collect());
    let n_samples = shards.iter().flat_map(|s| s.iter()).map(|x| x.len()).next().unwrap_or(0);
    unsafe {
        let mut ctxs: Vec<*mut rp_ctx> = Vec::new();
        let mut tms: Vec<*const rp_templates> = Vec::new();
        let free = |ctxs: &Vec<*mut rp_ctx>, tms: &Vec<*const rp_templates>| {
            for t in tms { rp_templates_free(*t as *mut rp_templates); }
            for c in ctxs { rp_ctx_free(*c); }
        };
        for &d in devices {
            let mut c = std::ptr::null_mut();
            if rp_ctx_new(d, RP_CTX_HOST_POINTERS, &mut c) < 0 { let e = last_error(); free(&ctxs, &tms); return Err(e); }
            ctxs.push(c);
            let mut t = std::ptr::null_mut();
            let (ap, al) = avg_flat.as_ref().map_or((std::ptr::null(), 0), |a| (a.as_ptr(), avg.unwrap().len() as c_int));
            if rp_templates_new(c, lens.len() as c_int, k as c_int, lens.as_ptr(), feats.as_ptr(), al, ap, &mut t) < 0 {
                let e = last_error(); free(&ctxs, &tms); return Err(e);
            }
            tms.push(t as *const rp_templates);
        }
        let flat: Vec<Vec<f32>> = shards.iter().map(|s| s.iter().flatten().copied().collect()).collect();
        let ptrs: Vec<*const std::ffi::c_void> = flat.iter().map(|f| f.as_ptr() as *const std::ffi::c_void).collect();
        let counts: Vec<usize> = shards.iter().map(|s| s.len()).collect();
        let total: usize = counts.iter().sum();
        let mut det = vec![rp_batch_detection::default(); total * max_det];
        let mut n_det = vec![0i32; total];
        let r = rp_batch_detect_sharded(ctxs.as_ptr(), tms.as_ptr(), devices.len() as c_int, ptrs.as_ptr(), 3, counts.as_ptr(), n_samples,
                                        n_samples, config, det.as_mut_ptr(), n_det.as_mut_ptr(), max_det as c_int);
        let res = if r < 0 { Err(last_error()) } else {
            Ok((0..total).map(|s| det[s * max_det..s * max_det + (n_det[s] as usize).min(max_det)].to_vec()).collect())
        };
        free(&ctxs, &tms);
        res
    }
}

