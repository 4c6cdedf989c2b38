"""ctypes binding of include/rustpotter_hip.h.

Class and method names follow the reference's public API (src/lib.rs:8-21,
src/detector.rs, src/config.rs) so that the parity tests read like the reference's
own tests (tests/detector.rs).
"""
import ctypes as C
import weakref
import enum
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RustpotterError(RuntimeError):
    """Result::Err(String) of the reference."""


def lib_path():
    # RP_LIB_PATH: load another build of the same library (kernel experiments under tools/scratch); the product is the in-tree one
    return os.environ.get("RP_LIB_PATH") or os.path.join(_HERE, "librustpotter_hip.so")


class SampleFormat(enum.IntEnum):  # src/audio/audio_types.rs:4-9
    I8 = 0
    I16 = 1
    I32 = 2
    F32 = 3


class Endianness(enum.IntEnum):  # src/audio/audio_types.rs:52-56
    Big = 0
    Little = 1
    Native = 2


class ScoreMode(enum.IntEnum):  # src/config.rs:86-96
    Average = 0
    Max = 1
    Median = 2
    P25 = 3
    P50 = 4
    P75 = 5
    P80 = 6
    P90 = 7
    P95 = 8


class VADMode(enum.IntEnum):  # src/config.rs:134-138 (+ None)
    Off = 0
    Easy = 1
    Medium = 2
    Hard = 3


class _AudioFmt(C.Structure):
    _fields_ = [("sample_rate", C.c_size_t), ("sample_format", C.c_int), ("channels", C.c_uint16), ("endianness", C.c_int)]


class _DetectorConfig(C.Structure):
    _fields_ = [("avg_threshold", C.c_float), ("threshold", C.c_float), ("min_scores", C.c_size_t), ("eager", C.c_bool),
                ("score_ref", C.c_float), ("band_size", C.c_uint16), ("score_mode", C.c_int), ("vad_mode", C.c_int)]


class _GainCfg(C.Structure):
    _fields_ = [("enabled", C.c_bool), ("has_gain_ref", C.c_bool), ("gain_ref", C.c_float), ("min_gain", C.c_float),
                ("max_gain", C.c_float)]


class _BandPassCfg(C.Structure):
    _fields_ = [("enabled", C.c_bool), ("low_cutoff", C.c_float), ("high_cutoff", C.c_float)]


class _FiltersCfg(C.Structure):
    _fields_ = [("gain_normalizer", _GainCfg), ("band_pass", _BandPassCfg)]


class _Config(C.Structure):
    _fields_ = [("fmt", _AudioFmt), ("detector", _DetectorConfig), ("filters", _FiltersCfg)]


class _Detection(C.Structure):
    _fields_ = [("name", C.c_char_p), ("avg_score", C.c_float), ("score", C.c_float), ("n_scores", C.c_size_t),
                ("score_names", C.POINTER(C.c_char_p)), ("scores", C.POINTER(C.c_float)), ("counter", C.c_size_t),
                ("gain", C.c_float)]


class _TrainOptions(C.Structure):
    _fields_ = [("m_type", C.c_int), ("learning_rate", C.c_float), ("epochs", C.c_size_t), ("test_epochs", C.c_size_t),
                ("mfcc_size", C.c_uint16), ("seed", C.c_uint64)]


class _BatchDetection(C.Structure):
    _fields_ = [("stream", C.c_int32), ("frame", C.c_int32), ("window", C.c_int32), ("counter", C.c_int32),
                ("avg_score", C.c_float), ("score", C.c_float)]


# every symbol include/rustpotter_hip.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "rp_config_default", "rp_new", "rp_free", "rp_add_wakeword_from_buffer", "rp_add_wakeword_from_file",
    "rp_remove_wakeword", "rp_remove_wakewords", "rp_get_samples_per_frame", "rp_get_bytes_per_frame",
    "rp_get_partial_detection", "rp_get_rms_level", "rp_get_gain", "rp_get_rms_level_ref", "rp_process_bytes",
    "rp_process_samples_i8", "rp_process_samples_i16", "rp_process_samples_i32", "rp_process_samples_f32",
    "rp_update_config", "rp_update_detector_config", "rp_update_filters_config", "rp_reset", "rp_last_error",
    "rp_ctx_new", "rp_ctx_free", "rp_ctx_set_stream", "rp_ctx_synchronize", "rp_ctx_dtw_ref_pairs", "rp_ctx_dtw_kernels", "rp_ctx_set_arithmetic", "rp_ctx_arithmetic", "rp_ctx_last_mlp_kernel", "rp_build_info", "rp_sharded_gather_info", "rp_mfcc_num_frames", "rp_mfcc_batch", "rp_mfcc_batch_fmt", "rp_batch_detect_fmt", "rp_batch_detect_ingest", "rp_frontend_batch", "rp_wakeword_ref_build", "rp_buffer_free",
    "rp_templates_new", "rp_templates_free", "rp_templates_max_len", "rp_dtw_score_batch", "rp_detect_scan", "rp_batch_detect",
    "rp_model_new", "rp_model_free", "rp_mlp_forward_batch", "rp_mlp_forward_windows", "rp_synth_pcm_batch", "rp_ctx_timing_enable", "rp_ctx_timing_read", "rp_ctx_timing_reset",
    "rp_version", "rp_stream_batch_new", "rp_stream_batch_free", "rp_stream_batch_process", "rp_stream_batch_reset",
    "rp_stream_batch_chunks_seen", "rp_resampler_frame_lengths", "rp_resample_batch",
    "rp_wakeword_model_train", "rp_stream_batch_set_input", "rp_stream_batch_samples_per_chunk",
    "rp_batch_detect_multi", "rp_batch_detect_model", "rp_batch_detect_sharded",
    "rp_stream_batch_new_multi", "rp_stream_batch_process_multi",
]


class _WakewordSpec(C.Structure):  # rp_wakeword_spec
    _fields_ = [("templates", C.c_void_p), ("model", C.c_void_p), ("none_index", C.c_int), ("precision", C.c_int),
                ("threshold", C.c_float), ("avg_threshold", C.c_float)]


def load_library():
    """Loads librustpotter_hip.so.  torch (if importable) is imported FIRST so that the
    process uses torch's bundled HIP runtime (same SONAME libamdhip64.so.7) instead of
    loading a second one."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RustpotterError(
            "librustpotter_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rustpotter_amd/csrc`; there is no CPU fallback" % path)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for pure C users
        pass
    L = C.CDLL(path)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.rp_last_error.restype = C.c_char_p
    L.rp_version.restype = C.c_char_p
    L.rp_config_default.argtypes = [C.POINTER(_Config)]
    L.rp_new.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.rp_free.argtypes = [vp]
    L.rp_add_wakeword_from_buffer.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.rp_add_wakeword_from_file.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.rp_remove_wakeword.argtypes = [vp, C.c_char_p]
    L.rp_remove_wakeword.restype = C.c_bool
    L.rp_remove_wakewords.argtypes = [vp]
    L.rp_remove_wakewords.restype = C.c_bool
    L.rp_get_samples_per_frame.argtypes = [vp]
    L.rp_get_samples_per_frame.restype = C.c_size_t
    L.rp_get_bytes_per_frame.argtypes = [vp]
    L.rp_get_bytes_per_frame.restype = C.c_size_t
    L.rp_get_partial_detection.argtypes = [vp, C.POINTER(_Detection)]
    for n in ("rp_get_rms_level", "rp_get_gain", "rp_get_rms_level_ref"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_float
    L.rp_process_bytes.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(_Detection)]
    for n in ("rp_process_samples_i8", "rp_process_samples_i16", "rp_process_samples_i32", "rp_process_samples_f32"):
        getattr(L, n).argtypes = [vp, vp, C.c_size_t, C.POINTER(_Detection)]
    L.rp_update_config.argtypes = [vp, C.POINTER(_Config)]
    L.rp_update_detector_config.argtypes = [vp, C.POINTER(_DetectorConfig)]
    L.rp_update_filters_config.argtypes = [vp, C.POINTER(_FiltersCfg)]
    L.rp_reset.argtypes = [vp]
    L.rp_ctx_new.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.rp_ctx_free.argtypes = [vp]
    L.rp_ctx_set_stream.argtypes = [vp, vp]
    L.rp_ctx_synchronize.argtypes = [vp]
    L.rp_ctx_dtw_ref_pairs.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rp_ctx_dtw_kernels.argtypes = [vp]
    L.rp_ctx_dtw_kernels.restype = C.c_int
    L.rp_ctx_set_arithmetic.argtypes = [vp, C.c_int, C.c_int]
    L.rp_ctx_set_arithmetic.restype = C.c_int
    L.rp_ctx_arithmetic.argtypes = [vp, C.POINTER(C.c_int)]
    L.rp_ctx_arithmetic.restype = C.c_int
    L.rp_sharded_gather_info.argtypes = []
    L.rp_sharded_gather_info.restype = C.c_char_p
    L.rp_build_info.argtypes = []
    L.rp_build_info.restype = C.c_char_p
    L.rp_ctx_last_mlp_kernel.argtypes = [vp]
    L.rp_ctx_last_mlp_kernel.restype = C.c_char_p
    L.rp_mfcc_num_frames.argtypes = [C.c_size_t]
    L.rp_mfcc_num_frames.restype = C.c_size_t
    L.rp_mfcc_batch.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.rp_mfcc_batch_fmt.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.rp_batch_detect_fmt.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.POINTER(_DetectorConfig), vp, vp, C.c_int, vp, vp]
    L.rp_batch_detect_ingest.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.POINTER(_DetectorConfig), vp, vp, C.c_int, C.c_size_t,
                                         C.POINTER(C.c_double)]
    L.rp_frontend_batch.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_FiltersCfg), C.c_float, C.c_size_t, vp, C.c_size_t, vp, vp]
    L.rp_wakeword_ref_build.argtypes = [vp, C.c_char_p, fp, fp, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_size_t), C.c_uint16, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.rp_buffer_free.argtypes = [vp]
    L.rp_templates_new.argtypes = [vp, C.c_int, C.c_int, ip, fp, C.c_int, fp, C.POINTER(vp)]
    L.rp_templates_free.argtypes = [vp]
    L.rp_templates_max_len.argtypes = [vp]
    L.rp_dtw_score_batch.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.rp_detect_scan.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(_DetectorConfig), C.c_int, vp, C.c_int, vp, vp, C.c_int]
    L.rp_batch_detect.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.POINTER(_DetectorConfig), vp, vp, C.c_int, vp, vp]
    L.rp_wakeword_model_train.argtypes = [vp, C.POINTER(_TrainOptions), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), fp, fp]
    L.rp_batch_detect_multi.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp),
                                        C.POINTER(_DetectorConfig), fp, fp, vp, vp, vp, C.c_int]
    L.rp_batch_detect_sharded.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_int, C.POINTER(C.c_size_t), C.c_size_t,
                                          C.c_size_t, C.POINTER(_DetectorConfig), vp, vp, C.c_int]
    L.rp_batch_detect_model.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_int, C.c_int,
                                        C.POINTER(_DetectorConfig), C.c_int, vp, vp, vp, C.c_int]
    L.rp_resampler_frame_lengths.argtypes = [C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.rp_resample_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    L.rp_stream_batch_new.argtypes = [vp, vp, C.POINTER(_DetectorConfig), C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.rp_stream_batch_new_multi.argtypes = [vp, C.c_size_t, C.POINTER(_WakewordSpec), C.c_int, C.POINTER(_DetectorConfig), C.c_size_t,
                                            C.c_size_t, C.POINTER(vp)]
    L.rp_stream_batch_process_multi.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, vp, vp, vp, vp, C.c_int]
    L.rp_stream_batch_free.argtypes = [vp]
    L.rp_stream_batch_free.restype = None
    L.rp_stream_batch_process.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t, vp, vp, C.c_int, vp]
    L.rp_stream_batch_reset.argtypes = [vp, C.c_longlong]
    L.rp_stream_batch_set_input.argtypes = [vp, C.c_size_t, C.c_int]
    L.rp_stream_batch_samples_per_chunk.argtypes = [vp]
    L.rp_stream_batch_samples_per_chunk.restype = C.c_size_t
    L.rp_stream_batch_chunks_seen.argtypes = [vp]
    L.rp_stream_batch_chunks_seen.restype = C.c_size_t
    L.rp_model_new.argtypes = [vp, C.c_int, ip, C.POINTER(fp), C.POINTER(fp), C.POINTER(vp)]
    L.rp_model_free.argtypes = [vp]
    L.rp_mlp_forward_batch.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.rp_mlp_forward_windows.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, vp]
    L.rp_synth_pcm_batch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    L.rp_ctx_timing_enable.argtypes = [vp, C.c_int]
    L.rp_ctx_timing_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), ip]
    L.rp_ctx_timing_reset.argtypes = [vp]
    _LIB = L
    return L


def _err():
    return RustpotterError(load_library().rp_last_error().decode("utf-8", "replace"))


def sharded_gather_info():
    """How this thread's last batch_detect_sharded[_dev] gathered its results (rp_sharded_gather_info)."""
    return load_library().rp_sharded_gather_info().decode()


def build_info():
    """Architecture and non-default compiler flags of the loaded library (rp_build_info)."""
    return load_library().rp_build_info().decode()


def resampler_frame_lengths(sample_rate):
    """(input frame length per channel, 16 kHz samples it yields) -- AudioEncoder::new, src/audio/encoder.rs:63-83."""
    a, b = C.c_size_t(), C.c_size_t()
    if load_library().rp_resampler_frame_lengths(sample_rate, C.byref(a), C.byref(b)) < 0:
        raise _err()
    return a.value, b.value


def mfcc_num_frames(n_samples):
    return int(load_library().rp_mfcc_num_frames(n_samples))


# ------------------------------------------------------------------ config mirror
class AudioFmt:  # src/config.rs:10-29
    def __init__(self):
        self.sample_rate = 16000
        self.sample_format = SampleFormat.F32
        self.channels = 1
        self.endianness = Endianness.Little


class GainNormalizationConfig:  # src/config.rs:32-52
    def __init__(self):
        self.enabled = False
        self.gain_ref = None
        self.min_gain = 0.1
        self.max_gain = 1.0


class BandPassConfig:  # src/config.rs:55-71
    def __init__(self):
        self.enabled = False
        self.low_cutoff = 80.0
        self.high_cutoff = 400.0


class FiltersConfig:  # src/config.rs:75-82
    def __init__(self):
        self.gain_normalizer = GainNormalizationConfig()
        self.band_pass = BandPassConfig()


class DetectorConfig:  # src/config.rs:172-207
    def __init__(self):
        self.avg_threshold = 0.2
        self.threshold = 0.5
        self.min_scores = 5
        self.eager = False
        self.score_ref = 0.22
        self.band_size = 5
        self.score_mode = ScoreMode.Max
        self.vad_mode = None

    def _c(self):
        c = _DetectorConfig()
        c.avg_threshold, c.threshold, c.min_scores, c.eager = self.avg_threshold, self.threshold, self.min_scores, self.eager
        c.score_ref, c.band_size, c.score_mode = self.score_ref, self.band_size, int(self.score_mode)
        c.vad_mode = int(self.vad_mode) if self.vad_mode is not None else 0
        return c


class RustpotterConfig:  # src/config.rs:212-219
    def __init__(self):
        self.fmt = AudioFmt()
        self.detector = DetectorConfig()
        self.filters = FiltersConfig()

    @staticmethod
    def default():
        return RustpotterConfig()

    def _filters_c(self):
        f = _FiltersCfg()
        g, b = self.filters.gain_normalizer, self.filters.band_pass
        f.gain_normalizer.enabled = g.enabled
        f.gain_normalizer.has_gain_ref = g.gain_ref is not None
        f.gain_normalizer.gain_ref = g.gain_ref if g.gain_ref is not None else 0.0
        f.gain_normalizer.min_gain, f.gain_normalizer.max_gain = g.min_gain, g.max_gain
        f.band_pass.enabled, f.band_pass.low_cutoff, f.band_pass.high_cutoff = b.enabled, b.low_cutoff, b.high_cutoff
        return f

    def _c(self):
        c = _Config()
        c.fmt.sample_rate, c.fmt.sample_format = self.fmt.sample_rate, int(self.fmt.sample_format)
        c.fmt.channels, c.fmt.endianness = self.fmt.channels, int(self.fmt.endianness)
        c.detector = self.detector._c()
        c.filters = self._filters_c()
        return c


class RustpotterDetection:  # src/detector.rs:488-501
    def __init__(self, d):
        import numpy as np
        self.name = d.name.decode()
        self.avg_score = np.float32(d.avg_score)
        self.score = np.float32(d.score)
        self.scores = {d.score_names[i].decode(): np.float32(d.scores[i]) for i in range(d.n_scores)}
        self.counter = int(d.counter)
        self.gain = np.float32(d.gain)

    def __repr__(self):
        return "RustpotterDetection(name=%r, avg_score=%r, score=%r, counter=%d)" % (self.name, float(self.avg_score),
                                                                                      float(self.score), self.counter)


class Rustpotter:
    """src/detector.rs:34-501.  Raises RustpotterError where the reference returns Err(String)."""

    def __init__(self, config):
        self._L = load_library()
        self._h = C.c_void_p()
        c = config._c()
        if self._L.rp_new(C.byref(c), C.byref(self._h)) < 0:
            self._h = None
            raise _err()

    @staticmethod
    def new(config):
        return Rustpotter(config)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rp_free(self._h)
            self._h = None

    def add_wakeword_from_file(self, key, path):
        if self._L.rp_add_wakeword_from_file(self._h, key.encode(), path.encode()) < 0:
            raise _err()

    def add_wakeword_from_buffer(self, key, buffer):
        if self._L.rp_add_wakeword_from_buffer(self._h, key.encode(), bytes(buffer), len(buffer)) < 0:
            raise _err()

    def remove_wakeword(self, key):
        return bool(self._L.rp_remove_wakeword(self._h, key.encode()))

    def remove_wakewords(self):
        return bool(self._L.rp_remove_wakewords(self._h))

    def get_samples_per_frame(self):
        return int(self._L.rp_get_samples_per_frame(self._h))

    def get_bytes_per_frame(self):
        return int(self._L.rp_get_bytes_per_frame(self._h))

    def get_partial_detection(self):
        d = _Detection()
        return RustpotterDetection(d) if self._L.rp_get_partial_detection(self._h, C.byref(d)) == 1 else None

    def get_rms_level(self):
        return float(self._L.rp_get_rms_level(self._h))

    def get_gain(self):
        return float(self._L.rp_get_gain(self._h))

    def get_rms_level_ref(self):
        return float(self._L.rp_get_rms_level_ref(self._h))

    def _ret(self, r, d):
        if r < 0:
            raise _err()
        return RustpotterDetection(d) if r == 1 else None

    def process_bytes(self, audio_bytes):
        d = _Detection()
        b = bytes(audio_bytes)
        return self._ret(self._L.rp_process_bytes(self._h, b, len(b), C.byref(d)), d)

    def process_samples(self, samples):
        """samples: numpy array of int8/int16/int32/float32 (the reference's `Sample` types)."""
        import numpy as np
        a = np.ascontiguousarray(samples)
        fn = {np.dtype(np.int8): self._L.rp_process_samples_i8, np.dtype(np.int16): self._L.rp_process_samples_i16,
              np.dtype(np.int32): self._L.rp_process_samples_i32, np.dtype(np.float32): self._L.rp_process_samples_f32}[a.dtype]
        d = _Detection()
        return self._ret(fn(self._h, a.ctypes.data, a.size, C.byref(d)), d)

    def update_config(self, config):
        c = config._c()
        self._L.rp_update_config(self._h, C.byref(c))

    def update_detector_config(self, config):
        c = config._c()
        self._L.rp_update_detector_config(self._h, C.byref(c))

    def update_filters_config(self, config):
        rc = RustpotterConfig()
        rc.filters = config
        f = rc._filters_c()
        self._L.rp_update_filters_config(self._h, C.byref(f))

    def reset(self):
        self._L.rp_reset(self._h)


# ------------------------------------------------------------- batched operators
class Templates:
    """rp_templates: one wakeword reference resident on the device."""

    def __init__(self, ctx, templates, avg=None):
        import numpy as np
        self._L = load_library()
        self.ctx = ctx
        self.T = len(templates)
        self.K = int(np.asarray(templates[0]).shape[1])
        lens = np.array([len(t) for t in templates], np.int32)
        feats = np.concatenate([np.ascontiguousarray(t, np.float32).reshape(-1) for t in templates]).astype(np.float32)
        avg_a = None if avg is None else np.ascontiguousarray(avg, np.float32)
        self._h = C.c_void_p()
        r = self._L.rp_templates_new(ctx._h, self.T, self.K, lens.ctypes.data_as(C.POINTER(C.c_int)),
                                     feats.ctypes.data_as(C.POINTER(C.c_float)), 0 if avg_a is None else avg_a.shape[0],
                                     None if avg_a is None else avg_a.ctypes.data_as(C.POINTER(C.c_float)), C.byref(self._h))
        if r < 0:
            self._h = None
            raise _err()
        self.max_len = int(self._L.rp_templates_max_len(self._h))
        self.has_avg = avg_a is not None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rp_templates_free(self._h)
            self._h = None


class Model:
    """rp_model: a wakeword model (Linear/ReLU stack) resident on the device."""

    def __init__(self, ctx, weights, biases):
        import numpy as np
        self._L = load_library()
        self.ctx = ctx
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in biases]
        dims = np.array([ws[0].shape[1]] + [w.shape[0] for w in ws], np.int32)
        fp = C.POINTER(C.c_float)
        wp = (fp * len(ws))(*[w.ctypes.data_as(fp) for w in ws])
        bp = (fp * len(bs))(*[b.ctypes.data_as(fp) for b in bs])
        self._h = C.c_void_p()
        if self._L.rp_model_new(ctx._h, len(ws), dims.ctypes.data_as(C.POINTER(C.c_int)), wp, bp, C.byref(self._h)) < 0:
            self._h = None
            raise _err()
        self.n_in, self.n_out = int(dims[0]), int(dims[-1])

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rp_model_free(self._h)
            self._h = None


DET_DTYPE = [("stream", "<i4"), ("frame", "<i4"), ("window", "<i4"), ("counter", "<i4"), ("avg_score", "<f4"), ("score", "<f4")]


class StreamBatch:
    """S live streams fed chunk by chunk (rp_stream_batch_*): the batched form of calling
    Rustpotter::process_samples on S instances sharing one wakeword and config."""

    def __init__(self, ctx, templates, detector_config, S, max_chunks_per_call=1, sample_rate=16000, channels=1, wakewords=None,
                 mfcc_size=None):
        """templates: one wakeword reference (rp_stream_batch_new).  wakewords (rp_stream_batch_new_multi): a list of
        dicts, each {"templates": Templates} or {"model": Model, "none_index": int, "precision": "f32" | "bf16"}, optionally
        with "threshold" / "avg_threshold" (the wakeword's own overrides); mfcc_size is then required."""
        self._L = load_library()
        self.ctx, self.templates, self.S, self.max_chunks = ctx, templates, S, max_chunks_per_call
        h = C.c_void_p()
        c = detector_config._c()
        if wakewords is None:
            if self._L.rp_stream_batch_new(ctx._h, templates._h, C.byref(c), S, max_chunks_per_call, C.byref(h)) < 0:
                raise _err()
        else:
            specs = (_WakewordSpec * len(wakewords))()
            self._keep = list(wakewords)   # the batch borrows the handles
            for sp, w in zip(specs, wakewords):
                sp.templates = w["templates"]._h if w.get("templates") is not None else None
                sp.model = w["model"]._h if w.get("model") is not None else None
                sp.none_index = w.get("none_index", -1)
                sp.precision = {"f32": 0, "bf16": 1}[w.get("precision", "f32")]
                sp.threshold = float("nan") if w.get("threshold") is None else w["threshold"]
                sp.avg_threshold = float("nan") if w.get("avg_threshold") is None else w["avg_threshold"]
            if self._L.rp_stream_batch_new_multi(ctx._h, len(wakewords), specs, mfcc_size, C.byref(c), S, max_chunks_per_call, C.byref(h)) < 0:
                raise _err()
        self._h = h
        if (sample_rate, channels) != (16000, 1) and self._L.rp_stream_batch_set_input(h, sample_rate, channels) < 0:
            raise _err()
        self.samples_per_chunk = self._L.rp_stream_batch_samples_per_chunk(h)
        # MFCC frames a stream gains per input frame: 3 (30 ms frames) or 4 (the 40 ms frames of 11.025 / 22.05 kHz input)
        self.frames_per_chunk = resampler_frame_lengths(sample_rate)[1] // 160

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rp_stream_batch_free(self._h)
            self._h = None

    @property
    def chunks_seen(self):
        return self._L.rp_stream_batch_chunks_seen(self._h)

    def process(self, pcm, max_det=4, want_agg=False, n_chunks=None):
        """pcm [S][n_chunks*480] numpy (i8 / i16 / i32 / f32) -> (det, n_det[, agg]) for these chunks.  With n_chunks
        given, rows may be longer than the chunks they carry (row pitch = pcm.shape[1])."""
        import numpy as np
        assert self.ctx.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        S, N = pcm.shape
        spc = self.samples_per_chunk
        if S != self.S or (n_chunks is None and N % spc) or (n_chunks is not None and n_chunks * spc > N):
            raise ValueError("pcm must be [S][n_chunks*samples_per_chunk]")
        nc = N // spc if n_chunks is None else n_chunks
        det = np.zeros((S, max_det), dtype=DET_DTYPE)
        n_det = np.zeros(S, np.int32)
        agg = np.empty((S, self.frames_per_chunk * nc), np.float32) if want_agg else None
        if self._L.rp_stream_batch_process(self._h, pcm.ctypes.data, fmt, nc, N, det.ctypes.data, n_det.ctypes.data, max_det,
                                           None if agg is None else agg.ctypes.data) < 0:
            raise _err()
        return (det, n_det, agg) if want_agg else (det, n_det)

    def process_multi(self, pcm, max_det=4, n_chunks=None):
        """rp_stream_batch_process_multi -> (det, det_wakeword, det_label, n_det)"""
        import numpy as np
        assert self.ctx.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        S, N = pcm.shape
        spc = self.samples_per_chunk
        if S != self.S or (n_chunks is None and N % spc) or (n_chunks is not None and n_chunks * spc > N):
            raise ValueError("pcm must be [S][n_chunks*samples_per_chunk]")
        nc = N // spc if n_chunks is None else n_chunks
        det = np.zeros((S, max_det), dtype=DET_DTYPE)
        dww = np.zeros((S, max_det), np.int32)
        dlab = np.zeros((S, max_det), np.int32)
        n_det = np.zeros(S, np.int32)
        if self._L.rp_stream_batch_process_multi(self._h, pcm.ctypes.data, fmt, nc, N, det.ctypes.data, dww.ctypes.data, dlab.ctypes.data,
                                                 n_det.ctypes.data, max_det) < 0:
            raise _err()
        return det, dww, dlab, n_det

    def process_dev(self, pcm_ptr, fmt, n_chunks, stride, det_ptr, n_det_ptr, max_det, agg_ptr=None):
        if self._L.rp_stream_batch_process(self._h, pcm_ptr, fmt, n_chunks, stride, det_ptr, n_det_ptr, max_det, agg_ptr) < 0:
            raise _err()

    def reset(self, stream=-1):
        if self._L.rp_stream_batch_reset(self._h, stream) < 0:
            raise _err()


def batch_detect_sharded(ctxs, templates, pcms, detector_config, max_det=8):
    """rp_batch_detect_sharded with host-pointer contexts: pcms = one [S_g][N] numpy array per shard (same N and dtype)
    -> (det [sum S][max_det], n_det [sum S]) gathered, global stream ids."""
    import numpy as np
    L = load_library()
    n = len(ctxs)
    assert n == len(templates) == len(pcms) and all(c.host for c in ctxs)
    arrs = [np.ascontiguousarray(p) for p in pcms]
    # the C call reads every shard as S_g * N samples of ONE format: refuse shards of another length or sample type
    if any(a.ndim != 2 for a in arrs):
        raise ValueError("batch_detect_sharded: every shard must be a 2-D array [streams][samples]")
    N = arrs[0].shape[1]
    if any(a.shape[1] != N or a.dtype != arrs[0].dtype for a in arrs):
        raise ValueError("batch_detect_sharded: all shards must hold streams of the same length and sample type")
    fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(arrs[0].dtype)
    if fmt is None:
        arrs, fmt = [np.ascontiguousarray(a, np.float32) for a in arrs], 3
    S = (C.c_size_t * n)(*[a.shape[0] for a in arrs])
    total = sum(a.shape[0] for a in arrs)
    det = np.zeros((total, max_det), dtype=DET_DTYPE)
    n_det = np.zeros(total, np.int32)
    c = detector_config._c()
    hc = (C.c_void_p * n)(*[x._h for x in ctxs])
    ht = (C.c_void_p * n)(*[x._h for x in templates])
    hp = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    if L.rp_batch_detect_sharded(hc, ht, n, hp, fmt, S, N, N, C.byref(c), det.ctypes.data, n_det.ctypes.data, max_det) < 0:
        raise _err()
    return det, n_det


def batch_detect_sharded_dev(ctxs, templates, pcm_ptrs, S_list, N, stride, detector_config, det_ptr, n_det_ptr, max_det):
    """Device-pointer form: pcm_ptrs[g] on ctxs[g]'s device, det / n_det gathered on ctxs[0]'s device."""
    L = load_library()
    n = len(ctxs)
    if not (n == len(templates) == len(pcm_ptrs) == len(S_list)):
        raise ValueError("batch_detect_sharded_dev: one context, template set, PCM pointer and stream count per shard")
    if any(x.host for x in ctxs):
        raise ValueError("batch_detect_sharded_dev: the contexts must take device pointers (host_pointers=False)")
    c = detector_config._c()
    hc = (C.c_void_p * n)(*[x._h for x in ctxs])
    ht = (C.c_void_p * n)(*[x._h for x in templates])
    hp = (C.c_void_p * n)(*pcm_ptrs)
    S = (C.c_size_t * n)(*S_list)
    if L.rp_batch_detect_sharded(hc, ht, n, hp, 3, S, N, stride, C.byref(c), det_ptr, n_det_ptr, max_det) < 0:
        raise _err()


MLP_PRECISION = {"f32": 0, "bf16": 1, "f32_strict": 2, "f32_fast": 3}   # RP_MLP_F32 / RP_MLP_BF16 / RP_MLP_F32_STRICT / RP_MLP_F32_FAST


_LIVE_CONTEXTS = weakref.WeakSet()   # every BatchContext alive (arithmetic_all)


class arithmetic_all:
    """Context manager for tests: every live BatchContext runs the calls inside with this arithmetic (rp_ctx_set_arithmetic on each),
    contexts made inside start with it too, and every setting comes back afterwards.  What the RP_DTW_MFMA / RP_DTW_RAGGED environment
    switches of rounds 3-5 did, through the ABI."""

    def __init__(self, arithmetic, ragged_matrix=False):
        self.arithmetic, self.ragged = arithmetic, ragged_matrix

    def __enter__(self):
        self.saved = [(c, c.get_arithmetic()) for c in list(_LIVE_CONTEXTS) if getattr(c, "_h", None)]
        for c, _ in self.saved:
            c.set_arithmetic(self.arithmetic, self.ragged)
        return self

    def __exit__(self, *a):
        for c, old in self.saved:
            if getattr(c, "_h", None):
                c.set_arithmetic(*old)


class BatchContext:
    """rp_ctx.  host_pointers=True: numpy in / numpy out (tests); False: raw device
    pointers (bench.py passes torch tensors' data_ptr())."""

    # RP_ARITH_* (include/rustpotter_hip.h): the arithmetic of the DTW cost's cosine products
    ARITH = {"f32_matrix": 0, "strict_f32": 1, "fast_split": 2}

    def __init__(self, device=0, host_pointers=True, full_scores=False, arithmetic="f32_matrix", ragged_matrix=False):
        """full_scores (RP_CTX_FULL_SCORES): batch_detect compares every window with every sample template even where the
        averaged-template gate would skip them.  arithmetic / ragged_matrix: RP_CTX_ARITH_* / RP_CTX_RAGGED_MATRIX."""
        self._L = load_library()
        self._h = C.c_void_p()
        self.host = host_pointers
        flags = (1 if host_pointers else 0) | (2 if full_scores else 0) | {0: 0, 1: 4, 2: 8}[self.ARITH[arithmetic]] | (16 if ragged_matrix else 0)
        if self._L.rp_ctx_new(device, flags, C.byref(self._h)) < 0:
            self._h = None
            raise _err()
        _LIVE_CONTEXTS.add(self)

    def set_arithmetic(self, arithmetic, ragged_matrix=False):
        """rp_ctx_set_arithmetic: "f32_matrix" (default: three bf16 parts, f32-grade), "strict_f32" (vector FMAs only), "fast_split" (two f16 parts)."""
        if self._L.rp_ctx_set_arithmetic(self._h, self.ARITH[arithmetic], 1 if ragged_matrix else 0) < 0:
            raise _err()

    def get_arithmetic(self):
        """(name, ragged_matrix) of rp_ctx_arithmetic."""
        r = C.c_int(0)
        m = int(self._L.rp_ctx_arithmetic(self._h, C.byref(r)))
        return {v: k for k, v in self.ARITH.items()}[m], bool(r.value)

    def arithmetic(self, arithmetic, ragged_matrix=False):
        """Context manager: the calls inside run with this arithmetic, the previous setting comes back afterwards."""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.old = ctx.get_arithmetic()
                ctx.set_arithmetic(arithmetic, ragged_matrix)
                return ctx

            def __exit__(self, *a):
                ctx.set_arithmetic(*self.old)
        return _Scope()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rp_ctx_free(self._h)
            self._h = None

    def set_stream(self, hip_stream):
        self._L.rp_ctx_set_stream(self._h, C.c_void_p(hip_stream))

    def synchronize(self):
        if self._L.rp_ctx_synchronize(self._h) < 0:
            raise _err()

    def last_mlp_kernel(self):
        """Which kernel(s) the last dense-row wakeword-model forward of this context ran (rp_ctx_last_mlp_kernel)."""
        self._L.rp_ctx_last_mlp_kernel.restype = C.c_char_p
        v = self._L.rp_ctx_last_mlp_kernel(self._h)
        return v.decode() if v else ""

    def dtw_ref_pairs(self):
        """(window, templates) pairs this context's DTW calls rescored with the reference-shaped cosine so far (rp_ctx_dtw_ref_pairs)."""
        v = C.c_uint64(0)
        if self._L.rp_ctx_dtw_ref_pairs(self._h, C.byref(v)) < 0:
            raise _err()
        return int(v.value)

    DTW_KERNELS = {1: "dtw_mfma_kernel", 2: "dtw_mfma_wide_kernel", 4: "dtw_ragged_kernel", 8: "register kernels", 16: "dtw_generic_kernel",
                   32: "dtw_single_kernel", 64: "dtw_ref_kernel (every window)", 128: "dtw_mfma_group_kernel"}
    DTW_PRODUCTS = {256: "bf16x3", 512: "f16x2"}

    def dtw_kernels(self):
        """Names of the DTW kernel families this context launched since the last call of this method (rp_ctx_dtw_kernels)."""
        m = int(self._L.rp_ctx_dtw_kernels(self._h))
        self.last_dtw_products = [n for b, n in self.DTW_PRODUCTS.items() if m & b]   # the matrix-core launches' product arithmetic
        return [n for b, n in self.DTW_KERNELS.items() if m & b]

    # --- numpy convenience (host_pointers=True)
    def mfcc(self, pcm, K):
        """pcm: float32, or int8/int16/int32 samples decoded on the device like the reference's `Sample` types."""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        nf = mfcc_num_frames(N)
        out = np.empty((S, nf, K), np.float32)
        if self._L.rp_mfcc_batch_fmt(self._h, pcm.ctypes.data, fmt, S, N, N, K, out.ctypes.data) < 0:
            raise _err()
        return out

    def build_wakeword_ref(self, name, samples, mfcc_size, threshold=None, avg_threshold=None, from_files=True):
        """WakewordRef::new_from_sample_files / _buffers + save_to_buffer: samples = ordered {name: wav bytes};
        returns the .rpw bytes."""
        names = [k.encode() for k in samples]
        bufs = [bytes(v) for v in samples.values()]
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_bufs = (C.c_char_p * n)(*bufs)
        c_lens = (C.c_size_t * n)(*[len(b) for b in bufs])
        thr = None if threshold is None else C.byref(C.c_float(threshold))
        athr = None if avg_threshold is None else C.byref(C.c_float(avg_threshold))
        out, out_len = C.c_void_p(), C.c_size_t()
        if self._L.rp_wakeword_ref_build(self._h, name.encode(), C.cast(thr, C.POINTER(C.c_float)) if thr else None,
                                         C.cast(athr, C.POINTER(C.c_float)) if athr else None, n, c_names, c_bufs, c_lens,
                                         mfcc_size, 1 if from_files else 0, C.byref(out), C.byref(out_len)) < 0:
            raise _err()
        data = C.string_at(out, out_len.value)
        self._L.rp_buffer_free(out)
        return data

    def train_wakeword_model(self, train, test, m_type="medium", learning_rate=0.027, epochs=10, test_epochs=10, mfcc_size=16,
                             seed=1, prev_model=None):
        """WakewordModel::train_from_buffers + save_to_buffer: train / test = ordered {file name: wav bytes}
        (label in [brackets]); returns (.rpw bytes, last loss, test accuracy)."""
        def pack(d):
            names = [k.encode() for k in d]
            bufs = [bytes(v) for v in d.values()]
            n = len(names)
            return n, (C.c_char_p * n)(*names), (C.c_char_p * n)(*bufs), (C.c_size_t * n)(*[len(b) for b in bufs])
        ntr, trn, trb, trl = pack(train)
        nte, ten, teb, tel = pack(test)
        opt = _TrainOptions({"tiny": 0, "small": 1, "medium": 2, "large": 3}[m_type], learning_rate, epochs, test_epochs, mfcc_size, seed)
        out, out_len, loss, acc = C.c_void_p(), C.c_size_t(), C.c_float(), C.c_float()
        if self._L.rp_wakeword_model_train(self._h, C.byref(opt), ntr, trn, trb, trl, nte, ten, teb, tel, prev_model,
                                           0 if prev_model is None else len(prev_model), C.byref(out), C.byref(out_len),
                                           C.byref(loss), C.byref(acc)) < 0:
            raise _err()
        data = C.string_at(out, out_len.value)
        self._L.rp_buffer_free(out)
        return data, loss.value, acc.value

    def frontend(self, pcm, filters_config, rms_level_ref, window_size):
        """Decode + gain normaliser + band-pass over whole streams -> (pcm f32, rms, gains)."""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        rc = RustpotterConfig()
        rc.filters = filters_config
        f = rc._filters_c()
        out = np.empty((S, N), np.float32)
        rms = np.empty((S, N // 480), np.float32)
        gains = np.empty((S, N // 480), np.float32)
        if self._L.rp_frontend_batch(self._h, pcm.ctypes.data, fmt, S, N, N, C.byref(f), rms_level_ref, window_size,
                                     out.ctypes.data, N, rms.ctypes.data, gains.ctypes.data) < 0:
            raise _err()
        return out, rms, gains

    def dtw_scores(self, mfcc, templates, score_ref=0.22, band_size=5, score_mode=ScoreMode.Max, with_avg=False):
        import numpy as np
        assert self.host
        mfcc = np.ascontiguousarray(mfcc, np.float32)
        if mfcc.ndim == 2:
            mfcc = mfcc[None]
        S, nf, K = mfcc.shape
        assert K == templates.K
        n_win = max(0, nf - templates.max_len + 1)
        scores = np.empty((S, n_win, templates.T), np.float32)
        agg = np.empty((S, n_win), np.float32)
        avg = np.empty((S, n_win), np.float32) if (with_avg and templates.has_avg) else None
        r = self._L.rp_dtw_score_batch(self._h, mfcc.ctypes.data, S, nf, templates._h, score_ref, band_size, int(score_mode),
                                       1 if avg is not None else 0, scores.ctypes.data,
                                       None if avg is None else avg.ctypes.data, agg.ctypes.data)
        if r < 0:
            raise _err()
        return scores, avg, agg

    def detect_scan(self, agg, avg, n_frames, max_len, detector_config, max_det=8, mfcc=None):
        import numpy as np
        assert self.host
        agg = np.ascontiguousarray(agg, np.float32)
        S = agg.shape[0]
        det = np.zeros((S, max_det), dtype=[("stream", "<i4"), ("frame", "<i4"), ("window", "<i4"), ("counter", "<i4"),
                                             ("avg_score", "<f4"), ("score", "<f4")])
        n_det = np.zeros(S, np.int32)
        c = detector_config._c()
        avg_a = None if avg is None else np.ascontiguousarray(avg, np.float32)
        mf = None if mfcc is None else np.ascontiguousarray(mfcc, np.float32)
        r = self._L.rp_detect_scan(self._h, agg.ctypes.data, None if avg_a is None else avg_a.ctypes.data, S, n_frames,
                                   max_len, C.byref(c), 1 if avg_a is not None else 0,
                                   None if mf is None else mf.ctypes.data, 0 if mf is None else mf.shape[-1],
                                   det.ctypes.data, n_det.ctypes.data, max_det)
        if r < 0:
            raise _err()
        return det, n_det

    def batch_detect(self, pcm, templates, detector_config, max_det=8, want_scores=False):
        """Whole path for S streams in one call (numpy in / out)."""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        nf = mfcc_num_frames(N)
        n_win = max(0, nf - templates.max_len + 1)
        det = np.zeros((S, max_det), dtype=[("stream", "<i4"), ("frame", "<i4"), ("window", "<i4"), ("counter", "<i4"),
                                             ("avg_score", "<f4"), ("score", "<f4")])
        n_det = np.zeros(S, np.int32)
        scores = np.empty((S, n_win, templates.T), np.float32) if want_scores else None
        agg = np.empty((S, n_win), np.float32) if want_scores else None
        c = detector_config._c()
        r = self._L.rp_batch_detect_fmt(self._h, pcm.ctypes.data, fmt, S, N, N, templates._h, C.byref(c), det.ctypes.data,
                                        n_det.ctypes.data, max_det, None if scores is None else scores.ctypes.data,
                                        None if agg is None else agg.ctypes.data)
        if r < 0:
            raise _err()
        return (det, n_det, scores, agg) if want_scores else (det, n_det)

    def resample(self, pcm, sample_rate, channels=1):
        """pcm [S][n_samples*channels] (i8 / i16 / i32 / f32, interleaved) at sample_rate -> [S][n_out] f32 at 16 kHz."""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        n = N // channels
        fi, fo = resampler_frame_lengths(sample_rate)
        n_out = (n // fi) * fo
        out = np.empty((S, n_out), np.float32)
        if self._L.rp_resample_batch(self._h, pcm.ctypes.data, fmt, channels, sample_rate, S, n, N, out.ctypes.data, n_out) < 0:
            raise _err()
        return out

    def resample_dev(self, pcm_ptr, fmt, channels, sample_rate, S, n, stride, out_ptr, out_stride):
        if self._L.rp_resample_batch(self._h, pcm_ptr, fmt, channels, sample_rate, S, n, stride, out_ptr, out_stride) < 0:
            raise _err()

    def batch_detect_multi(self, pcm, templates, detector_config, thresholds=None, avg_thresholds=None, max_det=8):
        """Several wakewords in one detector: templates = [Templates, ...]; thresholds / avg_thresholds = per-wakeword
        overrides (None entries = the config's).  -> (det, det_wakeword, n_det)"""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        n = len(templates)
        hs = (C.c_void_p * n)(*[t._h for t in templates])
        def arr(v):
            if v is None:
                return None
            a = np.array([np.nan if x is None else x for x in v], np.float32)
            return a
        th, ath = arr(thresholds), arr(avg_thresholds)
        det = np.zeros((S, max_det), dtype=DET_DTYPE)
        dww = np.zeros((S, max_det), np.int32)
        n_det = np.zeros(S, np.int32)
        c = detector_config._c()
        fpt = C.POINTER(C.c_float)
        if self._L.rp_batch_detect_multi(self._h, pcm.ctypes.data, fmt, S, N, N, n, hs, C.byref(c),
                                         None if th is None else th.ctypes.data_as(fpt), None if ath is None else ath.ctypes.data_as(fpt),
                                         det.ctypes.data, dww.ctypes.data, n_det.ctypes.data, max_det) < 0:
            raise _err()
        return det, dww, n_det

    def batch_detect_model(self, pcm, model, mfcc_size, none_index, detector_config, precision="f32", max_det=8):
        """A wakeword model inside the batched detector -> (det, det_label, n_det)."""
        import numpy as np
        assert self.host
        pcm = np.ascontiguousarray(pcm)
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2}.get(pcm.dtype)
        if fmt is None:
            pcm, fmt = np.ascontiguousarray(pcm, np.float32), 3
        if pcm.ndim == 1:
            pcm = pcm[None, :]
        S, N = pcm.shape
        det = np.zeros((S, max_det), dtype=DET_DTYPE)
        dlab = np.zeros((S, max_det), np.int32)
        n_det = np.zeros(S, np.int32)
        c = detector_config._c()
        if self._L.rp_batch_detect_model(self._h, pcm.ctypes.data, fmt, S, N, N, model._h, mfcc_size, none_index, C.byref(c),
                                         MLP_PRECISION[precision], det.ctypes.data, dlab.ctypes.data, n_det.ctypes.data, max_det) < 0:
            raise _err()
        return det, dlab, n_det

    def batch_detect_dev(self, pcm_ptr, S, N, stride, templates, detector_config, det_ptr, n_det_ptr, max_det,
                         scores_ptr=None, agg_ptr=None):
        c = detector_config._c()
        if self._L.rp_batch_detect(self._h, pcm_ptr, S, N, stride, templates._h, C.byref(c), det_ptr, n_det_ptr, max_det,
                                   scores_ptr, agg_ptr) < 0:
            raise _err()

    def batch_detect_fmt_dev(self, pcm_ptr, fmt, S, N, stride, templates, detector_config, det_ptr, n_det_ptr, max_det,
                             scores_ptr=None, agg_ptr=None):
        """rp_batch_detect_fmt on device pointers: pcm in one of the reference's sample formats (SampleFormat: 1 = i16, 3 = f32)."""
        c = detector_config._c()
        if self._L.rp_batch_detect_fmt(self._h, pcm_ptr, int(fmt), S, N, stride, templates._h, C.byref(c), det_ptr, n_det_ptr, max_det,
                                       scores_ptr, agg_ptr) < 0:
            raise _err()

    def batch_detect_ingest(self, pcm, templates, detector_config, max_det=8, block_streams=0):
        """rp_batch_detect_ingest: pcm = HOST array [S][N] (numpy; page-locked or not) of f32 / i8 / i16 / i32 samples, taken in blocks
        with the copies under the kernels.  Returns (det [S][max_det], n_det [S], seconds)."""
        import numpy as np
        pcm = np.ascontiguousarray(pcm)
        S, N = pcm.shape
        fmt = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2, np.dtype(np.float32): 3}[pcm.dtype]
        det = np.zeros((S, max_det), dtype=[("stream", "<i4"), ("frame", "<i4"), ("window", "<i4"), ("counter", "<i4"), ("avg_score", "<f4"), ("score", "<f4")])
        n_det = np.zeros((S,), np.int32)
        sec = C.c_double(0.0)
        c = detector_config._c()
        if self._L.rp_batch_detect_ingest(self._h, pcm.ctypes.data, fmt, S, N, N, templates._h, C.byref(c), det.ctypes.data, n_det.ctypes.data, max_det,
                                          block_streams, C.byref(sec)) < 0:
            raise _err()
        return det, n_det, sec.value

    def batch_detect_ingest_ptr(self, pcm_host_ptr, fmt, S, N, stride, templates, detector_config, det_host_ptr, n_det_host_ptr, max_det, block_streams=0):
        """The same on raw HOST pointers (e.g. torch pinned tensors); returns the call's wall seconds."""
        sec = C.c_double(0.0)
        c = detector_config._c()
        if self._L.rp_batch_detect_ingest(self._h, pcm_host_ptr, int(fmt), S, N, stride, templates._h, C.byref(c), det_host_ptr, n_det_host_ptr, max_det,
                                          block_streams, C.byref(sec)) < 0:
            raise _err()
        return sec.value

    def mlp_forward(self, x, model, precision="f32"):
        import numpy as np
        assert self.host
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((x.shape[0], model.n_out), np.float32)
        if self._L.rp_mlp_forward_batch(self._h, model._h, x.ctypes.data, x.shape[0], MLP_PRECISION[precision],
                                        out.ctypes.data) < 0:
            raise _err()
        return out

    def mlp_forward_windows(self, mfcc, model, precision="f32"):
        """rp_mlp_forward_windows: mfcc [S][n_frames][K] -> logits [S][n_win][labels] of every window of the model's train_size frames."""
        import numpy as np
        assert self.host
        mfcc = np.ascontiguousarray(mfcc, np.float32)
        S, nf, K = mfcc.shape
        L = model.n_in // K
        n_win = max(nf - L + 1, 0)
        out = np.empty((S, n_win, model.n_out), np.float32)
        if self._L.rp_mlp_forward_windows(self._h, model._h, mfcc.ctypes.data, S, nf, K, MLP_PRECISION[precision], out.ctypes.data) < 0:
            raise _err()
        return out

    def mlp_dev(self, model, x_ptr, B, precision, out_ptr):
        if self._L.rp_mlp_forward_batch(self._h, model._h, x_ptr, B, MLP_PRECISION[precision], out_ptr) < 0:
            raise _err()

    def synth_pcm(self, seed, first_stream, S, N):
        import numpy as np
        assert self.host
        out = np.empty((S, N), np.float32)
        if self._L.rp_synth_pcm_batch(self._h, seed, first_stream, S, N, N, out.ctypes.data) < 0:
            raise _err()
        return out

    # --- raw device-pointer calls (host_pointers=False)
    def mfcc_dev(self, pcm_ptr, S, N, stride, K, out_ptr):
        if self._L.rp_mfcc_batch(self._h, pcm_ptr, S, N, stride, K, out_ptr) < 0:
            raise _err()

    def dtw_dev(self, mfcc_ptr, S, n_frames, templates, score_ref, band_size, score_mode, with_avg, scores_ptr, avg_ptr, agg_ptr):
        if self._L.rp_dtw_score_batch(self._h, mfcc_ptr, S, n_frames, templates._h, score_ref, band_size, int(score_mode),
                                      int(with_avg), scores_ptr, avg_ptr, agg_ptr) < 0:
            raise _err()

    def scan_dev(self, agg_ptr, avg_ptr, S, n_frames, max_len, detector_config, det_ptr, n_det_ptr, max_det):
        c = detector_config._c()
        if self._L.rp_detect_scan(self._h, agg_ptr, avg_ptr, S, n_frames, max_len, C.byref(c), 1 if avg_ptr else 0, None, 0,
                                  det_ptr, n_det_ptr, max_det) < 0:
            raise _err()

    def synth_dev(self, seed, first_stream, S, N, stride, out_ptr):
        if self._L.rp_synth_pcm_batch(self._h, seed, first_stream, S, N, stride, out_ptr) < 0:
            raise _err()

    def timing_enable(self, on=True):
        self._L.rp_ctx_timing_enable(self._h, 1 if on else 0)

    def timing_reset(self):
        self._L.rp_ctx_timing_reset(self._h)

    def timing_read(self, kernel):
        ms, n = C.c_double(), C.c_int()
        self._L.rp_ctx_timing_read(self._h, kernel, C.byref(ms), C.byref(n))
        return ms.value, n.value
