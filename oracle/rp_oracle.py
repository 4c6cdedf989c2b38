"""ctypes binding of oracle/rp_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of rp_oracle.c).  The product (rustpotter_amd/)
never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librp_oracle.so")

SCORE_MODES = {"average": 0, "max": 1, "median": 2, "p25": 3, "p50": 4, "p75": 5, "p80": 6, "p90": 7, "p95": 8}
MAX_T = 256


def build(force=False):
    # RP_ORACLE_SO points the checker at another build of the same source (e.g. an ASan/UBSan one, CPU only)
    global _SO
    if os.environ.get("RP_ORACLE_SO"):
        _SO = os.environ["RP_ORACLE_SO"]
        return _SO
    src = os.path.join(_HERE, "rp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class Detection(C.Structure):
    _fields_ = [
        ("name", C.c_int),
        ("wakeword", C.c_int),
        ("avg_score", C.c_float),
        ("score", C.c_float),
        ("n_scores", C.c_int),
        ("scores", C.c_float * MAX_T),
        ("counter", C.c_int),
        ("gain", C.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    L.orc_hamming_window.argtypes = [C.c_int, fp]
    L.orc_mel_filter_bank.argtypes = [C.c_int, C.c_int, C.c_int, fp, ip]
    L.orc_mfcc_new.restype = C.c_void_p
    L.orc_mfcc_new.argtypes = [C.c_int]
    L.orc_mfcc_free.argtypes = [C.c_void_p]
    L.orc_mfcc_reset.argtypes = [C.c_void_p]
    L.orc_mfcc_compute.restype = C.c_int
    L.orc_mfcc_compute.argtypes = [C.c_void_p, fp, C.c_int, fp]
    L.orc_mfcc_stream.restype = C.c_long
    L.orc_mfcc_stream.argtypes = [fp, C.c_long, C.c_int, fp]
    L.orc_normalize.argtypes = [fp, C.c_int, C.c_int, fp]
    L.orc_dtw_banded.restype = C.c_float
    L.orc_dtw_banded.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_int]
    L.orc_compare.restype = C.c_float
    L.orc_compare.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_score_window.restype = C.c_float
    L.orc_score_window.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_aggregate.restype = C.c_float
    L.orc_aggregate.argtypes = [fp, C.c_int, C.c_int]
    L.orc_average_step.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int]
    L.orc_mlp_forward.argtypes = [fp, C.c_long, C.c_int, ip, C.POINTER(fp), C.POINTER(fp), fp]
    L.orc_mlp_forward_bf16.argtypes = L.orc_mlp_forward.argtypes
    L.orc_mlp_train.restype = C.c_float
    L.orc_mlp_train.argtypes = [fp, ip, C.c_long, C.c_int, ip, C.POINTER(fp), C.POINTER(fp), C.c_float, C.c_int]
    L.orc_calc_inverse_similarity.restype = C.c_float
    L.orc_calc_inverse_similarity.argtypes = [C.c_float, C.c_float, C.c_float]
    L.orc_resampler_new.restype = C.c_void_p
    L.orc_resampler_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_resampler_free.argtypes = [C.c_void_p]
    L.orc_resampler_cutoff.restype = C.c_float
    L.orc_resampler_cutoff.argtypes = [C.c_int]
    L.orc_resampler_in_len.argtypes = [C.c_void_p]
    L.orc_resampler_out_len.argtypes = [C.c_void_p]
    L.orc_resampler_filter.argtypes = [C.c_void_p, fp, fp]
    L.orc_resampler_process.argtypes = [C.c_void_p, fp, fp]
    L.orc_resample_stream.restype = C.c_long
    L.orc_resample_stream.argtypes = [fp, C.c_long, C.c_int, fp]
    L.orc_detector_process_n.restype = C.c_int
    L.orc_detector_process_n.argtypes = [C.c_void_p, fp, C.c_int, C.POINTER(Detection)]
    L.orc_detector_process_resampled.restype = C.c_int
    L.orc_detector_process_resampled.argtypes = [C.c_void_p, C.c_void_p, fp, C.POINTER(Detection)]
    L.orc_rms_level.restype = C.c_float
    L.orc_rms_level.argtypes = [fp, C.c_int]
    L.orc_frontend_stream.argtypes = [fp, C.c_long, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_float, C.c_float, fp, fp, fp]
    L.orc_detector_new.restype = C.c_void_p
    L.orc_detector_new.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    L.orc_detector_free.argtypes = [C.c_void_p]
    L.orc_detector_reset.argtypes = [C.c_void_p]
    L.orc_detector_remove.argtypes = [C.c_void_p, C.c_int]
    L.orc_detector_remove.restype = C.c_int
    L.orc_detector_update_config.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orc_detector_update_config.restype = None
    L.orc_detector_update_filters.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    L.orc_detector_update_filters.restype = None
    L.orc_detector_add_ref.restype = C.c_int
    L.orc_detector_add_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, fp, C.c_int, fp, C.c_float, C.c_float, C.c_float]
    L.orc_detector_add_model.restype = C.c_int
    L.orc_detector_add_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp, C.c_float]
    L.orc_detector_process.restype = C.c_int
    L.orc_detector_process.argtypes = [C.c_void_p, fp, C.POINTER(Detection)]
    L.orc_detector_process_i16.restype = C.c_int
    L.orc_detector_process_i16.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.POINTER(Detection)]
    L.orc_detector_state.restype = C.c_int
    L.orc_detector_state.argtypes = [C.c_void_p, ip, ip, ip, fp]
    L.orc_synth_pcm.argtypes = [C.c_uint64, C.c_uint64, C.c_long, fp]
    L.orc_score_stream.restype = C.c_long
    L.orc_score_stream.argtypes = [fp, C.c_long, C.c_int, C.c_int, ip, fp, C.c_int, C.c_float, C.c_int, fp, fp]
    L.orc_bench.restype = C.c_double
    L.orc_bench.argtypes = [C.c_uint64, C.c_long, C.c_long, C.c_int, C.c_int, ip, fp, C.c_int, C.c_float, C.c_int,
                            C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.orc_sizeof_detection.restype = C.c_int
    assert L.orc_sizeof_detection() == C.sizeof(Detection)
    _lib = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ functional API
def hamming_window(n=480):
    out = np.empty(n, np.float32)
    lib().orc_hamming_window(n, _f(out))
    return out


def mel_filter_bank(K, nbins=240, sample_rate=16000):
    """Returns (bank [K+1][nbins], centres [K+3]) for out_size K (src/mfcc/extractor.rs:47-59)."""
    out = np.empty((K + 1, nbins), np.float32)
    cen = np.empty(K + 3, np.int32)
    lib().orc_mel_filter_bank(sample_rate, nbins, K + 1, _f(out), _i(cen))
    return out, cen


def dct_table(K):
    """cos table [K+1][K+1] (k,n) as the reference evaluates it in f32 (extractor.rs:146-163)."""
    nc = K + 1
    pi_over_n = np.float32(np.float32(3.14159274101257324) / np.float32(nc))
    t = np.empty((nc, nc), np.float32)
    for k in range(nc):
        for n in range(nc):
            arg = np.float32(np.float32(pi_over_n * np.float32(np.float32(n) + np.float32(0.5))) * np.float32(k))
            t[k, n] = np.cos(arg, dtype=np.float32)
    return t


def mfcc_stream(pcm, K):
    pcm = _c32(pcm)
    n_frames = max(0, 3 * (len(pcm) // 480) - 3)
    out = np.empty((max(n_frames, 1), K), np.float32)
    n = lib().orc_mfcc_stream(_f(pcm), len(pcm), K, _f(out))
    assert n == n_frames, (n, n_frames)
    return out[:n_frames]


def mlp_train(x, labels, weights, biases, lr, epochs):
    """Full-batch log_softmax / nll / SGD epochs on an MLP (training_loop of wakeword_model_train.rs); returns
    (weights, biases, last loss) -- the inputs are not modified."""
    x = _c32(x)
    labels = np.ascontiguousarray(labels, np.int32)
    ws = [np.array(w, np.float32, order="C") for w in weights]
    bs = [np.array(b, np.float32, order="C") for b in biases]
    dims = np.array([ws[0].shape[1]] + [w.shape[0] for w in ws], np.int32)
    n = len(ws)
    fp = C.POINTER(C.c_float)
    wp = (fp * n)(*[_f(w) for w in ws])
    bp = (fp * n)(*[_f(b) for b in bs])
    loss = lib().orc_mlp_train(_f(x), _i(labels), x.shape[0], n, _i(dims), wp, bp, lr, epochs)
    return ws, bs, loss


def wav_features(pcm_f32, sample_rate, K):
    """MfccWavFileExtractor::compute_mfccs for already decoded mono samples: resample if needed, MFCC, normalise."""
    y = _c32(pcm_f32) if sample_rate == 16000 else resample_stream(pcm_f32, sample_rate)
    return normalize(mfcc_stream(y[:(len(y) // 480) * 480], K))


class Resampler:
    """rubato FftFixedInOut<f32> as AudioEncoder::new builds it (src/audio/encoder.rs:72-83)."""

    def __init__(self, fs_in, fs_out=16000, chunk_size_in=480):
        self._h = lib().orc_resampler_new(fs_in, fs_out, chunk_size_in)
        if not self._h:
            raise ValueError("Unsupported sample rate, unable to initialize the resampler")
        self.in_len = lib().orc_resampler_in_len(self._h)
        self.out_len = lib().orc_resampler_out_len(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_resampler_free(self._h)
            self._h = None

    def filter_spectrum(self):
        re, im = np.empty(self.in_len + 1, np.float32), np.empty(self.in_len + 1, np.float32)
        lib().orc_resampler_filter(self._h, _f(re), _f(im))
        return re + 1j * im

    def process(self, chunk):
        c = _c32(chunk)
        assert c.shape == (self.in_len,)
        out = np.empty(self.out_len, np.float32)
        lib().orc_resampler_process(self._h, _f(c), _f(out))
        return out


def resample_stream(pcm, fs_in):
    """chunks_exact(input frame) -> resample -> concatenate (src/mfcc/wav_file_extractor.rs:83-96)."""
    pcm = _c32(pcm)
    r = Resampler(fs_in)
    out = np.empty((len(pcm) // r.in_len) * r.out_len + 1, np.float32)
    n = lib().orc_resample_stream(_f(pcm), len(pcm), fs_in, _f(out))
    return out[:n]


def normalize(m):
    m = _c32(m)
    out = np.empty_like(m)
    if m.shape[0]:
        lib().orc_normalize(_f(m), m.shape[0], m.shape[1], _f(out))
    return out


def dtw_banded(a, b, band=5):
    a, b = _c32(a), _c32(b)
    return float(lib().orc_dtw_banded(_f(a), a.shape[0], _f(b), b.shape[0], a.shape[1], band))


def compare(a, b, band=5, score_ref=0.22):
    a, b = _c32(a), _c32(b)
    return float(lib().orc_compare(_f(a), a.shape[0], _f(b), b.shape[0], a.shape[1], band, score_ref))


def score_window(window, templ, band=5, score_ref=0.22):
    window, templ = _c32(window), _c32(templ)
    return float(lib().orc_score_window(_f(window), window.shape[0], _f(templ), templ.shape[0], templ.shape[1], band, score_ref))


def aggregate(scores, mode):
    scores = _c32(scores)
    return float(lib().orc_aggregate(_f(scores), len(scores), SCORE_MODES[mode] if isinstance(mode, str) else mode))


def average_templates(named):
    """compute_avg_samples_features, src/wakewords/comp/wakeword_ref_build.rs:90-110."""
    if len(named) <= 1:
        return None
    items = sorted(named.items(), key=lambda kv: (-len(kv[1]), kv[0]))
    origin = _c32(items[0][1]).copy()
    for _, fr in items[1:]:
        fr = _c32(fr)
        lib().orc_average_step(_f(origin), origin.shape[0], _f(fr), fr.shape[0], origin.shape[1])
    return origin


def mlp_forward(x, weights, biases, bf16_layer1=False):
    """x [B][in]; weights list of [out][in]; biases list of [out].  bf16_layer1: round the layer-1
    inputs and weights to bf16 first (what the HIP bf16 MFMA path computes)."""
    x = _c32(x)
    ws = [_c32(w) for w in weights]
    bs = [_c32(b) for b in biases]
    dims = np.array([x.shape[1]] + [w.shape[0] for w in ws], np.int32)
    fp = C.POINTER(C.c_float)
    wp = (fp * len(ws))(*[_f(w) for w in ws])
    bp = (fp * len(bs))(*[_f(b) for b in bs])
    out = np.empty((x.shape[0], int(dims[-1])), np.float32)
    fn = lib().orc_mlp_forward_bf16 if bf16_layer1 else lib().orc_mlp_forward
    fn(_f(x), x.shape[0], len(ws), _i(dims), wp, bp, _f(out))
    return out


def calc_inverse_similarity(n1, n2, ref):
    return float(lib().orc_calc_inverse_similarity(n1, n2, ref))


def frontend_stream(pcm, gain_normalizer=False, gain_ref=None, min_gain=0.1, max_gain=1.0, rms_level_ref=float("nan"),
                    window_size=1, band_pass=False, low_cutoff=80.0, high_cutoff=400.0):
    """process_audio's front-end over a whole stream -> (filtered pcm, rms per chunk, gain per chunk)."""
    pcm = _c32(pcm)
    n = len(pcm)
    out = np.empty(n, np.float32)
    rms = np.empty(n // 480, np.float32)
    gains = np.empty(n // 480, np.float32)
    lib().orc_frontend_stream(_f(pcm), n, int(gain_normalizer), float("nan") if gain_ref is None else gain_ref, min_gain,
                              max_gain, rms_level_ref, window_size, int(band_pass), low_cutoff, high_cutoff, _f(out), _f(rms),
                              _f(gains))
    return out, rms, gains


def synth_pcm(seed, stream, N):
    out = np.empty(N, np.float32)
    lib().orc_synth_pcm(seed, stream, N, _f(out))
    return out


def pack_templates(templates):
    lens = np.array([len(t) for t in templates], np.int32)
    feats = np.concatenate([_c32(t).reshape(-1) for t in templates]).astype(np.float32)
    return lens, feats


def score_stream(mfcc, templates, band=5, score_ref=0.22, mode="max"):
    """Score[s][t] for every window start s (see orc_score_stream)."""
    mfcc = _c32(mfcc)
    lens, feats = pack_templates(templates)
    T, K = len(templates), mfcc.shape[1]
    n_win = max(0, mfcc.shape[0] - int(lens.max()) + 1)
    scores = np.empty((max(n_win, 1), T), np.float32)
    agg = np.empty(max(n_win, 1), np.float32)
    n = lib().orc_score_stream(_f(mfcc), mfcc.shape[0], K, T, _i(lens), _f(feats), band, score_ref,
                               SCORE_MODES[mode], _f(scores), _f(agg))
    assert n == n_win
    return scores[:n_win], agg[:n_win]


def synth_templates(seed, T, L, K):
    """T utterances of 480*ceil((L+3)/3) samples, seeds seed+1+t, stream 0,
    oracle MFCC + whole-matrix normalise (wav_file_extractor.rs:59-67), truncated to L frames."""
    n = 480 * -(-(L + 3) // 3)
    out = []
    for t in range(T):
        m = normalize(mfcc_stream(synth_pcm(seed + 1 + t, 0, n), K))
        assert m.shape[0] >= L
        out.append(m[:L].copy())
    return out


def bench(seed, S, N, templates, band=5, score_ref=0.22, mode="max", threads=1):
    lens, feats = pack_templates(templates)
    K = templates[0].shape[1]
    sc = C.c_long(0)
    cs = C.c_double(0)
    secs = lib().orc_bench(seed, S, N, K, len(templates), _i(lens), _f(feats), band, score_ref, SCORE_MODES[mode],
                           threads, C.byref(sc), C.byref(cs))
    return secs, sc.value, cs.value


class Detector:
    """Mirror of `Rustpotter` (src/detector.rs) for 16 kHz mono input."""

    def __init__(self, avg_threshold=0.2, threshold=0.5, min_scores=5, eager=False, score_ref=0.22, band_size=5,
                 score_mode="max", vad_mode=None, gain_normalizer=False, gain_ref=None, min_gain=0.1, max_gain=1.0,
                 band_pass=False, low_cutoff=80.0, high_cutoff=400.0):
        vad = {None: 0, "easy": 1, "medium": 2, "hard": 3}[vad_mode]
        self._h = lib().orc_detector_new(avg_threshold, threshold, min_scores, int(eager), score_ref, band_size,
                                         SCORE_MODES[score_mode], vad, int(gain_normalizer),
                                         float("nan") if gain_ref is None else gain_ref, min_gain, max_gain,
                                         int(band_pass), low_cutoff, high_cutoff)
        self.names = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_detector_free(self._h)
            self._h = None

    def add_ref(self, ww):
        """ww: dict with keys name, samples_features (ordered dict name->[L][K]), avg_features, threshold,
        avg_threshold, rms_level (as produced by tests/rpw_py.py)."""
        names = list(ww["samples_features"].keys())
        lens, feats = pack_templates([ww["samples_features"][n] for n in names])
        K = np.asarray(ww["samples_features"][names[0]]).shape[1]
        avg = ww.get("avg_features")
        avg_a = _c32(avg) if avg is not None else np.zeros((0, K), np.float32)
        nan = float("nan")
        r = lib().orc_detector_add_ref(self._h, len(names), K, _i(lens), _f(feats), avg_a.shape[0],
                                       _f(avg_a) if avg_a.shape[0] else None,
                                       nan if ww.get("threshold") is None else ww["threshold"],
                                       nan if ww.get("avg_threshold") is None else ww["avg_threshold"],
                                       ww.get("rms_level", 0.0))
        if r < 0:
            raise ValueError("Usage of wakewords with different mfcc size is not supported, ignoring wakeword")
        self.names.append((ww["name"], names))
        return r

    def add_model(self, model):
        """model: dict labels, train_size, mfcc_size, weights {lnX.weight/bias: ndarray}, rms_level."""
        n_layers = len([k for k in model["weights"] if k.endswith(".weight")])
        ws = [_c32(model["weights"]["ln%d.weight" % (i + 1)]) for i in range(n_layers)]
        bs = [_c32(model["weights"]["ln%d.bias" % (i + 1)]) for i in range(n_layers)]
        dims = np.array([ws[0].shape[1]] + [w.shape[0] for w in ws], np.int32)
        flat = np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)]) for w, b in zip(ws, bs)]).astype(np.float32)
        labels = model["labels"]
        none_index = labels.index("none") if "none" in labels else -1
        r = lib().orc_detector_add_model(self._h, model["train_size"], model["mfcc_size"], len(labels), none_index,
                                         n_layers, _i(dims), _f(flat), model.get("rms_level", float("nan")))
        if r < 0:
            raise ValueError("Usage of wakewords with different mfcc size is not supported, ignoring wakeword")
        self.names.append((None, labels))
        return r

    def _out(self, det):
        ww_name, names = self.names[det.wakeword]
        return {
            "name": ww_name if ww_name is not None else names[det.name],
            "avg_score": np.float32(det.avg_score),
            "score": np.float32(det.score),
            "scores": {names[i]: np.float32(det.scores[i]) for i in range(det.n_scores)},
            "counter": det.counter,
            "gain": np.float32(det.gain),
        }

    def process_f32(self, samples480):
        s = _c32(samples480)
        assert s.shape == (480,)
        det = Detection()
        if lib().orc_detector_process(self._h, _f(s), C.byref(det)):
            return self._out(det)
        return None

    def process_chunk(self, samples):
        """One encoded 16 kHz chunk of any length (what process_audio sees behind the resampler)."""
        s = _c32(samples)
        det = Detection()
        if lib().orc_detector_process_n(self._h, _f(s), len(s), C.byref(det)):
            return self._out(det)
        return None

    def process_resampled(self, resampler, samples):
        """One input frame at the resampler's input rate (process_samples::<f32> with fmt.sample_rate != 16000)."""
        s = _c32(samples)
        assert s.shape == (resampler.in_len,)
        det = Detection()
        if lib().orc_detector_process_resampled(self._h, resampler._h, _f(s), C.byref(det)):
            return self._out(det)
        return None

    def process_i16(self, samples480):
        s = np.ascontiguousarray(samples480, dtype=np.int16)
        assert s.shape == (480,)
        det = Detection()
        if lib().orc_detector_process_i16(self._h, s.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(det)):
            return self._out(det)
        return None

    def remove(self, index):
        """Rustpotter::remove_wakeword (src/detector.rs:180-189) of the index-th wakeword (insertion order)."""
        return bool(lib().orc_detector_remove(self._h, index))  # self.names is indexed by the wakeword's uid: nothing to delete

    def update_detector_config(self, avg_threshold, threshold, min_scores, eager, score_ref, band_size, score_mode, vad_mode):
        """Rustpotter::update_detector_config (src/detector.rs:262-280)."""
        lib().orc_detector_update_config(self._h, avg_threshold, threshold, min_scores, int(eager), score_ref, band_size,
                                         SCORE_MODES[score_mode], {None: 0, "easy": 1, "medium": 2, "hard": 3}[vad_mode])

    def update_filters_config(self, gain_normalizer, gain_ref, min_gain, max_gain, band_pass, low_cutoff, high_cutoff):
        """Rustpotter::update_filters_config (src/detector.rs:284-288)."""
        lib().orc_detector_update_filters(self._h, int(gain_normalizer), float("nan") if gain_ref is None else gain_ref, min_gain,
                                          max_gain, int(band_pass), low_cutoff, high_cutoff)

    def reset(self):
        """Rustpotter::reset (src/detector.rs:290-302)."""
        lib().orc_detector_reset(self._h)

    def state(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        mx = lib().orc_detector_state(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"max_mfcc_frames": mx, "window_len": a.value, "countdown": b.value, "partial_counter": c.value,
                "partial_score": d.value}
