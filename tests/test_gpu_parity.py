"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same inputs, and against the reference's golden fixtures.

Tolerances (BASELINE.json north_star / SURVEY.md §8d):
  MFCC   |d| <= 1e-5 * max(|ref|, 1)
  scores relative 1e-5
  detection index / counter / countdown logic: exact
"""
import json
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

G = simstream.GOLDEN
EXP = json.load(open(os.path.join(G, "expectations.json")))
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


def mfcc_close(got, ref):
    """SURVEY §8d gate, element-wise: |d| <= 1e-5 * max(|ref|, 1)."""
    return np.all(np.abs(got - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0))


def mfcc_close_framescale(got, ref):
    """1e-5 relative to the frame's largest coefficient.  For mfcc_size >= 16 the
    element-wise gate is tighter than f32 itself allows: the DCT sums reach |60| (ulp
    3.8e-6) while small coefficients are ~1, and the oracle is just as far (1.6e-5) from
    an f64 evaluation as the kernel is (1.8e-5) -- tools/probe_mfcc_err.py."""
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1.0)
    return np.all(np.abs(got - ref) <= 1e-5 * scale)


def rel_close(got, ref, rtol=1e-5):
    return np.all(np.abs(got - ref) <= rtol * np.abs(ref))


def test_synth_generator_bit_exact(ctx):
    pcm = ctx.synth_pcm(SEED, 7, 3, 1000)
    for s in range(3):
        assert np.array_equal(pcm[s], orc.synth_pcm(SEED, 7 + s, 1000))


@pytest.mark.parametrize("K", [5, 16, 13, 1, 23])
def test_mfcc_synthetic_streams(ctx, K):
    S, N = 5, 480 * 37 + 123  # ragged tail: the last 123 samples are ignored like chunks_exact does
    pcm = np.stack([orc.synth_pcm(SEED, s, N) for s in range(S)])
    got = ctx.mfcc(pcm, K)
    assert got.shape == (S, 3 * 37 - 3, K)
    for s in range(S):
        ref = orc.mfcc_stream(pcm[s], K)
        assert mfcc_close_framescale(got[s], ref)
        if K <= 5:
            assert mfcc_close(got[s], ref)


def test_mfcc_fixture_wavs(ctx):
    for name in ["oye_casa_g_1.wav", "oye_casa_g_5.wav", "alexa2.wav"]:
        pcm, _ = rpw_py.read_wav_i16(os.path.join(G, name))
        x = simstream.i16_to_f32(pcm)
        got = ctx.mfcc(x, 5)[0]
        ref = orc.mfcc_stream(x, 5)
        assert got.shape == ref.shape and mfcc_close(got, ref)
    # golden: normalised GPU MFCC == matrices the reference stored in its .rpw (G1)
    w = rpw_py.load_rpw(os.path.join(G, "alexa.rpw"))
    for name, ref in w["samples_features"].items():
        pcm, _ = rpw_py.read_wav_i16(os.path.join(G, name))
        got = orc.normalize(ctx.mfcc(simstream.i16_to_f32(pcm), 5)[0])
        assert mfcc_close(got, ref)


@pytest.mark.parametrize("dtype,scale", [(np.int16, 32767.0), (np.int8, 127.0), (np.int32, 2147483648.0)])
def test_mfcc_integer_samples_decoded_on_device(ctx, dtype, scale):
    """`v as f32 / T::MAX as f32` (src/audio/audio_types.rs:98-127) inside the kernel == decoding first."""
    info = np.iinfo(dtype)
    rng = np.random.default_rng(4)
    for n in (480 * 21, 480 * 21 + 2):  # aligned rows -> vector loads; odd row length -> scalar loads
        raw = rng.integers(info.min, info.max, size=(3, n), dtype=dtype, endpoint=True)
        raw[0, :5] = [info.min, info.max, 0, -1, 1]
        got = ctx.mfcc(raw, 5)
        ref = ctx.mfcc(raw.astype(np.float32) / np.float32(scale), 5)
        assert np.array_equal(got, ref)
    pcm, _ = rpw_py.read_wav_i16(os.path.join(G, "alexa.wav"))
    assert mfcc_close(ctx.mfcc(pcm, 5)[0], orc.mfcc_stream(simstream.i16_to_f32(pcm), 5))


def test_mfcc_edge_sizes(ctx):
    assert ctx.mfcc(np.zeros((2, 479), np.float32), 5).shape == (2, 0, 5)      # no full chunk
    assert ctx.mfcc(np.zeros((2, 480), np.float32), 5).shape == (2, 0, 5)      # one chunk: extractor only fills
    x = orc.synth_pcm(SEED, 3, 960)
    got = ctx.mfcc(x, 5)
    assert got.shape == (1, 3, 5) and mfcc_close(got[0], orc.mfcc_stream(x, 5))


def test_mfcc_silence_is_bitwise_constant(ctx):
    """Digital silence must give identical frames that normalise to exactly 0."""
    got = ctx.mfcc(np.zeros((1, 480 * 50), np.float32), 5)[0]
    ref = orc.mfcc_stream(np.zeros(480 * 50, np.float32), 5)
    assert np.all(got == got[0])
    assert np.all(orc.normalize(got[:100]) == 0.0)
    assert mfcc_close(got, ref)


def _sim_mfcc():
    s = simstream.simulation_stream_i16()
    return orc.mfcc_stream(simstream.i16_to_f32(s), 5)


@pytest.mark.parametrize("rpw", ["alexa.rpw", "oye_casa_g.rpw"])
def test_dtw_scores_fixture_stream(ra, ctx, rpw):
    """BASELINE config C1: the reference's detector simulation stream against its own templates."""
    w = rpw_py.load_rpw(os.path.join(G, rpw))
    templates = list(w["samples_features"].values())
    mf = _sim_mfcc()
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    scores, avg, agg = ctx.dtw_scores(mf, tm, with_avg=True)
    ref_s, ref_a = orc.score_stream(mf, templates)
    assert scores.shape[1:] == ref_s.shape
    assert rel_close(scores[0], ref_s) and rel_close(agg[0], ref_a)
    Lmax = max(len(t) for t in templates)
    ref_avg = np.array([orc.score_window(mf[s:s + Lmax], w["avg_features"]) for s in range(0, ref_s.shape[0], 7)])
    assert rel_close(avg[0][::7], ref_avg)


@pytest.mark.parametrize("mode", ["average", "max", "median", "p25", "p50", "p75", "p80", "p90", "p95"])
def test_score_modes(ra, ctx, mode):
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    mf = _sim_mfcc()[480:900]
    tm = ra.Templates(ctx, templates)
    sm = {"average": ra.ScoreMode.Average, "max": ra.ScoreMode.Max, "median": ra.ScoreMode.Median, "p25": ra.ScoreMode.P25,
          "p50": ra.ScoreMode.P50, "p75": ra.ScoreMode.P75, "p80": ra.ScoreMode.P80, "p90": ra.ScoreMode.P90,
          "p95": ra.ScoreMode.P95}[mode]
    scores, _, agg = ctx.dtw_scores(mf, tm, score_mode=sm)
    # aggregate of the GPU's own scores must equal the oracle's aggregation bit for bit ...
    ref_from_gpu = np.array([orc.aggregate(scores[0][i], mode) for i in range(scores.shape[1])], np.float32)
    assert np.array_equal(agg[0], ref_from_gpu)
    # ... and the end-to-end value is within tolerance of the oracle path
    _, ref_a = orc.score_stream(mf, templates, mode=mode)
    assert rel_close(agg[0], ref_a)


@pytest.mark.parametrize("K,band,L", [(5, 5, 100), (5, 3, 64), (5, 4, 80), (5, 6, 90), (16, 5, 50), (5, 9, 30), (16, 3, 40), (3, 1, 12),
                                      (13, 5, 60), (13, 3, 40), (13, 6, 70), (13, 4, 33), (16, 4, 45), (16, 6, 48), (13, 7, 30), (12, 5, 30)])
def test_dtw_synthetic(ra, ctx, K, band, L):
    """Register kernels (mfcc_size 5 at band 3..6, one- and multi-template chunks; mfcc_size 13 / 16 at band 3..6) and the
    generic kernel (everything else) on BASELINE-style synthetic input."""
    T, S, N = 3, 4, 480 * 60
    templates = orc.synth_templates(SEED, T, L, K)
    templates[1] = templates[1][: L - 7].copy()  # ragged template lengths
    pcm = np.stack([orc.synth_pcm(SEED, s, N) for s in range(S)])
    mf = np.stack([orc.mfcc_stream(pcm[s], K) for s in range(S)])
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm, band_size=band)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates, band=band)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)


def test_dtw_many_templates_mixed_chunks(ra, ctx):
    """11 equal-length + 3 ragged templates + avg over 3 streams: chunk classes 8/4/2, several chunks,
    flattened tiles straddling streams, avg chunk."""
    K, L = 5, 70
    templates = orc.synth_templates(SEED, 14, L, K)
    templates[3] = templates[3][:60].copy()
    templates[7] = templates[7][:60].copy()
    templates[12] = templates[12][:41].copy()
    avg = orc.synth_templates(SEED + 77, 1, L, K)[0]
    S, N = 3, 480 * 60
    pcm = np.stack([orc.synth_pcm(SEED, 20 + s, N) for s in range(S)])
    mf = np.stack([orc.mfcc_stream(pcm[s], K) for s in range(S)])
    tm = ra.Templates(ctx, templates, avg=avg)
    scores, avg_s, agg = ctx.dtw_scores(mf, tm, with_avg=True, score_mode=ra.ScoreMode.P75)
    assert scores.shape == (S, 177 - L + 1, 14)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates, mode="p75")
        assert rel_close(scores[s], ref_s) and rel_close(agg[s], ref_a)
        ref_avg = np.array([orc.score_window(mf[s][w:w + L], avg) for w in range(0, scores.shape[1], 5)])
        assert rel_close(avg_s[s][::5], ref_avg)


def test_single_template_and_tiny_batch(ra, ctx):
    K = 5
    templates = orc.synth_templates(SEED, 1, 30, K)
    mf = orc.mfcc_stream(orc.synth_pcm(SEED, 2, 480 * 14), K)  # 39 frames -> 10 windows
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm)
    ref_s, ref_a = orc.score_stream(mf, templates)
    assert scores.shape == (1, 10, 1) and rel_close(scores[0], ref_s) and np.array_equal(agg[0], scores[0][:, 0])


def test_single_stream_call_latency(ra):
    """The drop-in API must keep up with real time: one process_samples call per 30 ms of audio."""
    import time
    c = ra.RustpotterConfig.default()
    c.detector.avg_threshold = 0.0
    rp = ra.Rustpotter.new(c)
    rp.add_wakeword_from_file("w", os.path.join(G, "oye_casa_g.rpw"))
    pcm = orc.synth_pcm(SEED, 5, 480 * 400) * np.float32(0.1)
    for i in range(0, 480 * 150, 480):
        rp.process_samples(pcm[i:i + 480].copy())
    t0 = time.perf_counter()
    n = 0
    for i in range(480 * 150, 480 * 400, 480):
        rp.process_samples(pcm[i:i + 480].copy())
        n += 1
    per_call = (time.perf_counter() - t0) / n
    print("single-stream process_samples: %.1f us per 30 ms chunk" % (per_call * 1e6))
    assert per_call < 0.030 / 10  # at least 10x faster than real time


def test_dtw_avg_longer_than_window(ra, ctx):
    """m != n: an averaged template longer than every sample template widens the band to |m-n|."""
    K = 5
    templates = orc.synth_templates(SEED, 2, 40, K)
    avg = orc.synth_templates(SEED + 9, 1, 46, K)[0]
    mf = orc.mfcc_stream(orc.synth_pcm(SEED, 1, 480 * 40), K)
    tm = ra.Templates(ctx, templates, avg=avg)
    scores, avg_s, _ = ctx.dtw_scores(mf, tm, with_avg=True)
    ref = np.array([orc.score_window(mf[s:s + 40], avg) for s in range(scores.shape[1])])
    assert rel_close(avg_s[0], ref)
    ref_s, _ = orc.score_stream(mf, templates)
    assert rel_close(scores[0], ref_s)


def test_dtw_silence_windows(ra, ctx):
    """All-zero normalised windows: every cell costs exactly 1 (comparator.rs:43-44)."""
    w = rpw_py.load_rpw(os.path.join(G, "alexa.rpw"))
    templates = list(w["samples_features"].values())
    mf = orc.mfcc_stream(np.zeros(480 * 60, np.float32), 5)
    tm = ra.Templates(ctx, templates)
    scores, _, _ = ctx.dtw_scores(mf, tm)
    ref_s, _ = orc.score_stream(mf, templates)
    assert np.array_equal(scores[0], ref_s)


def test_dtw_too_short_stream(ra, ctx):
    templates = orc.synth_templates(SEED, 2, 30, 5)
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(np.zeros((1, 29, 5), np.float32), tm)
    assert scores.shape == (1, 0, 2) and agg.shape == (1, 0)


def _oracle_detections(e, s):
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    d = orc.Detector(avg_threshold=e["avg_threshold"], threshold=e["threshold"], min_scores=e.get("min_scores", 5),
                     score_mode=e["score_mode"], vad_mode=e.get("vad_mode"),
                     gain_normalizer=e.get("gain_normalizer", False), band_pass=e.get("band_pass", False),
                     low_cutoff=e.get("low_cutoff", 80.0), high_cutoff=e.get("high_cutoff", 400.0))
    d.add_ref(w)
    out = []
    for i in range(0, len(s) - 479, 480):
        r = d.process_i16(s[i:i + 480])
        if r is not None:
            out.append((i // 480, r))
    return out


def _make_config(ra, e):
    c = ra.RustpotterConfig.default()
    c.fmt.sample_rate, c.fmt.sample_format, c.fmt.channels = 16000, ra.SampleFormat.I16, 1
    c.detector.avg_threshold, c.detector.threshold = e["avg_threshold"], e["threshold"]
    c.detector.min_scores = e.get("min_scores", 5)
    c.detector.score_mode = {"max": ra.ScoreMode.Max, "median": ra.ScoreMode.Median, "average": ra.ScoreMode.Average}[e["score_mode"]]
    c.detector.vad_mode = {None: None, "easy": ra.VADMode.Easy}[e.get("vad_mode")]
    c.filters.gain_normalizer.enabled = e.get("gain_normalizer", False)
    c.filters.band_pass.enabled = e.get("band_pass", False)
    c.filters.band_pass.low_cutoff = e.get("low_cutoff", 80.0)
    c.filters.band_pass.high_cutoff = e.get("high_cutoff", 400.0)
    return c


@pytest.mark.parametrize("case", sorted(EXP["simulation"].keys()))
def test_rustpotter_api_goldens(ra, case):
    """tests/detector.rs through the drop-in API (process_bytes, 30 ms chunks) on the GPU:
    same detections as the reference asserts, same chunk / counter as the oracle."""
    e = EXP["simulation"][case]
    s = simstream.simulation_stream_i16(*e.get("gains", [1.0, 1.0]))
    rp = ra.Rustpotter.new(_make_config(ra, e))
    rp.add_wakeword_from_file("wakeword", os.path.join(G, e["rpw"]))
    raw = s.astype("<i2").tobytes()
    bpf = rp.get_bytes_per_frame()
    assert bpf == 960 and rp.get_samples_per_frame() == 480
    dets = []
    for i in range(0, len(raw) - bpf + 1, bpf):
        d = rp.process_bytes(raw[i:i + bpf])
        if d is not None:
            dets.append((i // bpf, d))
    ref = _oracle_detections(e, s)
    assert len(dets) == len(e["detections"]) == len(ref)
    for (chunk, d), (rchunk, r), (gavg, gscore) in zip(dets, ref, e["detections"]):
        assert chunk == rchunk and d.counter == r["counter"]          # index / countdown logic exact
        assert abs(d.score - r["score"]) <= 1e-5 * abs(r["score"])
        assert abs(d.score - np.float32(gscore)) <= 1e-5 * gscore      # the value tests/detector.rs asserts
        if gavg is not None:
            assert abs(d.avg_score - np.float32(gavg)) <= 1e-5 * gavg
        assert d.name == r["name"] and set(d.scores) == set(r["scores"])
        for k in d.scores:
            assert abs(d.scores[k] - r["scores"][k]) <= 1e-5 * abs(r["scores"][k])


def test_rustpotter_api_behaviour(ra):
    c = ra.RustpotterConfig.default()
    rp = ra.Rustpotter.new(c)
    assert rp.process_samples(np.zeros(480, np.float32)) is None            # no wakewords -> None (detector.rs:348)
    rp.add_wakeword_from_file("a", os.path.join(G, "alexa.rpw"))
    assert rp.process_samples(np.zeros(100, np.float32)) is None            # wrong length -> None (:249)
    with pytest.raises(ra.RustpotterError, match="different mfcc size"):
        rp.add_wakeword_from_file("m", os.path.join(G, "ok_casa-tiny.rpw"))  # K=16 vs 5 (:308-320)
    with pytest.raises(ra.RustpotterError, match="Unable to open file"):
        rp.add_wakeword_from_file("x", os.path.join(G, "missing.rpw"))
    assert rp.remove_wakeword("a") and not rp.remove_wakeword("a") and not rp.remove_wakewords()
    c.fmt.sample_rate = 48000
    assert ra.Rustpotter.new(c).get_samples_per_frame() == 1440  # resampled input: 30 ms frames at the input rate
    c.fmt.sample_rate = 0
    with pytest.raises(ra.RustpotterError, match="Unsupported sample rate"):
        ra.Rustpotter.new(c)


def test_batch_scan_matches_streaming_state_machine(ra, ctx):
    """rp_detect_scan over precomputed scores == the chunked detector (frames, counters exact)."""
    for case in ("max", "median", "ignore_alexa", "vad_easy"):
        e = EXP["simulation"][case]
        w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
        templates = list(w["samples_features"].values())
        s = simstream.simulation_stream_i16()
        mf = ctx.mfcc(s, 5)  # i16 samples decoded on the device
        tm = ra.Templates(ctx, templates, avg=w["avg_features"])
        cfg = _make_config(ra, e).detector
        with_avg = e["avg_threshold"] != 0.0
        _, avg, agg = ctx.dtw_scores(mf, tm, score_mode=cfg.score_mode, with_avg=with_avg)
        det, n_det = ctx.detect_scan(agg, avg if with_avg else None, mf.shape[1], tm.max_len, cfg,
                                     mfcc=mf if e.get("vad_mode") else None)
        ref = _oracle_detections(e, s)
        assert n_det[0] == len(ref)
        for i, (chunk, r) in enumerate(ref):
            assert det[0][i]["frame"] // 3 + 1 == chunk and det[0][i]["counter"] == r["counter"]
            assert abs(det[0][i]["score"] - r["score"]) <= 1e-5 * r["score"]


def test_batch_detect_equals_per_stream_api(ra, ctx):
    """rp_batch_detect over S streams == S independent Rustpotter handles fed chunk by chunk."""
    e = EXP["simulation"]["max"]
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    templates = list(w["samples_features"].values())
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    rng = np.random.default_rng(3)
    streams = [base, np.roll(base, 480 * 7), (base + rng.standard_normal(len(base)).astype(np.float32) * np.float32(0.002))]
    n = (len(base) // 480) * 480
    pcm = np.stack([s[:n] for s in streams])
    cfg = _make_config(ra, e)
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    det, n_det, scores, agg = ctx.batch_detect(pcm, tm, cfg.detector, want_scores=True)
    cfg.fmt.sample_format = ra.SampleFormat.F32
    for si in range(len(streams)):
        rp = ra.Rustpotter.new(cfg)
        rp.add_wakeword_from_file("w", os.path.join(G, e["rpw"]))
        got = []
        for i in range(0, n, 480):
            d = rp.process_samples(pcm[si, i:i + 480].copy())
            if d is not None:
                got.append((i // 480, d))
        assert n_det[si] == len(got) and len(got) >= 1
        for j, (chunk, d) in enumerate(got):
            assert det[si][j]["frame"] // 3 + 1 == chunk and det[si][j]["counter"] == d.counter
            assert abs(det[si][j]["score"] - d.score) <= 1e-6 * d.score
            names = list(w["samples_features"].keys())
            row = scores[si][det[si][j]["window"]]
            for ti, nm in enumerate(names):
                assert abs(row[ti] - d.scores[nm]) <= 1e-6 * d.scores[nm]


def _det_tuple(d):
    return (int(d["frame"]), int(d["window"]), int(d["counter"]), float(d["score"]), float(d["avg_score"]))


@pytest.mark.parametrize("case,chunks_per_call", [("max", 1), ("max", 4), ("median", 3), ("ignore_alexa", 2), ("vad_easy", 1), ("vad_easy", 5)])
def test_stream_batch_equals_offline_batch(ra, ctx, case, chunks_per_call):
    """rp_stream_batch_process fed a few chunks per call (state carried on the device) emits, chunk for
    chunk, what rp_batch_detect finds over the whole stream -- which the tests above tie to the oracle's
    chunked detector.  Detections must come out in the call that contains their frame."""
    e = EXP["simulation"][case]
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    rng = np.random.default_rng(5)
    streams = [base[:n], np.roll(base[:n], 480 * 11), np.roll(base[:n], -480 * 4 - 77)]
    streams += [(s.astype(np.int32) + rng.integers(-30, 30, n)).clip(-32768, 32767).astype(np.int16) for s in streams]
    pcm = np.stack(streams)
    cfg = _make_config(ra, e).detector
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    det, n_det, _, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    assert n_det.sum() >= (0 if case == "ignore_alexa" else len(streams))
    sb = ra.StreamBatch(ctx, tm, cfg, len(streams), max_chunks_per_call=chunks_per_call)
    got = [[] for _ in streams]
    L = tm.max_len
    step = 480 * chunks_per_call
    for i in range(0, n, step):
        piece = pcm[:, i:i + step]
        d, nd, a = sb.process(piece, want_agg=True)
        f0 = 3 * (i // 480) - 3
        for si in range(len(streams)):
            for j in range(nd[si]):
                assert f0 <= d[si][j]["frame"] < f0 + 3 * (piece.shape[1] // 480) and d[si][j]["stream"] == si
                got[si].append(_det_tuple(d[si][j]))
            # aggregate scores of the windows that end in this call, bitwise the offline ones
            for k in range(a.shape[1]):
                wi = f0 + k - L + 1
                if 0 <= wi < agg.shape[1]:
                    assert a[si, k] == agg[si, wi]
    assert sb.chunks_seen == n // 480
    for si in range(len(streams)):
        assert got[si] == [_det_tuple(det[si][j]) for j in range(n_det[si])]


def test_stream_batch_reset_and_formats(ra, ctx):
    """reset(stream) == Rustpotter::reset on that stream only: compared with per-stream Rustpotter handles
    that are reset at the same chunk; f32 input, one chunk per call."""
    e = EXP["simulation"]["max"]
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    templates = list(w["samples_features"].values())
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], base[:n], np.roll(base[:n], 480 * 9)])
    cfg = _make_config(ra, e)
    cfg.fmt.sample_format = ra.SampleFormat.F32
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    # where does stream 0 fire first?  reset stream 1 in the middle of that partial detection
    det, n_det = ctx.batch_detect(pcm, tm, cfg.detector)
    assert n_det[0] >= 1
    reset_chunk = det[0][0]["frame"] // 3 + 1 - 6
    sb = ra.StreamBatch(ctx, tm, cfg.detector, 3)
    rps = []
    for _ in range(3):
        rp = ra.Rustpotter.new(cfg)
        rp.add_wakeword_from_file("w", os.path.join(G, e["rpw"]))
        rps.append(rp)
    got, ref = [[] for _ in range(3)], [[] for _ in range(3)]
    for c in range(n // 480):
        if c == reset_chunk:
            sb.reset(1)
            rps[1].reset()
        if c == reset_chunk + 40:
            sb.reset()
            for rp in rps:
                rp.reset()
        d, nd = sb.process(pcm[:, c * 480:(c + 1) * 480])
        for si in range(3):
            for j in range(nd[si]):
                got[si].append((c, int(d[si][j]["counter"]), float(d[si][j]["score"])))
            r = rps[si].process_samples(pcm[si, c * 480:(c + 1) * 480].copy())
            if r is not None:
                ref[si].append((c, r.counter, r.score))
    assert len(ref[0]) >= 1
    for si in range(3):
        assert len(got[si]) == len(ref[si])
        for g, r in zip(got[si], ref[si]):
            assert g[0] == r[0] and g[1] == r[1] and abs(g[2] - r[2]) <= 1e-6 * r[2]
    assert got[0] != got[1]  # the reset changed stream 1's detections
    with pytest.raises(ra.RustpotterError):
        sb.process(np.zeros((3, 480 * 2), np.float32))  # more chunks than max_chunks_per_call


def test_stream_batch_48k_stereo_equals_resample_then_offline(ra, ctx):
    """Live 48 kHz stereo i16 streams, two chunks per call: the per-call resampler (previous input frame kept per
    stream) + the streaming path == rp_resample_batch over the whole stream followed by rp_batch_detect."""
    e = EXP["audio_file"]["noise"]
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    pcm48, sr, _ = rpw_py.read_wav(os.path.join(G, e["wav"]))
    n = (len(pcm48) // 2880) * 2880
    left = np.round(np.clip(pcm48[:n] / 4.0, -1.0, 1.0) * 32767.0).astype(np.int16)  # the recording peaks at 2.2
    streams = np.stack([left, np.roll(left, 1440 * 9), (left // 2).astype(np.int16)])
    right = np.zeros_like(streams) + 123
    inter = np.stack([streams, right], axis=2).reshape(3, -1)           # L R L R ...
    cfg = _make_config(ra, e).detector
    cfg.min_scores = e["min_scores"]
    tm = ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])
    mono16 = ctx.resample(inter, 48000, channels=2)
    det, n_det, _, agg = ctx.batch_detect(mono16, tm, cfg, want_scores=True)
    assert n_det[0] >= 2
    sb = ra.StreamBatch(ctx, tm, cfg, 3, max_chunks_per_call=2, sample_rate=48000, channels=2)
    assert sb.samples_per_chunk == 2880
    got = [[] for _ in range(3)]
    step = 2 * 2880
    for i in range(0, inter.shape[1], step):
        d, nd, a = sb.process(inter[:, i:i + step], want_agg=True)
        f0 = 3 * (i // 2880) - 3
        for si in range(3):
            for j in range(nd[si]):
                got[si].append(_det_tuple(d[si][j]))
            for k in range(a.shape[1]):
                wi = f0 + k - tm.max_len + 1
                if 0 <= wi < agg.shape[1]:
                    assert a[si, k] == agg[si, wi]
    for si in range(3):
        assert got[si] == [_det_tuple(det[si][j]) for j in range(n_det[si])]
    # 22.05 kHz input: 882-sample frames -> 640 encoded samples = four MFCC frames per input frame
    sb22 = ra.StreamBatch(ctx, tm, cfg, 3, sample_rate=22050)
    assert sb22.samples_per_chunk == 882 and sb22.frames_per_chunk == 4


def test_stream_batch_wide_templates_generic_tiling(ra, ctx):
    """mfcc_size 16 and band 3: the streaming path outside the cross-stream tiling (few windows per stream per call
    through the per-stream tiles) still equals the offline pass."""
    for K, band in ((16, 5), (5, 3)):
        S, N = 96, 480 * 60
        tmpl = orc.synth_templates(SEED, 4, 40, K)
        tm = ra.Templates(ctx, tmpl)
        cfg = ra.RustpotterConfig.default().detector
        cfg.avg_threshold, cfg.threshold, cfg.min_scores, cfg.band_size = 0.0, 0.2, 1, band
        pcm = ctx.synth_pcm(SEED, 0, S, N)
        det, n_det, _, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
        sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=3)
        total = np.zeros(S, np.int64)
        for i in range(0, N, 1440):
            d, nd, a = sb.process(pcm[:, i:i + 1440], want_agg=True)
            f0 = 3 * (i // 480) - 3
            for k in range(9):
                wi = f0 + k - 39
                if 0 <= wi < agg.shape[1]:
                    assert np.array_equal(a[:, k], agg[:, wi])
            for si in np.nonzero(nd)[0]:
                for j in range(nd[si]):
                    assert _det_tuple(d[si][j]) == _det_tuple(det[si][total[si] + j])
            total += nd
        assert np.array_equal(total, n_det)


def test_stream_batch_many_streams_synthetic(ra, ctx):
    """4096 synthetic streams, 2 chunks per call (the cross-stream DTW tiling): per-call aggregates and
    detections identical to the offline pass."""
    S, N = 4096, 16000
    tmpl = orc.synth_templates(SEED, 8, 100, 5)
    tm = ra.Templates(ctx, tmpl)
    cfg = ra.RustpotterConfig.default().detector
    cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.0, 0.2, 1
    pcm = ctx.synth_pcm(SEED, 0, S, N)
    n = (N // 960) * 960
    det, n_det, _, agg = ctx.batch_detect(pcm[:, :n], tm, cfg, want_scores=True)
    sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=2)
    total = np.zeros(S, np.int64)
    for i in range(0, n, 960):
        d, nd, a = sb.process(pcm[:, i:i + 960], want_agg=True)
        f0 = 3 * (i // 480) - 3
        lo, hi = f0 - 99, f0 + 6 - 99  # windows ending at the 6 new frames
        k0 = max(0, -lo)
        if hi > 0:
            assert np.array_equal(a[:, k0:], agg[:, lo + k0:hi])
        for si in np.nonzero(nd)[0]:
            for j in range(nd[si]):
                assert _det_tuple(d[si][j]) == _det_tuple(det[si][total[si] + j])
        total += nd
    assert np.array_equal(total, n_det)


def test_allocation_failure_is_an_error_and_does_not_stick(ra, ctx):
    """A batch too large for the device is refused with the runtime's message, and the failure is not reported again by
    the next (valid) call: hipGetLastError() keeps failed calls' codes until they are read."""
    tm = ra.Templates(ctx, orc.synth_templates(SEED, 4, 40, 5))
    cfg = ra.RustpotterConfig.default().detector
    with pytest.raises(ra.RustpotterError, match="out of memory"):
        ra.StreamBatch(ctx, tm, cfg, 1 << 34)
    pcm = orc.synth_pcm(SEED, 1, 480 * 20)
    assert mfcc_close(ctx.mfcc(pcm[None, :], 5)[0], orc.mfcc_stream(pcm, 5))
    det, n_det = ctx.batch_detect(pcm[None, :], tm, cfg)
    assert n_det.shape == (1,)


def test_handles_release_their_device_memory(ra, ctx):
    """Creating and dropping detectors, template sets, models, live-stream batches and contexts 150 times leaves the
    device's free memory (hipMemGetInfo) and the process's resident set where they were, within the slack of the
    runtime's own pools."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    e = EXP["simulation"]["max"]
    cfg = _make_config(ra, e)
    s16 = simstream.simulation_stream_i16()
    tmpl = orc.synth_templates(SEED, 8, 100, 5)
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))

    def cycle():
        rp = ra.Rustpotter.new(cfg)
        rp.add_wakeword_from_file("w", os.path.join(G, e["rpw"]))
        for i in range(0, 480 * 40, 480):
            rp.process_samples(s16[i:i + 480])
        del rp
        c2 = ra.BatchContext(0)
        tm = ra.Templates(c2, tmpl)
        sb = ra.StreamBatch(c2, tm, cfg.detector, 256, max_chunks_per_call=2)
        sb.process(np.zeros((256, 960), np.float32))
        mdl = ra.Model(c2, [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]], [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]])
        c2.batch_detect(np.zeros((4, 480 * 50), np.float32), tm, cfg.detector)
        del sb, mdl, tm, c2

    import psutil
    for _ in range(5):
        cycle()
    before, rss0 = free_bytes(), psutil.Process().memory_info().rss
    for _ in range(150):
        cycle()
    after, rss1 = free_bytes(), psutil.Process().memory_info().rss
    assert before - after < 64 << 20, "device memory shrank by %.1f MB over 150 create / free cycles" % ((before - after) / 2**20)
    assert rss1 - rss0 < 96 << 20, "host memory grew by %.1f MB over 150 create / free cycles" % ((rss1 - rss0) / 2**20)


def test_distinct_handles_in_concurrent_threads(ra):
    """`WakewordDetector: Send`, one detector per thread (SURVEY 8b threading): four `Rustpotter` handles fed from four
    threads at once (ctypes drops the GIL during a call) each give the detections of a handle run alone."""
    import threading
    e = EXP["simulation"]["max"]
    s16 = simstream.simulation_stream_i16()
    n = (len(s16) // 480) * 480
    cfg = _make_config(ra, e)

    def run(shift, out):
        rp = ra.Rustpotter.new(cfg)
        rp.add_wakeword_from_file("w", os.path.join(G, e["rpw"]))
        x = np.roll(s16[:n], 480 * shift)
        got = []
        for i in range(0, n, 480):
            d = rp.process_samples(x[i:i + 480])
            if d is not None:
                got.append((i // 480, d.counter, float(d.score), float(d.avg_score)))
        out.append(got)

    alone = []
    for k in range(4):
        run(7 * k, alone)
    together = [[] for _ in range(4)]
    threads = [threading.Thread(target=run, args=(7 * k, together[k])) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(len(a) >= 1 for a in alone)
    assert [t[0] for t in together] == alone


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_stream_batch_direct_and_staged_input_interleaved(ra, ctx, dtype):
    """16 kHz mono chunks are read by the MFCC kernel where they lie (history chunk + new chunks from two buffers);
    rows whose pitch is not a multiple of 4 samples go through the staging copy.  Calls of both kinds, 1-3 chunks each,
    interleaved on the same streams: aggregates bitwise the offline ones, same detections."""
    S, N = 200, 480 * 60
    tm = ra.Templates(ctx, orc.synth_templates(SEED, 8, 100, 5))
    cfg = ra.RustpotterConfig.default().detector
    cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.0, 0.39, 1  # inside the score range of noise: detections come and go
    pcm = ctx.synth_pcm(SEED, 3, S, N)
    if dtype == np.int16:
        pcm = np.round(pcm * 32767.0).astype(np.int16)
    det, n_det, _, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=3)
    rng = np.random.default_rng(2)
    i, total, call = 0, np.zeros(S, np.int64), 0
    while i < N:
        nc = min(int(rng.integers(1, 4)), (N - i) // 480)
        piece = pcm[:, i:i + 480 * nc]
        if call % 2:  # odd row pitch: the staged path
            piece = np.concatenate([piece, np.zeros((S, 1), pcm.dtype)], axis=1)
        d, nd, a = sb.process(piece, want_agg=True, n_chunks=nc)
        f0 = 3 * (i // 480) - 3
        for k in range(a.shape[1]):
            wi = f0 + k - 99
            if 0 <= wi < agg.shape[1]:
                assert np.array_equal(a[:, k], agg[:, wi])
        for si in np.nonzero(nd)[0]:
            for j in range(nd[si]):
                assert _det_tuple(d[si][j]) == _det_tuple(det[si][total[si] + j])
        total += nd
        i += 480 * nc
        call += 1
    assert np.array_equal(total, n_det) and n_det.sum() > 0


def test_batch_detect_multi_equals_detector_with_two_wakewords(ra, ctx):
    """Two wakewords in one detector (run_wakeword_detectors, src/detector.rs:433-447): rp_batch_detect_multi against a
    Rustpotter handle that holds both .rpw files, on a stream in which both are spoken; per-wakeword threshold override."""
    rd = lambda f: simstream.i16_to_f32(rpw_py.read_wav_i16(os.path.join(G, f))[0])
    z = np.zeros(16000 * 2, np.float32)
    base = np.concatenate([z, rd("oye_casa_g_1.wav"), z, rd("alexa.wav"), z, rd("oye_casa_g_2.wav"), z, rd("alexa2.wav"), z, z])
    rng = np.random.default_rng(12)
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 13) + rng.standard_normal(n).astype(np.float32) * np.float32(0.001)])
    files = ["oye_casa_g.rpw", "alexa.rpw"]
    wws = [rpw_py.load_rpw(os.path.join(G, f)) for f in files]
    tms = [ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"]) for w in wws]
    cfg = ra.RustpotterConfig.default()
    cfg.fmt.sample_format = ra.SampleFormat.F32
    cfg.detector.threshold, cfg.detector.avg_threshold, cfg.detector.min_scores = 0.5, 0.2, 3
    det, dww, n_det = ctx.batch_detect_multi(pcm, tms, cfg.detector)
    for si in range(2):
        rp = ra.Rustpotter.new(cfg)
        for k, f in zip(("oye", "alexa"), files):
            rp.add_wakeword_from_file(k, os.path.join(G, f))
        got = []
        for i in range(0, n, 480):
            d = rp.process_samples(pcm[si, i:i + 480].copy())
            if d is not None:
                got.append((i // 480, d))
        names = [w["name"] for w in wws]
        assert n_det[si] == len(got) and len(got) >= 3
        seen = set()
        for j, (chunk, d) in enumerate(got):
            assert det[si][j]["frame"] // 3 + 1 == chunk and det[si][j]["counter"] == d.counter
            assert abs(det[si][j]["score"] - d.score) <= 1e-6 * d.score and abs(det[si][j]["avg_score"] - d.avg_score) <= 1e-6 * max(d.avg_score, 1e-3)
            assert names[dww[si][j]] == d.name
            seen.add(d.name)
        assert len(seen) == 2  # both wakewords fired
    # a wakeword's own threshold wins over the detector's (Option<f32> in the .rpw): silence "alexa" with 0.99
    det2, dww2, n2 = ctx.batch_detect_multi(pcm, tms, cfg.detector, thresholds=[None, 0.99])
    assert 0 < n2[0] < n_det[0] and all(dww2[0][j] == 0 for j in range(n2[0]))
    with pytest.raises(ra.RustpotterError, match="different mfcc size"):
        ctx.batch_detect_multi(pcm, [tms[0], ra.Templates(ctx, orc.synth_templates(SEED, 2, 40, 16))], cfg.detector)


@pytest.mark.parametrize("vad", ["easy", "medium", "hard"])
def test_batch_vad_gate_matches_oracle(ra, ctx, vad):
    """VadDetector in the batched scan: speech surrounded by low-level noise (the gate opens late
    and changes which frames are scored) against the oracle's chunked detector."""
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    rng = np.random.default_rng(9)
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    noisy = base + rng.standard_normal(len(base)).astype(np.float32) * np.float32(0.0005)
    n = (len(noisy) // 480) * 480
    pcm = noisy[:n][None, :]
    cfg = ra.RustpotterConfig.default().detector
    cfg.avg_threshold, cfg.threshold = 0.0, 0.5
    cfg.vad_mode = {"easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[vad]
    tm = ra.Templates(ctx, templates)
    det, n_det = ctx.batch_detect(pcm, tm, cfg)
    d = orc.Detector(avg_threshold=0.0, threshold=0.5, vad_mode=vad)
    d.add_ref(w)
    ref = []
    for i in range(0, n, 480):
        r = d.process_f32(pcm[0, i:i + 480])
        if r is not None:
            ref.append((i // 480, r))
    assert n_det[0] == len(ref)
    for j, (chunk, r) in enumerate(ref):
        assert det[0][j]["frame"] // 3 + 1 == chunk and det[0][j]["counter"] == r["counter"]
        assert abs(det[0][j]["score"] - r["score"]) <= 1e-5 * r["score"]


@pytest.mark.parametrize("case", ["band_pass", "gain_normalizer", "gain_and_band_pass", "ignore_alexa_filters"])
def test_batch_frontend_goldens(ra, ctx, case):
    """tests/detector.rs:100-162 through the BATCHED path: i16 stream -> rp_frontend_batch (decode + gain
    normaliser + band-pass on the device) -> rp_batch_detect; filtered audio bit-exact against the
    oracle's front-end, detections as the reference asserts."""
    e = EXP["simulation"][case]
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    templates = list(w["samples_features"].values())
    s16 = simstream.simulation_stream_i16(*e.get("gains", [1.0, 1.0]))
    n = (len(s16) // 480) * 480
    cfg = _make_config(ra, e)
    max_len = max(len(t) for t in templates)
    out, rms, gains = ctx.frontend(s16[:n], cfg.filters, w["rms_level"], max_len // 3)
    ref_out, ref_rms, ref_gains = orc.frontend_stream(simstream.i16_to_f32(s16[:n]), gain_normalizer=e.get("gain_normalizer", False),
                                                      rms_level_ref=w["rms_level"], window_size=max_len // 3,
                                                      band_pass=e.get("band_pass", False), low_cutoff=e.get("low_cutoff", 80.0),
                                                      high_cutoff=e.get("high_cutoff", 400.0))
    assert np.array_equal(rms[0], ref_rms) and np.array_equal(gains[0], ref_gains)
    assert np.array_equal(out[0], ref_out)  # integer decode, gain, clamp and biquad are all bit-exact
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    det, n_det = ctx.batch_detect(out, tm, cfg.detector)
    ref = _oracle_detections(e, s16)
    assert n_det[0] == len(ref) == len(e["detections"])
    for j, ((chunk, r), (_, gscore)) in enumerate(zip(ref, e["detections"])):
        assert det[0][j]["frame"] // 3 + 1 == chunk and det[0][j]["counter"] == r["counter"]
        assert abs(det[0][j]["score"] - np.float32(gscore)) <= 1e-5 * gscore
        # RustpotterDetection::gain = the gain of the chunk in which the best partial was scored
        assert gains[0][(det[0][j]["window"] + tm.max_len - 1) // 3 + 1] == r["gain"]


@pytest.mark.parametrize("tail", [37, 36, 0])
def test_frontend_many_streams_and_formats(ra, ctx, tail):
    """tail 37: row pitch not a multiple of 4 samples -> the one-lane-per-stream kernels; 36 / 0: the LDS-staged kernels
    (partial last workgroup, with and without a tail shorter than a chunk)."""
    rng = np.random.default_rng(8)
    S, N = 70, 480 * 25 + tail
    raw = (rng.standard_normal((S, N)) * 3000).astype(np.int16)
    raw[3] = 0  # silent stream: rms 0 -> gain stays 1, window untouched
    f = ra.FiltersConfig()
    f.gain_normalizer.enabled, f.gain_normalizer.min_gain, f.gain_normalizer.max_gain = True, 0.2, 3.0
    f.band_pass.enabled, f.band_pass.low_cutoff, f.band_pass.high_cutoff = True, 120.0, 900.0
    out, rms, gains = ctx.frontend(raw, f, 0.05, 7)
    for s in (0, 3, 69):
        ro, rr, rg = orc.frontend_stream(simstream.i16_to_f32(raw[s]), gain_normalizer=True, min_gain=0.2, max_gain=3.0,
                                         rms_level_ref=0.05, window_size=7, band_pass=True, low_cutoff=120.0, high_cutoff=900.0)
        assert np.array_equal(rms[s], rr) and np.array_equal(gains[s], rg) and np.array_equal(out[s], ro)
    f.gain_normalizer.gain_ref = 0.02  # fixed reference level
    f.band_pass.enabled = False
    out, rms, gains = ctx.frontend(raw.astype(np.float32) / np.float32(32767.0), f, float("nan"), 4)
    ro, rr, rg = orc.frontend_stream(simstream.i16_to_f32(raw[5]), gain_normalizer=True, gain_ref=0.02, min_gain=0.2, max_gain=3.0,
                                     window_size=4)
    assert np.array_equal(gains[5], rg) and np.array_equal(out[5], ro)


@pytest.mark.parametrize("dtype,tmax", [(np.int16, 32767.0), (np.int8, 127.0)])
def test_sample_decode_is_exact_for_every_value(ra, ctx, dtype, tmax):
    """`v as f32 / T::MAX as f32` (src/audio/encoder.rs): the kernels decode i8 / i16 with a 3-instruction exact division
    (rp_device.h div_small_int); every representable sample value must equal the IEEE quotient.  Run through the tiled
    front-end kernel (row pitch a multiple of 4), the one-lane-per-stream kernel (odd pitch) and the MFCC decode."""
    info = np.iinfo(dtype)
    vals = np.arange(info.min, info.max + 1, dtype=np.int64).astype(dtype)
    want = vals.astype(np.float32) / np.float32(tmax)
    for pad in (0, 1):
        n = (len(vals) + 479) // 480 * 480 + 480 + pad
        x = np.zeros((2, n), dtype)
        x[0, :len(vals)] = vals
        x[1, n - len(vals):] = vals  # also through the tail that is shorter than a chunk
        out, _, _ = ctx.frontend(x, ra.FiltersConfig(), float("nan"), 1)
        assert np.array_equal(out[0, :len(vals)], want) and np.array_equal(out[1, n - len(vals):], want)
    if dtype == np.int16:  # the MFCC kernel decodes with the same helper: features of the decoded f32 stream are identical
        assert np.array_equal(ctx.mfcc(np.tile(vals, 2)[None, :], 16), ctx.mfcc(np.tile(want, 2)[None, :], 16))


@pytest.mark.parametrize("rpw,wavs", [("oye_casa_g.rpw", ["oye_casa_g_%d.wav" % i for i in range(1, 6)]),
                                      ("alexa.rpw", ["alexa.wav", "alexa2.wav", "alexa3.wav"])])
def test_build_wakeword_ref_matches_reference_file(ra, ctx, rpw, wavs, tmp_path):
    """tests/wakeword.rs:27-54 on the device: WakewordRef::new_from_sample_files(.., 5) + save; the result is
    compared with the .rpw the reference itself wrote from the same wavs (G1 templates, G3 average, rms level)."""
    gold = rpw_py.load_rpw(os.path.join(G, rpw))
    samples = {w: open(os.path.join(G, w), "rb").read() for w in wavs}
    data = ctx.build_wakeword_ref(gold["name"], samples, 5)
    path = tmp_path / "built.rpw"
    path.write_bytes(data)
    built = rpw_py.load_rpw(str(path))  # the test-side CBOR reader parses what the product wrote
    assert built["kind"] == "ref" and built["name"] == gold["name"] and built["mfcc_size"] == 5
    assert built["threshold"] is None and built["avg_threshold"] is None
    assert np.float32(built["rms_level"]) == np.float32(gold["rms_level"])
    assert set(built["samples_features"]) == set(gold["samples_features"])
    for k, ref in gold["samples_features"].items():
        assert mfcc_close(built["samples_features"][k], ref)
    # the average follows the DTW path of the reference (same argmin decisions) -> same alignment, tiny numeric drift
    assert built["avg_features"].shape == gold["avg_features"].shape
    assert np.abs(built["avg_features"] - gold["avg_features"]).max() <= 2e-5
    # the product's own reader accepts it and the detector behaves as with the reference's file
    e = EXP["simulation"]["max"] if rpw.startswith("oye") else EXP["simulation"]["ignore_alexa"]
    rp = ra.Rustpotter.new(_make_config(ra, e))
    rp.add_wakeword_from_buffer("w", data)
    raw = simstream.simulation_stream_i16().astype("<i2").tobytes()
    dets = [d for d in (rp.process_bytes(raw[i:i + 960]) for i in range(0, len(raw) - 959, 960)) if d is not None]
    assert len(dets) == len(e["detections"])
    for d, (gavg, gscore) in zip(dets, e["detections"]):
        assert abs(d.score - np.float32(gscore)) <= 2e-5 * gscore and abs(d.avg_score - np.float32(gavg)) <= 1e-4 * gavg


# ---------------------------------------------------------------- resampler
def _audio_scale_close(got, ref, tol=4e-6):
    """resampled audio: |d| <= tol * max|ref|.  Both sides round to f32 along ~3 000-term sums (the oracle its
    spectra, the kernel its accumulator); measured max 2.1e-6, rms 4e-7 of the peak (tools/probe_resample_err.py)."""
    scale = max(float(np.abs(ref).max()), 1e-3)
    return float(np.abs(got - ref).max()) <= tol * scale


def test_resampler_frame_lengths(ra):
    for fs, lens in {48000: (1440, 480), 44100: (1323, 480), 32000: (960, 480), 8000: (240, 480), 16000: (480, 480),
                     22050: (882, 640), 11025: (441, 640), 96000: (2880, 480)}.items():
        assert ra.resampler_frame_lengths(fs) == lens
        if fs != 16000:
            r = orc.Resampler(fs)
            assert (r.in_len, r.out_len) == lens
    with pytest.raises(ra.RustpotterError, match="Unsupported sample rate"):
        ra.resampler_frame_lengths(0)


@pytest.mark.parametrize("fs", [48000, 44100, 8000, 22050, 32000])
def test_resample_batch_matches_oracle(ra, ctx, fs):
    """rp_resample_batch (one f32 MFMA product per output frame) against the oracle's frame-by-frame
    FFT / overlap-add restatement of rubato, several rates (480- and 640-sample output frames, up- and
    down-sampling, a matrix depth that is not a multiple of 16), ragged tail dropped."""
    fi, fo = ra.resampler_frame_lengths(fs)
    rng = np.random.default_rng(fs)
    n = fi * 7 + 123
    t = np.arange(n)
    pcm = np.stack([rng.uniform(-0.5, 0.5, n), 0.3 * np.sin(2 * np.pi * 440.0 * t / fs) + 0.01 * rng.standard_normal(n),
                    np.where((t // 500) % 2 == 0, 0.25, -0.25)]).astype(np.float32)
    got = ctx.resample(pcm, fs)
    assert got.shape == (3, 7 * fo)
    for s in range(3):
        ref = orc.resample_stream(pcm[s], fs)
        assert ref.shape == got[s].shape
        assert _audio_scale_close(got[s], ref)


def test_resample_batch_formats_and_channels(ra, ctx):
    """i16 stereo in, first channel used (reencode_to_mono, src/audio/encoder.rs:42-48), decoded like Sample::into_f32."""
    rng = np.random.default_rng(4)
    n = 1440 * 5
    l = rng.integers(-20000, 20000, n).astype(np.int16)
    r = rng.integers(-20000, 20000, n).astype(np.int16)
    inter = np.stack([l, r], axis=1).reshape(1, -1)
    got = ctx.resample(inter, 48000, channels=2)
    ref = orc.resample_stream(l.astype(np.float32) / np.float32(32767.0), 48000)
    assert got.shape == (1, len(ref)) and _audio_scale_close(got[0], ref)
    with pytest.raises(ra.RustpotterError, match="Unsupported sample rate"):
        ctx.resample(np.zeros((1, 1000), np.float32), 44101)  # coprime with 16 kHz: a 16 000-sample output frame


def test_resample_many_streams(ra, ctx):
    S, fs = 700, 48000
    pcm = ctx.synth_pcm(SEED, 0, S, 1440 * 9)
    got = ctx.resample(pcm, fs)
    for s in (0, 1, 63, 64, 333, 699):
        assert _audio_scale_close(got[s], orc.resample_stream(pcm[s], fs))


@pytest.mark.parametrize("name", sorted(EXP["filter_examples"].keys()))
def test_resample_and_frontend_match_reference_filter_examples(ra, ctx, name):
    """The reference's own filter tests (src/audio/band_pass_filter.rs:69-185, gain_normalizer_filter.rs:81-131) write
    the audio they produce from real_sample.wav: 48 kHz -> resampler -> gain normaliser / band-pass.  The batched device
    path (rp_resample_batch + rp_frontend_batch) against those files, 170 880 samples each."""
    e = EXP["filter_examples"][name]
    x, sr, _ = rpw_py.read_wav(os.path.join(G, "real_sample.wav"))
    ref, _, _ = rpw_py.read_wav(os.path.join(G, name))
    y = ctx.resample(x, sr)
    assert y.shape == (1, len(ref))
    fc = ra.FiltersConfig()
    fc.gain_normalizer.enabled = e.get("gain_normalizer", False)
    fc.gain_normalizer.gain_ref = e.get("gain_ref")
    fc.gain_normalizer.min_gain, fc.gain_normalizer.max_gain = e.get("min_gain", 0.1), e.get("max_gain", 1.0)
    fc.band_pass.enabled = e.get("band_pass", False)
    fc.band_pass.low_cutoff, fc.band_pass.high_cutoff = e.get("low_cutoff", 80.0), e.get("high_cutoff", 400.0)
    out, rms, gains = ctx.frontend(y, fc, float("nan"), 1)
    peak = float(np.abs(ref).max())
    d = np.abs(out[0] - ref)
    assert d.max() <= (4e-6 if e.get("band_pass") else 6e-7) * peak
    assert np.sqrt((d * d).mean()) <= 3e-7 * peak


def test_build_wakeword_ref_from_48k_recordings(ra, ctx, tmp_path):
    """tests/wakeword.rs:57-71 on the device: six 48 kHz f32 wavs -> resample -> MFCC -> normalise -> average;
    against the .rpw the reference wrote from the same files (4 680 MFCC values, average, rms level)."""
    gold = rpw_py.load_rpw(os.path.join(G, "oye_casa_real.rpw"))
    wavs = ["oye_casa_real_%d.wav" % i for i in range(1, 7)]
    data = ctx.build_wakeword_ref(gold["name"], {w: open(os.path.join(G, w), "rb").read() for w in wavs}, 5)
    path = tmp_path / "built.rpw"
    path.write_bytes(data)
    built = rpw_py.load_rpw(str(path))
    assert abs(built["rms_level"] - gold["rms_level"]) <= 1e-6 * gold["rms_level"]  # an RMS of resampled audio: 1 ulp off
    assert set(built["samples_features"]) == set(gold["samples_features"])
    for k, ref in gold["samples_features"].items():
        got = built["samples_features"][k]
        assert got.shape == ref.shape
        assert np.all(np.abs(got - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0) + 3e-5)
    assert built["avg_features"].shape == gold["avg_features"].shape
    assert np.abs(built["avg_features"] - gold["avg_features"]).max() <= 1e-4


@pytest.mark.parametrize("case", sorted(EXP["audio_file"].keys()))
def test_rustpotter_api_goldens_48k_recording(ra, case):
    """tests/detector.rs:163-213 through the drop-in API: config.fmt = the wav's spec (48 kHz f32), samples fed
    in get_samples_per_frame() = 1 440 chunks, 5 s of zeros appended; avg_score / score / counter of the three
    detections as asserted by the reference."""
    e = EXP["audio_file"][case]
    pcm, sr, ch = rpw_py.read_wav(os.path.join(G, e["wav"]))
    pcm = np.concatenate([pcm, np.zeros(sr * 5, np.float32)])
    cfg = _make_config(ra, e)
    cfg.fmt.sample_rate, cfg.fmt.sample_format, cfg.fmt.channels = sr, ra.SampleFormat.F32, ch
    cfg.detector.min_scores = e["min_scores"]
    if "min_gain" in e:
        cfg.filters.gain_normalizer.min_gain = e["min_gain"]
    rp = ra.Rustpotter.new(cfg)
    rp.add_wakeword_from_file("wakeword", os.path.join(G, e["rpw"]))
    n = rp.get_samples_per_frame()
    assert n == 1440 and rp.get_bytes_per_frame() == 1440 * 4
    dets = [d for d in (rp.process_samples(pcm[i:i + n].copy()) for i in range(0, len(pcm) - n + 1, n)) if d is not None]
    assert len(dets) == len(e["detections"])
    for d, (avg, score, counter) in zip(dets, e["detections"]):
        assert d.counter == counter
        assert abs(d.avg_score - np.float32(avg)) <= 1e-5 * avg and abs(d.score - np.float32(score)) <= 1e-5 * score


def test_rustpotter_api_22k_input_four_shifts_per_call(ra):
    """22.05 kHz input: 882 samples in, 640 out = four 10 ms shifts per call (one frame from the first call after
    a reset, four from every later one) -- the chunked detector against the oracle's, detection for detection."""
    e = EXP["simulation"]["max"]
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    # a 22.05 kHz rendition of the simulation stream (linear interpolation is good enough: both sides get the same input)
    t = np.arange(int(len(base) * 22050 / 16000)) * (16000.0 / 22050.0)
    x = np.interp(t, np.arange(len(base)), base).astype(np.float32)
    # no digital silence: frames made of nothing but the resampler's ringing (1e-9 of full scale) have log-mel
    # values decided by rounding and are not comparable between two implementations
    x += np.random.default_rng(8).standard_normal(len(x)).astype(np.float32) * np.float32(2e-4)
    cfg = _make_config(ra, e)
    cfg.fmt.sample_rate, cfg.fmt.sample_format = 22050, ra.SampleFormat.F32
    rp = ra.Rustpotter.new(cfg)
    rp.add_wakeword_from_file("w", os.path.join(G, e["rpw"]))
    n = rp.get_samples_per_frame()
    assert n == 882
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    d = orc.Detector(avg_threshold=e["avg_threshold"], threshold=e["threshold"], score_mode=e["score_mode"])
    d.add_ref(w)
    rs = orc.Resampler(22050)
    got, ref = [], []
    for i in range(0, len(x) - n + 1, n):
        a = rp.process_samples(x[i:i + n].copy())
        b = d.process_resampled(rs, x[i:i + n])
        if a is not None:
            got.append((i // n, a.counter, a.score, a.avg_score))
        if b is not None:
            ref.append((i // n, b["counter"], b["score"], b["avg_score"]))
    assert len(ref) >= 2 and len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g[0] == r[0] and g[1] == r[1]
        assert abs(g[2] - r[2]) <= 1e-5 * r[2] and abs(g[3] - r[3]) <= 1e-5 * r[3]


# ---------------------------------------------------------------- NN training (second half)
def _train_sets():
    e = EXP["train"]
    rd = lambda f: open(os.path.join(G, f), "rb").read()
    return e, {k: rd(v) for k, v in e["train"].items()}, {k: rd(v) for k, v in e["test"].items()}


def test_train_wakeword_model_like_the_reference_test(ra, ctx, tmp_path):
    """tests/wakeword.rs:86-98: ModelType::Medium, lr 0.027, 10 epochs, mfcc 16 on tests/resources/{train,test};
    the reference asserts labels / weights / train_size / mfcc_size.  The trained file loads into the detector."""
    e, train, test = _train_sets()
    data, loss, acc = ctx.train_wakeword_model(train, test, e["m_type"], e["learning_rate"], e["epochs"], e["test_epochs"],
                                               e["mfcc_size"], seed=7)
    p = tmp_path / "trained.rpw"
    p.write_bytes(data)
    m = rpw_py.load_rpw(str(p))
    assert m["kind"] == "model" and m["m_type"] == "Medium"
    assert len(m["labels"]) == e["labels"] and set(m["labels"]) == {"none", "oye casa"}
    assert len(m["weights"]) == e["weights"] and m["train_size"] == e["train_size"] and m["mfcc_size"] == e["mfcc_size"]
    fr = e["train_size"]
    assert m["weights"]["ln1.weight"].shape == (fr // 3, fr * 16) and m["weights"]["ln2.weight"].shape == (fr // 6, fr // 3)
    assert m["weights"]["ln3.weight"].shape == (2, fr // 6)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0
    # rms level: running (a+b)/2 over the labelled samples (wakeword_model_train.rs:311-317)
    lv = np.float32("nan")
    for name, f in e["train"].items():
        if "[" not in name:
            continue
        pcm, sr, _ = rpw_py.read_wav(os.path.join(G, f))
        y = orc.resample_stream(pcm, sr)
        r = sorted(orc.lib().orc_rms_level(orc._f(np.ascontiguousarray(y[i:i + 480])), 480) for i in range(0, len(y) - 479, 480))
        cur = np.float32(r[len(r) // 2])
        lv = cur if np.isnan(lv) else np.float32((lv + cur) / np.float32(2.0))
    assert abs(m["rms_level"] - lv) <= 2e-6 * lv
    c = ra.RustpotterConfig.default()
    c.fmt.sample_rate, c.fmt.sample_format = 48000, ra.SampleFormat.F32
    rp = ra.Rustpotter.new(c)
    rp.add_wakeword_from_buffer("trained", data)
    assert rp.process_samples(np.zeros(1440, np.float32)) is None


def test_train_epochs_match_oracle(ra, ctx, tmp_path):
    """Same start, same data, same number of SGD epochs -> same weights as the oracle's restatement of the
    training loop.  The start is the product's own seeded initialisation (0 epochs), read back from its file."""
    e, train, test = _train_sets()
    init, _, _ = ctx.train_wakeword_model(train, test, "small", 0.027, 0, 1, 16, seed=3)
    p0 = tmp_path / "init.rpw"
    p0.write_bytes(init)
    m0 = rpw_py.load_rpw(str(p0))
    labels = m0["labels"]
    # candle_nn::linear's initialisation: weights ~ N(0, 2/fan_in), biases within +-1/sqrt(fan_in)
    w1 = m0["weights"]["ln1.weight"]
    assert abs(w1.std() * np.sqrt(w1.shape[1] / 2.0) - 1.0) < 0.02 and abs(w1.mean()) < 1e-3
    assert np.abs(m0["weights"]["ln1.bias"]).max() <= 1.0 / np.sqrt(w1.shape[1])
    # features the way compute_mfccs makes them, from the oracle
    L = m0["train_size"] * 16
    xs, ys = [], []
    for name, f in e["train"].items():
        pcm, sr, _ = rpw_py.read_wav(os.path.join(G, f))
        if pcm.dtype != np.float32:
            pcm = simstream.i16_to_f32(pcm)
        feat = orc.wav_features(pcm, sr, 16).reshape(-1)
        row = np.zeros(L, np.float32)
        row[:min(L, len(feat))] = feat[:L]
        xs.append(row)
        ys.append(labels.index(name[name.index("[") + 1:name.index("]")].lower() if "[" in name else "none"))
    ws = [m0["weights"]["ln%d.weight" % i] for i in (1, 2, 3)]
    bs = [m0["weights"]["ln%d.bias" % i] for i in (1, 2, 3)]
    epochs = 6
    data, loss, acc = ctx.train_wakeword_model(train, test, "tiny", 0.5, epochs, 1, 5, seed=99, prev_model=init)  # options ignored
    p1 = tmp_path / "trained.rpw"
    p1.write_bytes(data)
    m1 = rpw_py.load_rpw(str(p1))
    assert m1["m_type"] == "Small" and m1["labels"] == labels and m1["train_size"] == m0["train_size"]
    # NOTE: with prev_model the reference keeps the model's type but takes the learning rate from the options
    rw, rb, rloss = orc.mlp_train(np.stack(xs), ys, ws, bs, 0.5, epochs)
    assert abs(loss - rloss) <= 1e-4 * max(abs(rloss), 1e-3)
    for i in (1, 2, 3):
        for kind, ref in (("weight", rw[i - 1]), ("bias", rb[i - 1])):
            got = m1["weights"]["ln%d.%s" % (i, kind)]
            assert got.shape == ref.shape
            moved = np.abs(ref - (ws if kind == "weight" else bs)[i - 1]).max()
            assert np.abs(got - ref).max() <= 1e-4 * max(float(np.abs(ref).max()), float(moved))
    assert any(np.abs(rw[i] - ws[i]).max() > 1e-4 for i in range(3))  # the epochs did move the weights


def test_train_errors(ra, ctx):
    e, train, test = _train_sets()
    with pytest.raises(ra.RustpotterError, match="No training data provided"):
        ctx.train_wakeword_model({}, test)
    with pytest.raises(ra.RustpotterError, match="No test data provided"):
        ctx.train_wakeword_model(train, {})
    only_noise = {k: v for k, v in train.items() if "[" not in k}
    with pytest.raises(ra.RustpotterError, match="at least two labels"):
        ctx.train_wakeword_model(only_noise, {k: v for k, v in test.items() if "[" not in k})
    with pytest.raises(ra.RustpotterError, match="Forbidden label 'alexa'"):
        ctx.train_wakeword_model(train, {"x[Alexa].wav": open(os.path.join(G, "alexa.wav"), "rb").read()})


def test_build_wakeword_ref_errors_and_options(ra, ctx, tmp_path):
    one = {"a.wav": open(os.path.join(G, "alexa.wav"), "rb").read()}
    built = ctx.build_wakeword_ref("solo", one, 7, threshold=0.4, avg_threshold=0.1, from_files=False)
    p = tmp_path / "solo.rpw"
    p.write_bytes(built)
    w = rpw_py.load_rpw(str(p))
    assert w["avg_features"] is None and w["mfcc_size"] == 7 and w["samples_features"]["a.wav"].shape[1] == 7
    assert abs(w["threshold"] - 0.4) < 1e-7 and abs(w["avg_threshold"] - 0.1) < 1e-7
    with pytest.raises(ra.RustpotterError, match="Can not create an empty wakeword"):
        ctx.build_wakeword_ref("none", {}, 5)
    with pytest.raises(ra.RustpotterError, match="RIFF"):
        ctx.build_wakeword_ref("bad", {"x.wav": b"not a wav file at all"}, 5)


def test_mlp_forward_model_file(ra, ctx):
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    x = np.random.default_rng(0).standard_normal((165, 3120)).astype(np.float32)
    ws = [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]]
    bs = [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]]
    model = ra.Model(ctx, ws, bs)
    got = ctx.mlp_forward(x, model)                       # f32-input MFMA: exact f32
    ref = orc.mlp_forward(x, ws, bs)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5)
    got16 = ctx.mlp_forward(x, model, precision="bf16")   # bf16 MFMA vs the bf16-rounding oracle: rel 1e-3
    ref16 = orc.mlp_forward(x, ws, bs, bf16_layer1=True)
    assert np.allclose(got16, ref16, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dims", [(3120, 32, 16, 2), (1040, 65, 32, 3), (2080, 130, 32, 2), (64, 13, 2), (36, 7, 5, 2), (30, 6, 2)])
def test_mlp_forward_shapes(ra, ctx, dims):
    """Small/Medium/Large-shaped stacks (wakeword_nn.rs:325-389) and odd shapes (generic kernel when in % 4 != 0)."""
    rng = np.random.default_rng(sum(dims))
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
    x = rng.standard_normal((77, dims[0])).astype(np.float32)
    model = ra.Model(ctx, ws, bs)
    got = ctx.mlp_forward(x, model)
    ref = orc.mlp_forward(x, ws, bs)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5), np.abs(got - ref).max()
    if dims[0] % 4 == 0:
        got16 = ctx.mlp_forward(x, model, precision="bf16")
        ref16 = orc.mlp_forward(x, ws, bs, bf16_layer1=True)
        assert np.allclose(got16, ref16, rtol=1e-3, atol=1e-3), np.abs(got16 - ref16).max()


def test_rustpotter_api_model_wakeword(ra):
    """NN wakeword through the API on 16 kHz synthetic audio vs the oracle detector (parity of
    the forward pass is pinned by our own oracle only: SURVEY §8c G5)."""
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    c = ra.RustpotterConfig.default()
    c.detector.avg_threshold = 0.0
    c.detector.threshold = 0.0001
    rp = ra.Rustpotter.new(c)
    rp.add_wakeword_from_file("w", os.path.join(G, "ok_casa-tiny.rpw"))
    d = orc.Detector(avg_threshold=0.0, threshold=0.0001)
    d.add_model(m)
    pcm = orc.synth_pcm(SEED, 11, 480 * 150) * np.float32(0.2)
    got, ref = [], []
    for i in range(0, len(pcm), 480):
        a = rp.process_samples(pcm[i:i + 480].copy())
        b = d.process_f32(pcm[i:i + 480])
        if a is not None:
            got.append((i // 480, a))
        if b is not None:
            ref.append((i // 480, b))
    assert [g[0] for g in got] == [r[0] for r in ref]
    for (_, a), (_, b) in zip(got, ref):
        assert a.name == b["name"] and a.counter == b["counter"]
        assert abs(a.score - b["score"]) <= 1e-4


@pytest.mark.parametrize("avg_threshold", [0.0, 0.3])
def test_batch_detect_model_equals_per_stream_api(ra, ctx, avg_threshold):
    """A wakeword model in the batched detector (rp_batch_detect_model: windows -> MLP -> label / score logic -> state
    machine) against Rustpotter handles holding the same model file, stream by stream; f32 MFMA like the handle."""
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    ws = [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]]
    bs = [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]]
    model = ra.Model(ctx, ws, bs)
    none_index = m["labels"].index("none")
    c = ra.RustpotterConfig.default()
    c.detector.avg_threshold, c.detector.threshold, c.detector.min_scores = avg_threshold, 0.6, 3
    # the model's own recording (48 kHz -> 16 kHz through the oracle) over low noise, plus synthetic streams
    x48, sr, _ = rpw_py.read_wav(os.path.join(G, "ok_casa.wav"))
    rng = np.random.default_rng(21)
    speech = orc.resample_stream(x48, sr)
    n = 480 * 420
    streams = []
    for shift in (16000, 40000):
        s = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
        s[shift:shift + len(speech)] += speech
        streams.append(s)
    streams.append(orc.synth_pcm(SEED, 11, n) * np.float32(0.2))
    pcm = np.stack(streams)
    det, dlab, n_det = ctx.batch_detect_model(pcm, model, m["mfcc_size"], none_index, c.detector)
    total = 0
    for si in range(len(streams)):
        rp = ra.Rustpotter.new(c)
        rp.add_wakeword_from_file("w", os.path.join(G, "ok_casa-tiny.rpw"))
        got = []
        for i in range(0, n, 480):
            d = rp.process_samples(pcm[si, i:i + 480].copy())
            if d is not None:
                got.append((i // 480, d))
        assert n_det[si] == len(got)
        total += len(got)
        for j, (chunk, d) in enumerate(got):
            assert det[si][j]["frame"] // 3 + 1 == chunk and det[si][j]["counter"] == d.counter
            assert m["labels"][dlab[si][j]] == d.name
            assert abs(det[si][j]["score"] - d.score) <= 1e-5 and abs(det[si][j]["avg_score"] - d.avg_score) <= 1e-5
    assert total >= 2


def test_batch_detect_model_generic_shape_from_trained_file(ra, ctx, tmp_path):
    """A model trained on the device with mfcc_size 5 (feature rows not 16-byte sized: materialised rows + the generic
    layer kernel) -> the same file drives a Rustpotter handle and rp_batch_detect_model; detections must agree."""
    e, train, test = _train_sets()
    data, loss, acc = ctx.train_wakeword_model(train, test, "tiny", 0.05, 30, 10, 5, seed=5)
    p = tmp_path / "m5.rpw"
    p.write_bytes(data)
    m = rpw_py.load_rpw(str(p))
    assert m["mfcc_size"] == 5
    model = ra.Model(ctx, [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]], [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]])
    none_index = m["labels"].index("none")
    c = ra.RustpotterConfig.default()
    c.detector.avg_threshold, c.detector.threshold, c.detector.min_scores = 0.0, 0.55, 2
    x48, sr, _ = rpw_py.read_wav(os.path.join(G, "oye_casa_real_1.wav"))
    speech = orc.resample_stream(x48, sr)
    rng = np.random.default_rng(3)
    n = 480 * 300
    s0 = rng.standard_normal(n).astype(np.float32) * np.float32(0.003)
    s0[30000:30000 + len(speech)] += speech
    pcm = np.stack([s0, orc.synth_pcm(SEED, 2, n) * np.float32(0.1)])
    det, dlab, n_det = ctx.batch_detect_model(pcm, model, 5, none_index, c.detector)
    for si in range(2):
        rp = ra.Rustpotter.new(c)
        rp.add_wakeword_from_buffer("w", data)
        got = [(i // 480, d) for i in range(0, n, 480) for d in [rp.process_samples(pcm[si, i:i + 480].copy())] if d is not None]
        assert n_det[si] == len(got)
        for j, (chunk, d) in enumerate(got):
            assert det[si][j]["frame"] // 3 + 1 == chunk and det[si][j]["counter"] == d.counter and m["labels"][dlab[si][j]] == d.name
            assert abs(det[si][j]["score"] - d.score) <= 1e-5


# --------------------------------------------------------------------------- full BASELINE sizes
def test_randomised_parity_sweep(ra, ctx):
    """24 random (reference, config, streams) cases, offline and live, against the oracle's chunked detector
    (tests/sweep_parity.py; longer runs: `python tests/sweep_parity.py --cases 300`)."""
    import sweep_parity
    n, total, ties = sweep_parity.run_sweep(ra, ctx, 24, seed=7)
    assert n == 24 and total >= 10 and ties == 0
    # live-stream batches with single streams reset at random call boundaries
    n, total = sweep_parity.run_live_reset_sweep(ra, ctx, 16, seed=7)
    assert n == 16
    # live-stream batches behind the resampler (8-96 kHz, stereo): bitwise the offline pass over the resampled recording
    n, total, ties = sweep_parity.run_live_rate_sweep(ra, ctx, 10, seed=7)
    assert n == 10 and ties == 0
    # 1-3 wakewords with their own thresholds in rp_batch_detect_multi
    n, total = sweep_parity.run_multi_sweep(ra, ctx, 16, seed=7)
    assert n == 16 and total >= 3


def test_long_streams_offline_and_live(ra, ctx):
    """Two ~2.5 minute streams (5 000 chunks: hundreds of window-buffer compactions in the live batch, thousands of
    detections' worth of state-machine resets): offline batch and live batch against the oracle's chunked detector."""
    import sweep_parity
    case = sweep_parity.make_case(np.random.default_rng([7, 123]))
    pcm = case["pcm"][:2, :(case["pcm"].shape[1] // 480) * 480]
    reps = -(-5000 * 480 // pcm.shape[1])
    case["pcm"] = np.ascontiguousarray(np.tile(pcm, (1, reps)))
    case["cfg"].update(threshold=0.35, min_scores=2, vad_mode=None)
    sweep_parity.kMaxDet, keep = 4096, sweep_parity.kMaxDet
    try:
        ref = sweep_parity.oracle_detections(case)
        offline, live, _ = sweep_parity.device_detections(ra, ctx, case)
    finally:
        sweep_parity.kMaxDet = keep
    assert sum(len(r) for r in ref) >= 20
    for o, l, r in zip(offline, live, ref):
        assert len(o) == len(l) == len(r)
        assert [x[:2] for x in o] == [x[:2] for x in r] and o == l
        assert all(abs(a[2] - b[2]) <= 1e-5 * abs(b[2]) for a, b in zip(o, r))
    # the same first stream through one `Rustpotter` handle, chunk by chunk (window history compactions of the handle)
    c = case["cfg"]
    rc = ra.RustpotterConfig.default()
    rc.fmt.sample_format = ra.SampleFormat.I16 if case["pcm"].dtype == np.int16 else ra.SampleFormat.F32
    d = rc.detector
    d.avg_threshold, d.threshold, d.min_scores, d.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
    d.score_ref, d.band_size = c["score_ref"], c["band_size"]
    d.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
    rp = ra.Rustpotter.new(rc)
    rp.add_wakeword_from_buffer("w", rpw_py.dump_rpw_ref("w", {"t%d" % i: t for i, t in enumerate(case["templates"])}, case["avg"]))
    x = case["pcm"][0]
    got = []
    for k in range(len(x) // 480):
        r = rp.process_samples(x[480 * k:480 * (k + 1)])
        if r is not None:
            got.append((k, r.counter, float(r.score)))
    assert [g[:2] for g in got] == [r[:2] for r in ref[0]]
    assert all(abs(g[2] - r[2]) <= 1e-5 * abs(r[2]) for g, r in zip(got, ref[0]))


def test_randomised_mfcc_sweep(ra, ctx):
    """120 random (signal kind, level, mfcc size, length, sample type) MFCC cases against the oracle."""
    import sweep_parity
    n, worst = sweep_parity.run_mfcc_sweep(ra, ctx, 120, seed=7)
    assert n == 120 and len(worst) == 7


def test_randomised_frontend_sweep(ra, ctx):
    """40 random decode + gain normaliser + band-pass cases (stream counts around a workgroup, aligned / odd row
    lengths, four sample types, filter parameters) bit for bit against the oracle."""
    import sweep_parity
    n, checked = sweep_parity.run_frontend_sweep(ra, ctx, 40, seed=7)
    assert n == 40 and checked >= 60


def test_randomised_resample_sweep(ra, ctx):
    """40 random resampler cases (16 input rates from 4 to 192 kHz, 1-3 channels, four sample types) against the oracle."""
    import sweep_parity
    n, worst = sweep_parity.run_resample_sweep(ra, ctx, 40, seed=7)
    assert n == 40 and 0.0 < worst <= 8e-6


def test_randomised_builder_sweep(ra, ctx):
    """20 wakeword references built from random wav files (8 / 16 / 32-bit PCM, float, stereo, 48 kHz) against the oracle's
    wav extractor and averager."""
    import sweep_parity
    n, checked = sweep_parity.run_builder_sweep(ra, ctx, 20, seed=7)
    assert n == 20 and checked >= 20


def test_randomised_train_sweep(ra, ctx):
    """6 wakeword models trained from random labelled wav sets (four model types, mfcc sizes, learning rates, epochs):
    same weights and loss as the oracle's training loop from the same start."""
    import sweep_parity
    assert sweep_parity.run_train_sweep(ra, ctx, 6, seed=7) >= 4


def test_randomised_api_sweep(ra, ctx):
    """12 random single-stream cases through `Rustpotter` chunk by chunk (several wakewords, filters, VAD, resets,
    stereo, 48 kHz) against the oracle's detector: same chunks fire, same name / counter / partial state, scores 1e-5."""
    import sweep_parity
    n, total = sweep_parity.run_api_sweep(ra, 12, seed=7)
    assert n == 12 and total >= 3
    # wakeword models of the four types with random layer sizes / weights / labels
    n, total = sweep_parity.run_model_sweep(ra, 16, seed=7, ctx=ctx)  # ctx: rp_batch_detect_model on the same streams too
    assert n == 16 and total >= 3


def _full_size_run(ra, S, T, seed_templates=SEED):
    """Whole path on device-resident synthetic input exactly as bench.py sets it up."""
    import torch
    N, L, K = 64000, 100, 5
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(seed_templates, T, L, K)
    tmpl = ra.Templates(ctx, templates)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
    n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    return ctx, tmpl, cfg, templates, pcm, scores, agg, det, n_det


@pytest.mark.parametrize("S,T", [(65536, 8), (8192, 64)])
def test_full_size_properties(ra, S, T):
    """BASELINE configs C3 (65 536 streams x 8 templates) and one GPU's share of C4 (8 192 x 64) at FULL size,
    checked through size-independent properties: scores are probabilities, the aggregate is the row maximum
    (ScoreMode::Max), the run is bit-reproducible, a stream's result does not depend on the batch it is in,
    and sampled streams agree with the oracle to 1e-5."""
    import torch
    ctx, tmpl, cfg, templates, pcm, scores, agg, det, n_det = _full_size_run(ra, S, T)
    N, L = 64000, 100
    assert scores.shape == (S, 297, T)
    assert bool(torch.isfinite(scores).all()) and float(scores.min()) > 0.0 and float(scores.max()) < 1.0
    assert torch.equal(agg, scores.max(dim=2).values)
    # a stream can only fire where its aggregate crosses the threshold (0.5); with 8 templates none does: the
    # work is data independent.  (With 64 templates the xor-seeded generator makes stream 0 a sample-permuted
    # copy of some templates and it legitimately fires.)
    quiet = agg.max(dim=1).values <= 0.5
    assert int(n_det[quiet].sum()) == 0
    if T == 8:
        assert bool(quiet.all()) and int(n_det.sum()) == 0
    # checksum of checksums, bit-reproducible across runs
    c1 = scores.view(torch.int32).to(torch.int64).sum().item()
    scores2 = torch.empty_like(scores)
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores2.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert scores2.view(torch.int32).to(torch.int64).sum().item() == c1 and torch.equal(scores, scores2)
    # batch invariance + oracle agreement on sampled streams (first, last, straddling tiles)
    for s in (0, 1, S // 2 + 3, S - 1):
        small = pcm[s:s + 1].clone()
        sc_small = torch.empty((1, 297, T), dtype=torch.float32, device="cuda")
        ag_small = torch.empty((1, 297), dtype=torch.float32, device="cuda")
        ctx.batch_detect_dev(small.data_ptr(), 1, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, sc_small.data_ptr(), ag_small.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(sc_small[0], scores[s])
    for s in (0, S - 1):
        ref_pcm = orc.synth_pcm(SEED, s, N)
        assert np.array_equal(pcm[s].cpu().numpy(), ref_pcm)
        ref_s, _ = orc.score_stream(orc.mfcc_stream(ref_pcm, 5)[:L + 40], templates)  # first 41 windows suffice
        got = scores[s, :ref_s.shape[0]].cpu().numpy()
        assert rel_close(got, ref_s)
