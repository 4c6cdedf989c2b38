"""GPU tests added in round 3 (all through the C ABI): live-stream batches whose detectors hold several wakewords and / or a
wakeword model, the per-stream "can fire" flags between the aggregate pass and the scan, the tc-4 split of small DTW
batches, and the line-streaming MLP kernel at odd batch sizes / unaligned row starts."""
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = simstream.GOLDEN
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


def _wav_i16(x, rate=16000):
    import struct
    raw = np.asarray(x, np.int16).tobytes()
    fmt = struct.pack("<HHIIHH", 1, 1, rate, rate * 2, 2, 16)
    return b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(raw)) + raw


def _load_rpw_bytes(data, tmp_path):
    p = os.path.join(str(tmp_path), "w.rpw")
    with open(p, "wb") as f:
        f.write(data)
    return rpw_py.load_rpw(p)


def _rd(f):
    return simstream.i16_to_f32(rpw_py.read_wav_i16(os.path.join(G, f))[0])


def _two_wakeword_streams():
    z = np.zeros(16000 * 2, np.float32)
    base = np.concatenate([z, _rd("oye_casa_g_1.wav"), z, _rd("alexa.wav"), z, _rd("oye_casa_g_2.wav"), z, _rd("alexa2.wav"), z, z])
    rng = np.random.default_rng(12)
    n = (len(base) // 480) * 480
    return np.stack([base[:n], np.roll(base[:n], 480 * 13) + rng.standard_normal(n).astype(np.float32) * np.float32(0.001),
                     np.roll(base[:n], 480 * 41)])


def _feed(sb, pcm, pieces, max_det=4):
    """pcm [S][N] through the live-stream batch in calls of pieces[i % len] chunks; returns per stream the list of
    (det record, wakeword, label)."""
    S, N = pcm.shape
    out = [[] for _ in range(S)]
    pos, k = 0, 0
    while pos < N:
        nc = min(pieces[k % len(pieces)], (N - pos) // 480)
        k += 1
        det, dww, dlab, n_det = sb.process_multi(pcm[:, pos:pos + 480 * nc], max_det=max_det)
        pos += 480 * nc
        for s in range(S):
            assert n_det[s] <= max_det
            for j in range(n_det[s]):
                out[s].append((det[s][j].copy(), int(dww[s][j]), int(dlab[s][j])))
    return out


def _same(rec, ref):
    return all(rec[f] == ref[f] for f in ("frame", "window", "counter")) and rec["score"].tobytes() == ref["score"].tobytes() and \
        rec["avg_score"].tobytes() == ref["avg_score"].tobytes()


@pytest.mark.parametrize("pieces", [(1,), (3, 1, 2), (4,)])
def test_stream_batch_two_references_equals_batch_detect_multi(ra, ctx, pieces):
    """rp_stream_batch_new_multi with two wakeword references (one with its own threshold): fed piece by piece, every stream
    reports the detections of rp_batch_detect_multi over the whole stream -- frame, window, counter, both scores bit for
    bit, and the wakeword that fired (run_wakeword_detectors, src/detector.rs:433-447)."""
    pcm = _two_wakeword_streams()
    wws = [rpw_py.load_rpw(os.path.join(G, f)) for f in ("oye_casa_g.rpw", "alexa.rpw")]
    tms = [ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"]) for w in wws]
    cfg = ra.DetectorConfig()
    cfg.threshold, cfg.avg_threshold, cfg.min_scores = 0.5, 0.2, 3
    for thr in ([None, None], [None, 0.52]):
        det, dww, n_det = ctx.batch_detect_multi(pcm, tms, cfg, thresholds=thr)
        sb = ra.StreamBatch(ctx, None, cfg, pcm.shape[0], max_chunks_per_call=max(pieces), mfcc_size=5,
                            wakewords=[{"templates": tms[0], "threshold": thr[0]}, {"templates": tms[1], "threshold": thr[1]}])
        got = _feed(sb, pcm, pieces)
        fired = set()
        for s in range(pcm.shape[0]):
            assert len(got[s]) == n_det[s], (s, len(got[s]), n_det[s])
            for j, (rec, w, lab) in enumerate(got[s]):
                assert _same(rec, det[s][j]) and w == dww[s][j] and lab == -1 and rec["stream"] == s
                fired.add(w)
        assert fired == {0, 1} and n_det.sum() >= 6


def _close(rec, ref, rel=2e-6):
    return all(rec[f] == ref[f] for f in ("frame", "window", "counter")) and abs(float(rec["score"]) - float(ref["score"])) <= rel * abs(float(ref["score"])) and \
        abs(float(rec["avg_score"]) - float(ref["avg_score"])) <= rel * max(abs(float(ref["avg_score"])), 1e-30)


@pytest.mark.parametrize("whole_stream_kernel", [False, True])
def test_stream_batch_model_equals_batch_detect_model(ra, ctx, whole_stream_kernel):
    """A wakeword model in a live-stream batch against rp_batch_detect_model over the concatenation: same detections, same
    labels.  With both sides on mlp_mfma_kernel (windows read in place from the frame rows; RP_MLP_WINDOWS=0) the scores
    are equal bit for bit; by default the offline batch takes mlp_windows_kernel (a stream's frames staged once, 16-wide
    k-steps) while a live call with its few new windows per stream keeps mlp_mfma_kernel (32-wide k-steps): the same
    products summed in another order -- scores within 2e-6."""
    if not whole_stream_kernel:
        os.environ["RP_MLP_WINDOWS"] = "0"
    try:
        _stream_batch_model_case(ra, ctx, _close if whole_stream_kernel else _same)
    finally:
        os.environ.pop("RP_MLP_WINDOWS", None)


def _stream_batch_model_case(ra, ctx, same):
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    ws = [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]]
    bs = [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]]
    model = ra.Model(ctx, ws, bs)
    none_index = m["labels"].index("none")
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
    for avg_threshold in (0.0, 0.3):
        cfg = ra.DetectorConfig()
        cfg.avg_threshold, cfg.threshold, cfg.min_scores = avg_threshold, 0.6, 3
        det, dlab, n_det = ctx.batch_detect_model(pcm, model, m["mfcc_size"], none_index, cfg)
        assert n_det.sum() >= 2
        # (12 chunks per call = 36 new windows per stream: the live call takes mlp_windows_kernel too, on rows of the ring's frame pitch)
        for pieces in ((1,), (2, 5, 1), (12, 3)):
            sb = ra.StreamBatch(ctx, None, cfg, pcm.shape[0], max_chunks_per_call=max(pieces), mfcc_size=m["mfcc_size"],
                                wakewords=[{"model": model, "none_index": none_index, "precision": "f32"}])
            got = _feed(sb, pcm, pieces)
            for s in range(pcm.shape[0]):
                assert len(got[s]) == n_det[s]
                for j, (rec, w, lab) in enumerate(got[s]):
                    assert same(rec, det[s][j]) and w == 0 and lab == dlab[s][j]


def test_stream_batch_reference_and_model_equal_rustpotter_handles(ra, ctx, tmp_path):
    """A detector that holds a wakeword reference AND a wakeword model of the same mfcc_size (any mix, src/detector.rs:304-346):
    the live-stream batch against one Rustpotter handle per stream that was given both, chunk by chunk -- same chunks fire,
    same wakeword / label, same counters, scores to 1e-6.  The window is as long as the model's 195 frames and the
    reference scores its oldest frames."""
    K = 16
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    ws = [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]]
    bs = [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]]
    model = ra.Model(ctx, ws, bs)
    none_index = m["labels"].index("none")
    x48, sr, _ = rpw_py.read_wav(os.path.join(G, "ok_casa.wav"))
    speech = orc.resample_stream(x48, sr)
    rng = np.random.default_rng(5)
    n = 480 * 500
    # a reference of mfcc_size 16 built on the device from three noisy copies of a synthetic utterance.  The tiny model answers
    # noise and bursts with scores up to ~0.95, as it does its own recording, and where both wakewords pass the better score
    # wins the frame (src/detector.rs:445) -- so the detector's threshold is 0.9 (the model fires now and then) and the
    # reference carries its own threshold 0.5 (Option<f32> in the .rpw); the utterance is the first candidate on which the model
    # has no window at all above 0.9 (so that one stream certainly belongs to the reference)
    c = ra.RustpotterConfig.default()
    c.detector.avg_threshold, c.detector.threshold, c.detector.min_scores = 0.2, 0.9, 3
    utt, quiet_stream = None, None
    probe_cfg = ra.DetectorConfig()
    probe_cfg.avg_threshold, probe_cfg.threshold, probe_cfg.min_scores = 0.2, 0.9, 1   # min_scores 1: any passing window shows
    for seed in range(77, 117):
        cand = orc.synth_pcm(SEED + seed, 3, 480 * 30) * np.float32(0.3)
        cand *= np.linspace(0.05, 1.0, len(cand), dtype=np.float32) ** (seed % 3)
        st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
        st[120000:120000 + len(cand)] += cand
        _, _, n_probe = ctx.batch_detect_model(st[None, :], model, K, none_index, probe_cfg)
        if n_probe[0] == 0:   # the model has no window above its thresholds anywhere in this stream
            utt, quiet_stream = cand, st
            break
    assert utt is not None, "every candidate utterance triggers the model"
    wavs = {}
    for i in range(3):
        v = utt + rng.standard_normal(len(utt)).astype(np.float32) * np.float32(0.003)
        wavs["u%d.wav" % i] = _wav_i16((np.clip(v, -1, 1) * 32767).astype(np.int16))
    rpw = ctx.build_wakeword_ref("utt", wavs, K, threshold=0.5)
    ref = _load_rpw_bytes(rpw, tmp_path)
    tm = ra.Templates(ctx, list(ref["samples_features"].values()), avg=ref["avg_features"])
    # stream 0: the utterance alone (only the reference can fire); the others: the model's recording, with and without the utterance
    streams = [quiet_stream]
    for a, b in ((20000, 120000), (90000, None), (50000, 160000), (140000, 30080), (10000, None), (70000, None)):
        st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
        st[a:a + len(speech)] += speech
        if b is not None:
            st[b:b + len(utt)] += utt
        streams.append(st)
    pcm = np.stack(streams)
    sb = ra.StreamBatch(ctx, None, c.detector, pcm.shape[0], max_chunks_per_call=3, mfcc_size=K,
                        wakewords=[{"templates": tm, "threshold": ref["threshold"]}, {"model": model, "none_index": none_index, "precision": "f32"}])
    assert abs(ref["threshold"] - 0.5) < 1e-7
    got = _feed(sb, pcm, (3, 1, 2))
    names_seen = set()
    for s in range(pcm.shape[0]):
        rp = ra.Rustpotter.new(c)
        rp.add_wakeword_from_buffer("utt", rpw)
        rp.add_wakeword_from_file("model", os.path.join(G, "ok_casa-tiny.rpw"))
        want = []
        for i in range(0, n, 480):
            d = rp.process_samples(pcm[s, i:i + 480].copy())
            if d is not None:
                want.append((i // 480, d))
        assert len(got[s]) == len(want), (len(got[s]), len(want))
        for (rec, w, lab), (chunk, d) in zip(got[s], want):
            assert rec["frame"] // 3 + 1 == chunk and rec["counter"] == d.counter
            assert abs(rec["score"] - d.score) <= 1e-5 * max(d.score, 1e-3) and abs(rec["avg_score"] - d.avg_score) <= 1e-5 * max(d.avg_score, 1e-3)
            name = "utt" if w == 0 else m["labels"][lab]
            assert name == d.name and (lab == -1) == (w == 0)
            names_seen.add(w)
    assert names_seen == {0, 1}


def test_stream_batch_multi_refuses_bad_specs(ra, ctx):
    cfg = ra.DetectorConfig()
    tm5 = ra.Templates(ctx, orc.synth_templates(SEED, 2, 40, 5))
    tm16 = ra.Templates(ctx, orc.synth_templates(SEED, 2, 40, 16))
    with pytest.raises(ra.RustpotterError, match="different mfcc size"):
        ra.StreamBatch(ctx, None, cfg, 4, mfcc_size=5, wakewords=[{"templates": tm5}, {"templates": tm16}])
    with pytest.raises(ra.RustpotterError, match="reference OR a model"):
        ra.StreamBatch(ctx, None, cfg, 4, mfcc_size=5, wakewords=[{}])
    with pytest.raises(ra.RustpotterError, match="1..8 wakewords"):
        ra.StreamBatch(ctx, None, cfg, 4, mfcc_size=5, wakewords=[{"templates": tm5}] * 9)
    sb = ra.StreamBatch(ctx, None, cfg, 2, mfcc_size=5, wakewords=[{"templates": tm5}])
    with pytest.raises(ra.RustpotterError):   # no single aggregate per window in a multi batch
        sb.process(np.zeros((2, 480), np.float32), want_agg=True)
    # a plain one-reference batch answers process_multi with wakeword 0 / label -1
    one = ra.StreamBatch(ctx, tm5, cfg, 2)
    det, dww, dlab, n_det = one.process_multi(np.zeros((2, 480), np.float32))
    assert n_det.sum() == 0 and (dww == 0).all() and (dlab == -1).all()


# ------------------------------------------------------------------ small batches: hot flags + tc-4 split
def test_quiet_streams_skip_the_scan_and_loud_ones_do_not(ra, ctx):
    """The aggregate pass raises a flag per stream that has a window above the threshold; scan_kernel returns at once for the
    others.  Mixed batch: quiet noise streams between streams that hold the utterance -- detections equal a batch of the
    loud streams alone and the oracle's chunked detector; n_det of the quiet ones is 0 and their slots are zero."""
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    n = (len(base) // 480) * 480
    rng = np.random.default_rng(3)
    quiet = [rng.standard_normal(n).astype(np.float32) * np.float32(0.01) for _ in range(5)]
    loud = [base[:n], np.roll(base[:n], 480 * 7)]
    pcm = np.stack([quiet[0], loud[0], quiet[1], quiet[2], loud[1], quiet[3], quiet[4]])
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    tm = ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])
    for avg_threshold in (0.0, 0.2):
        cfg = ra.DetectorConfig()
        cfg.threshold, cfg.avg_threshold = 0.45, avg_threshold
        det, n_det = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        det_l, n_l = ctx.batch_detect(np.stack(loud), tm, cfg, max_det=4)
        assert list(n_det) == [0, n_l[0], 0, 0, n_l[1], 0, 0] and n_l.min() >= 1
        for s, ls in ((1, 0), (4, 1)):
            for j in range(n_det[s]):
                assert all(det[s][j][f] == det_l[ls][j][f] for f in ("frame", "window", "counter", "score", "avg_score")) and det[s][j]["stream"] == s
        for s in (0, 2, 3, 5, 6):
            assert det[s].tobytes() == bytes(det[s].nbytes)
        # and with the per-window arrays requested (every window scored): same detections
        det2, n2, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
        assert np.array_equal(n2, n_det) and det2.tobytes() == det.tobytes()


@pytest.mark.parametrize("S", [64, 700, 1024])
def test_small_batches_tc4_split_gives_the_same_scores(ra, S):
    """A batch whose tc-8 DTW waves would fill the chip less than three times is scored by tc-4 half chunks instead
    (launch_dtw_k5).  Same operations per cell, so the scores are bit-identical to the tc-8 launch (RP_DTW_NO_SPLIT=1 in a
    child process) and within 1e-5 of the oracle."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import rustpotter_amd as ra
from oracle import rp_oracle as orc
S, SEED = %d, 0x5EED000000000001
ctx = ra.BatchContext(0)
templates = orc.synth_templates(SEED, 8, 60, 5)
pcm = ctx.synth_pcm(SEED, 0, S, 480 * 50)
mf = ctx.mfcc(pcm, 5)
scores, _, agg = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
np.save(sys.argv[1], scores)
""" % (ROOT, os.path.join(ROOT, "tests"), S)
    import tempfile
    outs = []
    for env_extra in ({}, {"RP_DTW_NO_SPLIT": "1"}):
        with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as f:
            path = f.name
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.unlink(path)
    assert outs[0].tobytes() == outs[1].tobytes()
    templates = orc.synth_templates(SEED, 8, 60, 5)
    for s in (0, S // 2, S - 1):
        ref_s, _ = orc.score_stream(orc.mfcc_stream(orc.synth_pcm(SEED, s, 480 * 50), 5), templates)
        assert np.all(np.abs(outs[0][s] - ref_s) <= 1e-5 * np.abs(ref_s))


# ------------------------------------------------------------------ the line-streaming MLP kernel
@pytest.mark.parametrize("dims", [(3120, 32, 16, 2), (3120, 13, 2), (1040, 32, 16, 3), (64, 13, 2), (4096, 20, 255, 4)])
@pytest.mark.parametrize("B", [1, 15, 257, 1300])
def test_mlp_stream_kernel_shapes_and_batch_sizes(ra, ctx, dims, B):
    """mlp_stream_kernel (layer-1 width <= 32, row pitch a multiple of 64 bytes) at batch sizes that leave ragged last tiles,
    with row pitches that put odd rows in the middle of a 128-byte line (3120, 1040 floats) and ones that do not (64, 4096),
    against the oracle (f32 1e-5; bf16 vs the bf16-rounding oracle 1e-3); bit-identical to itself on a second call and
    independent of where a row sits in the batch."""
    os.environ["RP_MLP_STREAM"] = "2"   # the stream kernel for f32 too (the library picks it for bf16 only by default)
    rng = np.random.default_rng(sum(dims) + B)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
    x = rng.standard_normal((B, dims[0])).astype(np.float32)
    model = ra.Model(ctx, ws, bs)
    got = ctx.mlp_forward(x, model)
    ref = orc.mlp_forward(x, ws, bs)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5), np.abs(got - ref).max()
    assert ctx.mlp_forward(x, model).tobytes() == got.tobytes()
    if B > 2:   # a row's logits do not depend on its position (even / odd rows start at different line phases)
        sub = ctx.mlp_forward(x[1:], model)
        assert sub.tobytes() == got[1:].tobytes()
    if len(dims) > 3 and dims[2] > 200:
        # a hidden layer this wide does not fit the fused kernels' LDS: the per-layer f32 kernel serves it, there is no bf16 form
        with pytest.raises(ra.RustpotterError, match="no bf16 MFMA kernel"):
            ctx.mlp_forward(x, model, precision="bf16")
        os.environ.pop("RP_MLP_STREAM")
        return
    got16 = ctx.mlp_forward(x, model, precision="bf16")
    ref16 = orc.mlp_forward(x, ws, bs, bf16_layer1=True)
    assert np.allclose(got16, ref16, rtol=1e-3, atol=1e-3), np.abs(got16 - ref16).max()
    os.environ.pop("RP_MLP_STREAM")
    # and as the library chooses by itself: bf16 streamed; f32 callers (RP_MLP_F32) streamed too, layer 1 from exact three-part bf16 splits
    # of inputs and weights (kMlpBf16x3) -- `got` above is that form already; RP_MLP_F32_FAST = f16 two-way splits (kMlpF16x2: within 1e-5 of
    # the f32 matrix instructions and not their bits), batch-invariant and bit-reproducible like the other forms; a row with a feature beyond
    # the f16 range comes from the f32 matrix instructions there
    assert ctx.mlp_forward(x, model).tobytes() == got.tobytes()
    assert "bf16x3" in ctx.last_mlp_kernel()
    split = ctx.mlp_forward(x, model, precision="f32_fast")
    assert "f16x2" in ctx.last_mlp_kernel()
    assert np.allclose(split, ref, rtol=1e-5, atol=1e-5)
    exact = ctx.mlp_forward(x, model, precision="f32_strict")
    assert "f32 matrix instructions" in ctx.last_mlp_kernel()
    assert np.allclose(split, exact, rtol=1e-5, atol=1e-5), np.abs(split - exact).max()
    assert np.allclose(got, exact, rtol=1e-5, atol=1e-5), np.abs(got - exact).max()
    assert np.allclose(split, got, rtol=1e-5, atol=1e-5), np.abs(split - got).max()
    if B > 200:   # three arithmetics, three sets of bits
        assert split.tobytes() != exact.tobytes() and split.tobytes() != got.tobytes() and got.tobytes() != exact.tobytes()
    assert ctx.mlp_forward(x, model, precision="f32_fast").tobytes() == split.tobytes()
    if B > 2:
        assert ctx.mlp_forward(x[1:], model, precision="f32_fast").tobytes() == split[1:].tobytes()
        # a feature beyond the f16 range: the two-part form has its row computed again by the f32 matrix instructions (round 4; it used to be
        # NaN); the three-part form has the f32 exponent range and needs no second pass
        xb = x.copy()
        xb[1, 7] = 7.0e4
        big = ctx.mlp_forward(xb, model, precision="f32_fast")
        strict = ctx.mlp_forward(xb, model, precision="f32_strict")
        assert big[1].tobytes() == strict[1].tobytes() and np.isfinite(big).all()
        assert np.delete(big, 1, axis=0).tobytes() == np.delete(split, 1, axis=0).tobytes()
        big3 = ctx.mlp_forward(xb, model)
        assert np.isfinite(big3).all() and np.allclose(big3[1], strict[1], rtol=2e-5, atol=2e-5 * 7.0e4)
        assert np.delete(big3, 1, axis=0).tobytes() == np.delete(got, 1, axis=0).tobytes()
    assert ctx.mlp_forward(x, model, precision="bf16").tobytes() == got16.tobytes()


def test_mlp_stream_kernel_unaligned_base_and_nan_neighbours(ra):
    """Device-pointer form with the rows starting at every 16-byte offset inside a 128-byte line (the kernel reads whole lines
    from the line start below a row: what lies before the first row and behind the last one must never reach a result),
    NaN planted around the array and in a neighbouring row: only that row's logits are NaN."""
    import torch
    dims = (3120, 32, 16, 2)
    os.environ["RP_MLP_STREAM"] = "2"
    rng = np.random.default_rng(9)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    dctx = ra.BatchContext(0, host_pointers=False)
    dctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(dctx, ws, bs)
    B = 300
    x = rng.standard_normal((B, dims[0])).astype(np.float32)
    ref = orc.mlp_forward(x, ws, bs)
    outs = []
    for off in range(0, 32, 4):   # float offsets 0, 4, .. 28 = 16-byte steps through a line
        buf = torch.full((off + B * dims[0] + 64,), float("nan"), dtype=torch.float32, device="cuda")
        buf[off:off + B * dims[0]] = torch.from_numpy(x.reshape(-1)).cuda()
        out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()   # torch's default stream is handle 0 = "the context's own stream" to rp_ctx_set_stream: order by hand
        dctx.mlp_dev(model, buf.data_ptr() + 4 * off, B, "f32", out.data_ptr())
        dctx.synchronize()
        o = out.cpu().numpy()
        assert np.isfinite(o).all() and np.allclose(o, ref, rtol=1e-5, atol=1e-5), off
        outs.append(o)
    assert all(o.tobytes() == outs[0].tobytes() for o in outs)   # the k order does not depend on the phase
    xn = x.copy()
    xn[17, 5] = np.nan
    buf = torch.from_numpy(xn.reshape(-1)).cuda()
    out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    dctx.mlp_dev(model, buf.data_ptr(), B, "f32", out.data_ptr())
    dctx.synchronize()
    o = out.cpu().numpy()
    assert np.isnan(o[17]).all() and np.isfinite(np.delete(o, 17, axis=0)).all()
    assert np.delete(o, 17, axis=0).tobytes() == np.delete(outs[0], 17, axis=0).tobytes()
    os.environ.pop("RP_MLP_STREAM")


# ------------------------------------------------------------------ the averaged-template gate behind the generic DTW kernel
@pytest.mark.parametrize("K,band", [(7, 5), (5, 9), (12, 2)])
def test_avg_gate_skip_behind_the_generic_kernel(ra, K, band):
    """Template sets only dtw_generic_kernel serves (mfcc_size / band_size without a register kernel): a detect-only call skips
    the sample templates for every wave of 64 windows whose averaged-template scores are all below avg_threshold
    (wakeword_comp.rs:85-93).  Same detections, bit for bit, as RP_CTX_FULL_SCORES, at thresholds that reject nothing, about
    half, and almost everything; and the oracle's chunked detector agrees on chunk and counter."""
    rng = np.random.default_rng(K * 31 + band)
    L, n = 50, 480 * 260
    utt = (orc.synth_pcm(SEED + 5, 1, 480 * 20) * np.float32(0.4)).astype(np.float32)
    templates = []
    for i in range(3):
        v = utt + rng.standard_normal(len(utt)).astype(np.float32) * np.float32(0.004)
        m = orc.mfcc_stream(v, K)
        templates.append(np.ascontiguousarray((m - m.mean(axis=0, dtype=np.float32))[:L], np.float32))
    avg_t = np.ascontiguousarray(np.mean(templates, axis=0, dtype=np.float32), np.float32)
    streams = []
    for s in range(6):
        st = rng.standard_normal(n).astype(np.float32) * np.float32(0.003)
        if s % 2 == 0:
            p = 160 * (200 + 37 * s)
            st[p:p + len(utt)] += utt
        streams.append(st)
    pcm = np.stack(streams)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    tg, tf = ra.Templates(gated, templates, avg=avg_t), ra.Templates(full, templates, avg=avg_t)
    cfg = ra.DetectorConfig()
    cfg.band_size, cfg.threshold, cfg.min_scores = band, 0.45, 1
    some = False
    for avg_threshold in (0.05, 0.35, 0.6):
        cfg.avg_threshold = avg_threshold
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=4)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=4)
        assert np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes(), avg_threshold
        some = some or n_f.sum() > 0
        from collections import OrderedDict
        for s in (0, 1):
            d = orc.Detector(avg_threshold=avg_threshold, threshold=0.45, band_size=band, min_scores=1)
            d.add_ref({"name": "w", "samples_features": OrderedDict(("t%d" % i, t) for i, t in enumerate(templates)), "avg_features": avg_t,
                       "threshold": None, "avg_threshold": None, "rms_level": 0.0})
            want = [(i // 480, r) for i in range(0, n, 480) for r in [d.process_f32(pcm[s, i:i + 480])] if r is not None]
            assert len(want) == n_f[s]
            for j, (chunk, r) in enumerate(want):
                assert det_f[s][j]["frame"] // 3 + 1 == chunk and det_f[s][j]["counter"] == r["counter"]
    assert some


def test_live_multi_sweep_few_cases(ra, ctx):
    """A few cases of the randomised live-stream sweep with several wakewords / a model per detector (tests/sweep_parity.py
    --live-multi-cases; 150 cases in profiles/sweep_r03.txt): the oracle's chunked detector and, for references only, the
    offline batch bit for bit."""
    import sweep_parity
    n, total, with_model, ties = sweep_parity.run_live_multi_sweep(ra, ctx, 20, seed=7)
    assert n == 20 and total >= 5 and ties == 0
