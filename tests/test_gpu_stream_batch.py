"""Live-stream batches (rp_stream_batch_*) whose detectors hold several wakewords and / or a wakeword model: equal to the offline calls over
the concatenation and to the single-stream Rustpotter handles; bad specs are refused; a few cases of the live multi-wakeword sweep."""
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
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


def test_live_multi_sweep_few_cases(ra, ctx):
    """A few cases of the randomised live-stream sweep with several wakewords / a model per detector (tests/sweep_parity.py
    --live-multi-cases; 150 cases in profiles/sweep_r03.txt): the oracle's chunked detector and, for references only, the
    offline batch bit for bit."""
    import sweep_parity
    n, total, with_model, ties = sweep_parity.run_live_multi_sweep(ra, ctx, 20, seed=7)
    assert n == 20 and total >= 5 and ties == 0
