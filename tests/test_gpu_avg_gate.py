"""The averaged-template gate as a skip (wakeword_comp.rs:85-93), through the C ABI: windows whose averaged-template score is below
avg_threshold are never compared with the sample templates -- same detections as full scoring, on the reference's recordings and synthetic
noise, for wide frames, in live-stream batches and with several wakewords."""
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


def _fixture_streams(n_variants=6):
    """Variants of the reference's simulation stream (tests/detector.rs:372-426): shifted by whole chunks, with noise."""
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    rng = np.random.default_rng(7)
    n = (len(base) // 480) * 480
    out = [base[:n]]
    for i in range(1, n_variants):
        v = np.roll(base, 480 * (3 + 5 * i))
        if i % 2:
            v = v + rng.standard_normal(len(base)).astype(np.float32) * np.float32(0.001 * i)
        out.append(v[:n].astype(np.float32))
    return np.stack(out)


def _wakeword(ra, ctx, name="oye_casa_g.rpw"):
    w = rpw_py.load_rpw(os.path.join(G, name))
    return ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])


# ------------------------------------------------------------------ the averaged-template gate as a skip
@pytest.mark.parametrize("avg_threshold,threshold", [(0.2, 0.5), (0.4, 0.45), (0.55, 0.45), (0.62, 0.3), (0.9, 0.3)])
def test_avg_gate_skip_gives_the_detections_of_full_scoring(ra, avg_threshold, threshold):
    """wakeword_comp.rs:85-93: a window whose avg_score is below avg_threshold is not compared with the sample templates.
    The skipping path (one DTW for gated windows) and RP_CTX_FULL_SCORES (T+1 DTWs for every window) must report the
    same detections, field by field and bit by bit, whatever part of the windows the gate removes."""
    pcm = _fixture_streams()
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold, cfg.threshold = avg_threshold, threshold
    det_g, n_g = gated.batch_detect(pcm, _wakeword(ra, gated), cfg)
    det_f, n_f = full.batch_detect(pcm, _wakeword(ra, full), cfg)
    assert np.array_equal(n_g, n_f)
    assert det_g.tobytes() == det_f.tobytes()
    # the per-window arrays stay complete when they are asked for (then nothing is skipped)
    det_s, n_s, scores, agg = gated.batch_detect(pcm, _wakeword(ra, gated), cfg, want_scores=True)
    assert np.array_equal(n_s, n_f) and det_s.tobytes() == det_f.tobytes()
    assert np.isfinite(scores).all() and np.isfinite(agg).all()
    if avg_threshold <= 0.55:
        assert n_f.sum() >= 1   # the planted utterances are found
    if avg_threshold >= 0.9:
        assert n_f.sum() == 0   # nothing passes the gate: the list is empty and every tile of pass 3 exits


def test_avg_gate_skip_on_synthetic_noise_many_streams(ra):
    """Many streams whose windows straddle the gate (threshold at the median avg_score): detections with a low score
    threshold must agree between the two paths; band sizes 3..6 take the same route."""
    S, N, K, L, T = 700, 480 * 60, 5, 40, 5
    templates = orc.synth_templates(SEED, T, L, K)
    avg = np.mean(templates, axis=0, dtype=np.float32)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    pcm = gated.synth_pcm(SEED, 0, S, N)
    tg, tf = ra.Templates(gated, templates, avg=avg), ra.Templates(full, templates, avg=avg)
    mf = gated.mfcc(pcm, K)
    for band, q in ((5, 0.5), (3, 0.5), (6, 0.5), (5, 0.0), (5, 0.03), (4, 0.97)):
        # q: the quantile of the avg scores the gate sits at -- 0.5: half of the windows are listed (list mode); 0 / 0.03: all /
        # nearly all pass (the dense form of pass 3: every window through the staged kernels); 0.97: a short list
        _, av, ag = gated.dtw_scores(mf, tg, band_size=band, with_avg=True)
        cfg = ra.DetectorConfig()
        cfg.band_size = band
        cfg.avg_threshold = float(np.quantile(av, q))
        cfg.threshold = float(np.quantile(ag, 0.7))
        cfg.min_scores = 2
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=6)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=6)
        assert n_f.sum() > (S // 4 if q <= 0.5 else 0), "the case must produce detections"
        assert np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes(), (band, q)


@pytest.mark.parametrize("K", [13, 16])
def test_avg_gate_skip_wide_frames(ra, K):
    """The same for mfcc_size 13 / 16 (dtw_band_wide_kernel in list mode), ragged templates so that one- and two-template
    chunks both occur."""
    S, N, L, T = 300, 480 * 50, 40, 5
    templates = orc.synth_templates(SEED, T, L, K)
    templates[1] = templates[1][:L - 5].copy()
    templates[3] = templates[3][:L - 9].copy()
    avg = np.mean([t[:L - 9] for t in templates], axis=0, dtype=np.float32)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    pcm = gated.synth_pcm(SEED, 0, S, N)
    tg, tf = ra.Templates(gated, templates, avg=avg), ra.Templates(full, templates, avg=avg)
    mf = gated.mfcc(pcm, K)
    for band in (5, 4):
        _, av, ag = gated.dtw_scores(mf, tg, band_size=band, with_avg=True)
        cfg = ra.DetectorConfig()
        cfg.band_size, cfg.min_scores = band, 2
        cfg.avg_threshold, cfg.threshold = float(np.median(av)), float(np.quantile(ag, 0.7))
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=6)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=6)
        assert n_f.sum() > S // 4 and np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes()


@pytest.mark.parametrize("K,T,kernel", [(16, 8, "dtw_mfma_wide_kernel"), (13, 6, "dtw_mfma_wide_kernel"), (5, 4, "dtw_mfma_kernel"), (5, 11, "dtw_mfma_kernel")])
def test_avg_gate_skip_through_the_three_part_matrix_kernels_list_mode(ra, K, T, kernel):
    """Same-length template sets in the default arithmetic: the windows that pass the gate are a LIST for dtw_mfma_wide3_kernel (mfcc_size
    13 / 16, chunks of up to four) and for both shapes of dtw_mfma_kernel (a chunk of four; eight + three) -- the detections are those of
    full scoring bit for bit, and the library reports the three-part products."""
    S, N, L = 200, 480 * 50, 40
    templates = orc.synth_templates(SEED + K, T, L, K)
    avg = np.mean(templates, axis=0, dtype=np.float32)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    pcm = gated.synth_pcm(SEED, 0, S, N)
    tg, tf = ra.Templates(gated, templates, avg=avg), ra.Templates(full, templates, avg=avg)
    _, av, ag = gated.dtw_scores(gated.mfcc(pcm, K), tg, with_avg=True)
    cfg = ra.DetectorConfig()
    cfg.min_scores = 2
    cfg.avg_threshold, cfg.threshold = float(np.quantile(av, 0.6)), float(np.quantile(ag, 0.7))
    gated.dtw_kernels()
    det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=6)
    assert kernel in gated.dtw_kernels() and gated.last_dtw_products == ["bf16x3"]
    det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=6)
    assert n_f.sum() > S // 4 and np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes()


@pytest.mark.parametrize("avg_threshold,chunks_per_call", [(0.2, 1), (0.5, 1), (0.5, 4), (0.62, 7)])
def test_avg_gate_skip_in_live_stream_batches(ra, avg_threshold, chunks_per_call):
    """rp_stream_batch_process skips the sample templates of gated windows too: fed chunk by chunk it reports the detections
    of the offline call that scores everything."""
    pcm = _fixture_streams(5)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold, cfg.threshold = avg_threshold, 0.45
    full = ra.BatchContext(0, full_scores=True)
    det_f, n_f = full.batch_detect(pcm, _wakeword(ra, full), cfg, max_det=6)
    ctx = ra.BatchContext(0)
    sb = ra.StreamBatch(ctx, _wakeword(ra, ctx), cfg, pcm.shape[0], max_chunks_per_call=chunks_per_call)
    live = [[] for _ in range(pcm.shape[0])]
    step = 480 * chunks_per_call
    for i in range(0, pcm.shape[1], step):
        d, nd = sb.process(np.ascontiguousarray(pcm[:, i:i + step]), max_det=8)
        for s in range(pcm.shape[0]):
            live[s] += [d[s][j] for j in range(nd[s])]
    for s in range(pcm.shape[0]):
        assert len(live[s]) == n_f[s]
        for a, b in zip(live[s], det_f[s][:n_f[s]]):
            assert (a["frame"], a["window"], a["counter"]) == (b["frame"], b["window"], b["counter"])
            assert a["score"] == b["score"] and a["avg_score"] == b["avg_score"]
    assert n_f.sum() >= (5 if avg_threshold <= 0.5 else 0)


def test_avg_gate_skip_with_several_wakewords(ra):
    """rp_batch_detect_multi: each wakeword's own avg_threshold gates its own sample templates; same detections and same
    firing wakeword as the path that scores everything."""
    pcm = _fixture_streams(4)
    cfg = ra.DetectorConfig()
    cfg.threshold, cfg.min_scores = 0.45, 3
    outs = []
    for full in (False, True):
        ctx = ra.BatchContext(0, full_scores=full)
        tms = [_wakeword(ra, ctx, "oye_casa_g.rpw"), _wakeword(ra, ctx, "alexa.rpw")]
        outs.append(ctx.batch_detect_multi(pcm, tms, cfg, avg_thresholds=[0.5, 0.3]))
    (d0, w0, n0), (d1, w1, n1) = outs
    assert np.array_equal(n0, n1) and n0.sum() >= 4 and d0.tobytes() == d1.tobytes() and np.array_equal(w0, w1)
