"""The arithmetic domain of the scale-invariant cosine (src/mfcc/comparator.rs:28-48): the reference divides by sqrt(dot_a * dot_b) in f32 and
answers similarity 0 when that product underflows to 0; the device kernels hand the (window, templates) pairs whose squared norms leave
2^-60 .. 2^30 / 2^60 to dtw_ref_kernel (rp_dtw.hip), which forms the cell as the reference does.  Scores against the oracle at 1e-5 for tiny /
huge windows and templates, every kernel family, lists that overflow, detectors with a template set outside the range."""
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


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


def _streams(S, n_frames, K=5, first=0):
    n = 480 * (n_frames // 3 + 2)
    mf = [orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(S)]
    assert all(m.shape[0] == n_frames for m in mf)
    return np.stack(mf)


# (mfcc_size, templates, frames, windows per stream): which kernel family scores them
SHAPES = [
    (5, 8, 100, 70),    # dtw_mfma_kernel, eight template slots (BASELINE C2 / C3 shape), LDS-staged tiles
    (5, 4, 32, 40),     # dtw_mfma_kernel, four slots
    (5, 2, 40, 70),     # dtw_band_kernel<5, 5, 2>
    (5, 1, 37, 130),    # dtw_band2_kernel (two windows per lane)
    (16, 3, 30, 40),    # dtw_band_wide_kernel (rp_dtw_score_batch rows have no slack: the register kernels)
    (13, 2, 25, 40),    # dtw_band_wide_kernel<13, 5, 2>
    (7, 3, 20, 40),     # dtw_generic_kernel (no register kernel for mfcc_size 7)
]


@pytest.mark.parametrize("K,T,L,n_win", SHAPES)
def test_cosine_norm_product_out_of_range_matches_the_oracle(ra, ctx, K, T, L, n_win):
    """Windows and / or templates scaled so that the reference's f32 product dot_a * dot_b is subnormal, zero or infinite
    (comparator.rs:42-47).  The verdict's probe: one pair scaled by s gives 0.310666 for s >= 1e-9, 0.310842 at 1e-11,
    0.218791 at 1e-12 in the oracle -- a scale-invariant kernel says 0.3107 throughout."""
    S = 2
    templates = orc.synth_templates(SEED + 17 * K + L, T, L, K)
    mf = _streams(S, n_win + L - 1, K, first=300 + L)
    base_t = ra.Templates(ctx, templates)
    base, _, _ = ctx.dtw_scores(mf, base_t)
    worst = 0.0
    differs = 0
    # (window scale, template scale)
    cases = [(s, s) for s in (1e-9, 1e-10, 1e-11, 1e-12, 1e-13)] + [(1e-20, 1.0), (1.0, 1e-20), (1e-22, 1.0), (1.0, 3e-23), (1e-30, 1.0),
                                                                     (1.0, 1e-30), (1e-24, 1e-3), (3e-19, 1.0), (1e12, 1.0), (1e17, 1e3),
                                                                     (1.0, 1e17), (1e15, 1e-15)]
    for sw, st in cases:
        mfs = (mf.astype(np.float64) * sw).astype(np.float32)
        ts = [(t.astype(np.float64) * st).astype(np.float32) for t in templates]
        before = ctx.dtw_ref_pairs()
        tm = ra.Templates(ctx, ts)
        scores, _, agg = ctx.dtw_scores(mfs, tm)
        for s in range(S):
            ref_s, ref_a = orc.score_stream(mfs[s], ts)
            e = max(rel_err(scores[s], ref_s), rel_err(agg[s], ref_a))
            worst = max(worst, e)
            assert e <= 1e-5, (sw, st, e)
        if rel_err(scores, base) > 1e-4:
            differs += 1
            # (a side scaled by 1e-24 or less: every square underflows in f32, its vectors ARE zero vectors to both forms -- no rescoring needed)
            assert ctx.dtw_ref_pairs() > before or min(sw, st) < 2e-23, "a score that depends on the scale can only come from the reference-shaped cell"
    assert differs >= 8, "the scales above must reach the product's underflow / overflow"
    # ordinary data: nothing is rescored
    before = ctx.dtw_ref_pairs()
    again, _, _ = ctx.dtw_scores(mf, base_t)
    assert ctx.dtw_ref_pairs() == before and np.array_equal(again, base)


def test_cosine_out_of_range_in_single_windows_only(ra, ctx):
    """Only a few windows of a stream hold tiny frames (a burst of near-constant MFCC rows): those pairs are rescored, the rest
    keep the fast kernels' bits."""
    K, T, L, n_win = 5, 8, 60, 200
    templates = orc.synth_templates(SEED + 99, T, L, K)
    mf = _streams(3, n_win + L - 1, K, first=77)
    tm = ra.Templates(ctx, templates)
    base, _, _ = ctx.dtw_scores(mf, tm)
    mfs = mf.copy()
    # frames 100..100+L+5 of stream 1 are constant + 1e-12 noise: windows starting at 100..105 see only tiny centred frames
    rng = np.random.default_rng(4)
    mfs[1, 100:100 + L + 6] = np.float32(1e-11) * rng.standard_normal((L + 6, K)).astype(np.float32)
    before = ctx.dtw_ref_pairs()
    scores, _, agg = ctx.dtw_scores(mfs, tm)
    listed = ctx.dtw_ref_pairs() - before
    assert 1 <= listed <= 8, listed   # windows 100..105 (their frames are all tiny; others hold ordinary frames too and so have ordinary means)
    for s in range(3):
        ref_s, ref_a = orc.score_stream(mfs[s], templates)
        assert rel_err(scores[s], ref_s) <= 1e-5 and rel_err(agg[s], ref_a) <= 1e-5
    assert np.array_equal(scores[0], base[0]) and np.array_equal(scores[2], base[2])


def test_cosine_out_of_range_single_stream_handful_of_windows(ra, ctx):
    """rp_dtw_score_batch with one stream and <= 8 windows: dtw_single_kernel (one wave per DTW) forms the reference-shaped
    costs itself."""
    for K, T, L in ((5, 3, 50), (16, 2, 30)):
        templates = orc.synth_templates(SEED + 3 * K, T, L, K)
        mf = _streams(1, L + 4, K, first=9)
        for sw, st in ((1e-12, 1e-12), (1e-20, 1.0), (1.0, 1e-20), (1.0, 1.0)):
            mfs = (mf.astype(np.float64) * sw).astype(np.float32)
            ts = [(t.astype(np.float64) * st).astype(np.float32) for t in templates]
            before = ctx.dtw_ref_pairs()
            scores, _, agg = ctx.dtw_scores(mfs, ra.Templates(ctx, ts))
            ref_s, ref_a = orc.score_stream(mfs[0], ts)
            assert rel_err(scores[0], ref_s) <= 1e-5 and rel_err(agg[0], ref_a) <= 1e-5, (K, sw, st)
            assert (ctx.dtw_ref_pairs() > before) == (sw != 1.0 or st != 1.0)


def test_cosine_out_of_range_averaged_template_longer_than_the_window(ra, ctx):
    """m != n (an averaged template longer than the window, dtw.rs:64-67 widens the band): dtw_generic_kernel lists per template."""
    K = 5
    templates = orc.synth_templates(SEED + 41, 3, 30, K)
    avg = orc.synth_templates(SEED + 42, 1, 36, K)[0]
    mf = _streams(2, 30 + 50, K, first=500)
    for sw, st in ((1e-12, 1e-12), (1e-20, 1.0), (1.0, 1e-20)):
        mfs = (mf.astype(np.float64) * sw).astype(np.float32)
        ts = [(t.astype(np.float64) * st).astype(np.float32) for t in templates]
        av = (avg.astype(np.float64) * st).astype(np.float32)
        tm = ra.Templates(ctx, ts, avg=av)
        scores, avg_s, agg = ctx.dtw_scores(mfs, tm, with_avg=True, score_mode=ra.ScoreMode.Average)
        for s in range(2):
            ref_s, ref_a = orc.score_stream(mfs[s], ts, mode="average")
            assert rel_err(scores[s], ref_s) <= 1e-5 and rel_err(agg[s], ref_a) <= 2e-5
            ref_avg = np.array([orc.score_window(mfs[s][w:w + 30], av) for w in range(scores.shape[1])], np.float32)
            assert rel_err(avg_s[s], ref_avg) <= 1e-5


def test_cosine_range_more_pairs_than_the_list_holds(ra, ctx):
    """The list of out-of-range pairs holds 2^18 entries; beyond that dtw_ref_kernel rescored EVERY window of the call (its ALL
    mode).  128 streams x 260 windows x 8 single-template chunks (eight different lengths) = 266 240 pairs, all of them tiny."""
    K, S, n_win = 5, 128, 260
    rng = np.random.default_rng(11)
    lens = [12, 13, 14, 15, 16, 17, 18, 19]
    templates = [(rng.standard_normal((L, K)) * 1e-12).astype(np.float32) for L in lens]
    templates[3] = (templates[3].astype(np.float64) * 1e12).astype(np.float32)   # one ordinary template among the tiny ones: the set still has rows below the range
    mf = (rng.standard_normal((S, n_win + max(lens) - 1, K)) * 1e-12).astype(np.float32)
    tm = ra.Templates(ctx, templates)
    before = ctx.dtw_ref_pairs()
    scores, _, agg = ctx.dtw_scores(mf, tm, score_mode=ra.ScoreMode.Median)
    assert ctx.dtw_ref_pairs() - before >= S * n_win * len(lens)
    for s in range(0, S, 9):
        ref_s, ref_a = orc.score_stream(mf[s], templates, mode="median")
        assert rel_err(scores[s], ref_s) <= 1e-5 and rel_err(agg[s], ref_a) <= 1e-5
    # ordinary templates, tiny windows: the fast kernels run, list more than 2^18 pairs, and the ALL mode takes over
    templates = [(rng.standard_normal((L, K))).astype(np.float32) for L in lens]
    tm = ra.Templates(ctx, templates)
    before = ctx.dtw_ref_pairs()
    scores, _, agg = ctx.dtw_scores(mf, tm)
    assert ctx.dtw_ref_pairs() - before >= S * n_win * len(lens)
    for s in range(0, S, 9):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_err(scores[s], ref_s) <= 1e-5 and rel_err(agg[s], ref_a) <= 1e-5
    # and the next ordinary call starts from an empty list
    mf1 = _streams(3, 60 + 19 - 1, K, first=7)
    before = ctx.dtw_ref_pairs()
    sc1, _, _ = ctx.dtw_scores(mf1, tm)
    assert ctx.dtw_ref_pairs() == before
    for s in range(3):
        assert rel_err(sc1[s], orc.score_stream(mf1[s], templates)[0]) <= 1e-5


@pytest.mark.parametrize("seed", range(24))
def test_cosine_range_randomised_frames(ra, ctx, seed):
    """Random mixtures inside ONE call: ordinary, zero, tiny (1e-12, 1e-25) and huge (1e10, 1e17) frames and template rows at random
    places, every kernel family by turns -- whatever the mixture, the scores are the oracle's (the reference's arithmetic, with its
    under- and overflows) to 1e-5."""
    rng = np.random.default_rng([4, seed])
    K, T, L, n_win, S = [(5, 8, 40, 70, 3), (5, 4, 33, 40, 2), (5, 2, 25, 70, 2), (5, 1, 20, 130, 2), (16, 3, 22, 40, 2), (13, 2, 18, 40, 2),
                         (7, 3, 15, 40, 2), (5, 3, 30, 4, 1)][seed % 8]
    scales = np.array([1.0, 0.0, 1e-12, 1e-25, 1e10, 1e17])
    p_frame = [[0.9, 0.03, 0.03, 0.02, 0.01, 0.01], [0.5, 0.1, 0.2, 0.1, 0.05, 0.05], [0.98, 0.02, 0, 0, 0, 0]][seed % 3]
    p_row = [[1, 0, 0, 0, 0, 0], [0.9, 0.05, 0.05, 0, 0, 0], [0.7, 0.05, 0.1, 0.05, 0.05, 0.05]][(seed // 3) % 3]
    templates = [(rng.standard_normal((L, K)) * 3).astype(np.float64) * scales[rng.choice(6, size=(L, 1), p=p_row)] for _ in range(T)]
    templates = [t.astype(np.float32) for t in templates]
    mf = ((rng.standard_normal((S, n_win + L - 1, K)) * 3).astype(np.float64) * scales[rng.choice(6, size=(S, n_win + L - 1, 1), p=p_frame)]).astype(np.float32)
    if seed % 4 == 0:   # a run of identical frames: windows inside it are exactly zero after the mean is taken out
        mf[0, 10:10 + L + 5] = mf[0, 10]
    scores, _, agg = ctx.dtw_scores(mf, ra.Templates(ctx, templates), score_mode=ra.ScoreMode.Average)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates, mode="average")
        ok = np.isfinite(ref_s)
        assert np.array_equal(np.isfinite(scores[s]), ok)
        assert rel_err(scores[s][ok], ref_s[ok]) <= 1e-5, (seed, rel_err(scores[s][ok], ref_s[ok]))
        oka = np.isfinite(ref_a)
        assert rel_err(agg[s][oka], ref_a[oka]) <= 2e-5


def _scaled_rpw(scale):
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    sf = {k: (np.asarray(v, np.float64) * scale).astype(np.float32) for k, v in w["samples_features"].items()}
    av = (np.asarray(w["avg_features"], np.float64) * scale).astype(np.float32)
    return dict(w, samples_features=sf, avg_features=av)


@pytest.mark.parametrize("avg_threshold", [0.0, 0.2])
def test_detectors_with_a_template_set_outside_the_norm_range(ra, ctx, avg_threshold, tmp_path):
    """A .rpw whose rows are tiny (caller input: nothing in the format forbids it): every entry point that takes a wakeword
    reference -- rp_batch_detect (plain and gated), the live-stream batch, the single-stream Rustpotter handle -- scores it
    with the reference-shaped cell and agrees with the oracle's chunked detector."""
    w = _scaled_rpw(3e-23)   # squares of ~1e-44: a few bits of a subnormal -- the oracle's scores move in the third digit
    names = list(w["samples_features"].keys())
    templates = [w["samples_features"][n] for n in names]
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 9)])
    cfg = ra.RustpotterConfig.default()
    cfg.detector.avg_threshold = avg_threshold
    cfg.detector.threshold = 0.35   # scores move when the norms' product underflows: keep some detections alive
    cfg.detector.min_scores = 3
    cfg.fmt.sample_format = ra.SampleFormat.I16
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    before = ctx.dtw_ref_pairs()
    det, n_det, scores, agg = ctx.batch_detect(pcm, tm, cfg.detector, want_scores=True)
    assert ctx.dtw_ref_pairs() > before
    mf = ctx.mfcc(pcm, 5)
    for s in range(2):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_err(scores[s], ref_s) <= 1e-5 and rel_err(agg[s], ref_a) <= 1e-5
    # detect-only call (the averaged-template gate runs as a skip when avg_threshold != 0), live streams, per-stream handles
    det2, n_det2 = ctx.batch_detect(pcm, tm, cfg.detector)
    assert np.array_equal(n_det, n_det2)
    sb = ra.StreamBatch(ctx, tm, cfg.detector, 2, max_chunks_per_call=3)
    live = [[] for _ in range(2)]
    for i in range(0, n, 480 * 3):
        d, nd, _ = sb.process(pcm[:, i:i + 480 * 3], want_agg=True)
        for s in range(2):
            live[s] += [(int(d[s][j]["frame"]), int(d[s][j]["counter"]), float(d[s][j]["score"])) for j in range(nd[s])]
    rpw_bytes = rpw_py.dump_rpw_ref(w["name"], w["samples_features"], w["avg_features"], w.get("threshold"), w.get("avg_threshold"),
                                    w.get("rms_level", 0.0), 5)
    for s in range(2):
        o = orc.Detector(avg_threshold=avg_threshold, threshold=0.35, min_scores=3)
        o.add_ref(w)
        rp = ra.Rustpotter.new(cfg)
        rp.add_wakeword_from_buffer("w", rpw_bytes)
        ref, got = [], []
        for i in range(0, n, 480):
            r = o.process_i16(pcm[s, i:i + 480])
            if r is not None:
                ref.append((i // 480, r))
            g = rp.process_samples(pcm[s, i:i + 480].copy())
            if g is not None:
                got.append((i // 480, g))
        assert len(ref) == n_det[s] == n_det2[s] == len(got) == len(live[s])
        for j, (chunk, r) in enumerate(ref):
            for d in (det[s][j], det2[s][j]):
                assert d["frame"] // 3 + 1 == chunk and d["counter"] == r["counter"]
                assert abs(d["score"] - r["score"]) <= 1e-5 * r["score"] and abs(d["avg_score"] - r["avg_score"]) <= 1e-5 * max(r["avg_score"], 1e-30)
            assert got[j][0] == chunk and got[j][1].counter == r["counter"] and abs(got[j][1].score - r["score"]) <= 1e-5 * r["score"]
            assert live[s][j][0] == det[s][j]["frame"] and live[s][j][1] == r["counter"] and abs(live[s][j][2] - r["score"]) <= 1e-5 * r["score"]
    assert sum(n_det) >= 2, "the case must keep detections to compare"
    assert abs(det[0][0]["score"] - 0.7310586) > 1e-3, "golden score of the unscaled file (tests/detector.rs:24-40): the scaled rows must move it"
