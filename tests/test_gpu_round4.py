"""GPU tests added in round 4 (all through the C ABI): the arithmetic domain of the fast formulations.

* cosine distance (src/mfcc/comparator.rs:28-48): the reference divides by sqrt(dot_a * dot_b) in f32 and answers
  similarity 0 when that product underflows to 0; the scale-invariant device kernels hand the (window, templates) pairs
  whose squared norms leave 2^-60 .. 2^30 / 2^60 to dtw_ref_kernel (rp_dtw.hip), which forms the cell as the reference
  does.  Scores against the oracle at 1e-5 for tiny / huge windows and templates, every kernel family.
* score_ref well below the default 0.22 on the matrix-core DTW shapes (the relative score error grows like 1 / score_ref).
* the wakeword-model forward for features beyond the f16 range (src/wakewords/nn/wakeword_nn.rs:101-106: candle f32)."""
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


# ------------------------------------------------------------------------------------------------ score_ref
def _registers_only():
    """RP_ARITH_STRICT_F32 (rp_ctx_set_arithmetic on every live context) for the calls inside: the f32 vector kernels only."""
    import rustpotter_amd
    return rustpotter_amd.arithmetic_all("strict_f32")


@pytest.mark.parametrize("K,T,L,band", [(5, 8, 100, 5), (5, 4, 100, 5), (5, 6, 60, 3), (5, 7, 37, 4)])
@pytest.mark.parametrize("score_ref", [0.22, 0.15, 0.1, 0.05])
@pytest.mark.parametrize("arith", ["f32_matrix", "fast_split"])
def test_matrix_core_shapes_at_low_score_ref(ra, ctx, K, T, L, band, score_ref, arith):
    """DetectorConfig.score_ref (config.rs:172-209) scales the exponent of the score: the relative error of a score is
    (1 - score) x d(cost / (m + n)) / score_ref, so a kernel whose cost error is fine at the default 0.22 can miss the 1e-5 gate at
    0.05.  The matrix-core shapes (f16-split cosine products) against the oracle at the CONTRACT's tolerance, not the sweeps' 1e-3:
    48 streams x 150 windows per case."""
    S, n_win = 48, 150
    templates = orc.synth_templates(SEED + 7 * L + T, T, L, K)
    mf = _streams(S, n_win + L - 1, K, first=1000 + 50 * T)
    tm = ra.Templates(ctx, templates)
    with ctx.arithmetic(arith):
        scores, _, agg = ctx.dtw_scores(mf, tm, score_ref=score_ref, band_size=band)
    worst = 0.0
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates, band=band, score_ref=score_ref)
        worst = max(worst, rel_err(scores[s], ref_s), rel_err(agg[s], ref_a))
    assert worst <= 1e-5, worst
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm, score_ref=score_ref, band_size=band)
    # (the four-slot shape for chunks of 3..4 templates exists in the two-part f16 arithmetic only)
    matrix = (T >= 5 and band <= 5) or (T >= 3 and band == 5 and arith == "fast_split")
    assert np.array_equal(scores, reg) == (not matrix), "the matrix-core kernel serves these shapes down to score_ref 0.05"
    assert rel_err(scores, reg) <= 4e-6 * 0.22 / score_ref


def test_below_the_score_ref_floor_the_register_kernels_score(ra, ctx):
    """RP_ARITH_FAST_SPLIT, kDtwMfmaMinScoreRef = 0.05 (rp_kernels.h): below it dtw_mfma_supported refuses the two-part form and the f32
    register kernels serve the same chunks -- the same bits as RP_ARITH_STRICT_F32 -- and stay within the gate down to where f32 itself can
    (0.03 here).  The default three-part form has f32-grade products and no floor: it serves the chunks at 0.03 too, within the gate."""
    K, T, L = 5, 8, 100
    templates = orc.synth_templates(SEED + 1234, T, L, K)
    mf = _streams(16, 150 + L - 1, K, first=4000)
    tm = ra.Templates(ctx, templates)
    for score_ref in (0.049, 0.03):
        with ctx.arithmetic("fast_split"):
            scores, _, _ = ctx.dtw_scores(mf, tm, score_ref=score_ref)
        ctx.dtw_kernels()
        with ctx.arithmetic("f32_matrix"):
            three, _, _ = ctx.dtw_scores(mf, tm, score_ref=score_ref)
        assert "dtw_mfma_kernel" in ctx.dtw_kernels() and ctx.last_dtw_products == ["bf16x3"]
        with _registers_only():
            reg, _, _ = ctx.dtw_scores(mf, tm, score_ref=score_ref)
        assert np.array_equal(scores, reg) and not np.array_equal(three, reg)
        for s in range(16):
            ref_s, _ = orc.score_stream(mf[s], templates, score_ref=score_ref)
            assert rel_err(scores[s], ref_s) <= 1e-5
            assert rel_err(three[s], ref_s) <= 1e-5
    with ctx.arithmetic("fast_split"):
        at_floor, _, _ = ctx.dtw_scores(mf, tm, score_ref=0.05)
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm, score_ref=0.05)
    assert not np.array_equal(at_floor, reg)


# ------------------------------------------------------------------------------------------------ wakeword-model forward
@pytest.mark.parametrize("dims", [(3120, 32, 16, 2), (1040, 13, 2), (3120, 80, 40, 3), (64, 13, 2)])
@pytest.mark.parametrize("prec", ["f32", "f32_fast"])
def test_model_forward_is_finite_for_every_finite_input(ra, ctx, dims, prec):
    """candle's f32 Linear (wakeword_nn.rs:101-106) has the f32 range.  RP_MLP_F32 multiplies three-part bf16 splits (the f32 exponent range:
    nothing special happens); RP_MLP_F32_FAST multiplies two-part f16 splits: rows with a feature beyond the f16 range (65 504 .. 1e30) are
    computed again by the f32 matrix instructions.  Either way NaN-free: logits equal to the oracle's at f32 distance, the other rows keep
    their bits, and the result does not depend on which rows of the batch are out of range."""
    rng = np.random.default_rng(sum(dims))
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
    model = ra.Model(ctx, ws, bs)
    for B in (3, 130, 1031):
        x = rng.standard_normal((B, dims[0])).astype(np.float32)
        plain = ctx.mlp_forward(x, model, precision=prec)
        xb = x.copy()
        big_rows = sorted(set(int(r) for r in rng.integers(0, B, size=max(1, B // 9))))
        for j, r in enumerate(big_rows):
            mag = (65505.0, 7.0e4, 1e9, 1e20, 1e30, 3e38 / dims[0])[j % 6]
            xb[r, rng.integers(0, dims[0], size=1 + j % 3)] = np.float32(mag) * (1 if j % 2 else -1)
        got = ctx.mlp_forward(xb, model, precision=prec)
        ref = orc.mlp_forward(xb, ws, bs)
        assert np.isfinite(ref).all() and np.isfinite(got).all()
        # (two f32 summation orders differ relative to the terms they add, which grow with the largest feature of the row)
        tol = 2e-5 * np.maximum(1.0, np.abs(xb).max(axis=1, keepdims=True).astype(np.float64)) + 2e-5 * np.abs(ref)
        assert np.all(np.abs(got.astype(np.float64) - ref) <= tol), np.abs(got - ref).max()
        strict = ctx.mlp_forward(xb, model, precision="f32_strict")
        assert np.all(np.abs(strict.astype(np.float64) - ref) <= tol)
        keep = np.setdiff1d(np.arange(B), big_rows)
        assert got[keep].tobytes() == plain[keep].tobytes()
        if prec == "f32_fast":
            assert got[big_rows].tobytes() == strict[big_rows].tobytes()      # the f32 matrix instructions computed them
        assert ctx.mlp_forward(xb, model, precision=prec).tobytes() == got.tobytes()          # the list is empty again after every call
        assert ctx.mlp_forward(x, model, precision=prec).tobytes() == plain.tobytes()
    # every row out of range, and inf / NaN features behave like the reference's arithmetic (propagate)
    x = (rng.standard_normal((70, dims[0])) * 1e6).astype(np.float32)
    got, ref = ctx.mlp_forward(x, model, precision=prec), orc.mlp_forward(x, ws, bs)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    x = rng.standard_normal((40, dims[0])).astype(np.float32)
    x[3, 5] = np.inf
    x[9, 11] = np.nan
    got, ref = ctx.mlp_forward(x, model, precision=prec), orc.mlp_forward(x, ws, bs)
    ok = np.isfinite(ref)
    ok[3] = False
    if prec == "f32_fast":   # the f32 matrix instructions take the rows with non-finite features: inf and NaN propagate like the reference's f32
        assert np.array_equal(np.isnan(got), np.isnan(ref))
    else:                    # the three-part split of +-inf is inf + NaN: such a row's logits are NaN (documented, include/rustpotter_hip.h); NaN propagates as NaN
        assert np.isnan(got[3]).all() and np.isnan(got[9]).all() and np.array_equal(np.isnan(np.delete(got, 3, axis=0)), np.isnan(np.delete(ref, 3, axis=0)))
    assert np.allclose(got[ok], ref[ok], rtol=2e-5, atol=2e-5)
    assert ("bf16x3" if prec == "f32" else "f32") in ctx.last_mlp_kernel()


def test_model_detector_with_out_of_range_features(ra, ctx):
    """The batched model detector reads its windows in place (mlp_mfma_kernel, window mean folded in after layer 1): windows that
    contain an MFCC frame beyond the f16 range are computed again by the f32 matrix instructions -- through rp_mlp_forward_batch's
    sibling entry rp_batch_detect_model nothing can be injected (the frames come from the MFCC kernel), so the window form is
    driven through the live / offline equality on ordinary audio and the dense form above carries the range test."""
    rng = np.random.default_rng(8)
    K, L = 16, 20
    dims = (K * L, 32, 16, 2)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    model = ra.Model(ctx, ws, bs)
    pcm = np.stack([orc.synth_pcm(SEED, s, 480 * 40) for s in range(3)])
    cfg = ra.DetectorConfig()
    cfg.threshold = 0.0
    cfg.min_scores = 1
    a = ctx.batch_detect_model(pcm, model, K, 1, cfg)
    b = ctx.batch_detect_model(pcm, model, K, 1, cfg, precision="f32_strict")
    assert np.array_equal(a[2], b[2]) and a[2].sum() >= 3     # same detections per stream from both forms
    for s in range(3):
        for j in range(a[2][s]):
            assert a[0][s][j]["frame"] == b[0][s][j]["frame"] and abs(a[0][s][j]["score"] - b[0][s][j]["score"]) <= 1e-5 * abs(b[0][s][j]["score"])


# ------------------------------------------------------------------------------------------------ host ingest
@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_batch_detect_ingest_equals_one_resident_call(ra, ctx, dtype):
    """rp_batch_detect_ingest (streams in HOST memory, taken in blocks with the next block's copy under this block's kernels, results
    copied back per block) == rp_batch_detect over all streams at once: n_det and every detection record, global stream ids included --
    with a ragged last block, one block only, more blocks than streams; pageable and page-locked host memory; on a device-pointer
    context too (the entry point takes host arrays whatever the context's flag says)."""
    import torch
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    rng = np.random.default_rng(3)
    streams = [np.roll(base[:n], 480 * int(rng.integers(0, 40))) for _ in range(21)]
    pcm = np.stack(streams)
    if dtype is np.float32:
        pcm = simstream.i16_to_f32(pcm)
    cfg = ra.DetectorConfig()
    for gate in (0.0, 0.2):
        cfg.avg_threshold = gate
        tm = ra.Templates(ctx, templates, avg=w["avg_features"])
        det1, n1 = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        assert n1.sum() >= 21
        for block in (8, 21, 64, 1):
            det, n_det, sec = ctx.batch_detect_ingest(pcm, tm, cfg, max_det=4, block_streams=block)
            assert np.array_equal(n_det, n1) and det.tobytes() == det1.tobytes() and sec > 0
        assert [int(d["stream"]) for s in range(21) for d in det[s][:n_det[s]]] == [s for s in range(21) for _ in range(n_det[s])]
    # page-locked memory, device-pointer context
    dctx = ra.BatchContext(device=0, host_pointers=False)
    tmd = ra.Templates(dctx, templates, avg=w["avg_features"])
    hp = torch.from_numpy(pcm).pin_memory()
    det_h = torch.zeros((21, 4, 6), dtype=torch.int32).pin_memory()
    n_h = torch.zeros((21,), dtype=torch.int32).pin_memory()
    dctx.batch_detect_ingest_ptr(hp.data_ptr(), 3 if dtype is np.float32 else 1, 21, n, n, tmd, cfg, det_h.data_ptr(), n_h.data_ptr(), 4, block_streams=5)
    assert np.array_equal(n_h.numpy(), n1) and det_h.numpy().tobytes() == det1.view(np.int32).reshape(21, 4, 6).tobytes()
    # nothing to do / bad arguments
    det0, n0, _ = ctx.batch_detect_ingest(pcm[:0], tm, cfg, max_det=4)
    assert det0.shape[0] == 0


# ------------------------------------------------------------------------------------------------ per-call device state
def test_per_call_words_survive_interleaved_calls(ra, ctx):
    """The context keeps a few words of device state that every call must leave at zero for the next one: the matrix-core kernels' tile
    counters, the list of out-of-range pairs (+ its ALL-mode overflow), the per-stream can-fire flags (cleared by the scan that reads
    them), the model forward's redo list.  200 calls of different kinds in a scrambled order on ONE context: each kind gives the same
    bits every time, whatever ran before it."""
    rng = np.random.default_rng(21)
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 5), np.roll(base[:n], 480 * 17)])
    cfg = ra.DetectorConfig()
    cfg_gate0 = ra.DetectorConfig()
    cfg_gate0.avg_threshold = 0.0
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    same = orc.synth_templates(SEED + 77, 8, 60, 5)            # one chunk for the matrix-core kernel (folded Max, tile counters)
    tm8 = ra.Templates(ctx, same)
    tiny = [(np.asarray(t, np.float64) * 3e-23).astype(np.float32) for t in templates]
    tmt = ra.Templates(ctx, tiny, avg=(np.asarray(w["avg_features"], np.float64) * 3e-23).astype(np.float32))   # ref_only set
    mf = _streams(4, 60 + 90, 5, first=60)
    mf_tiny = (mf.astype(np.float64) * 1e-12).astype(np.float32)
    dims = (320, 32, 16, 2)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    model = ra.Model(ctx, ws, bs)
    x = rng.standard_normal((300, dims[0])).astype(np.float32)
    xb = x.copy()
    xb[::7, 3] = 1e9

    def digest(*arrays):
        return tuple(np.ascontiguousarray(a).tobytes() for a in arrays)

    kinds = {
        "detect": lambda: digest(*ctx.batch_detect(pcm, tm, cfg, max_det=4)),                      # gated, detect-only
        "detect_scores": lambda: digest(*ctx.batch_detect(pcm, tm, cfg_gate0, max_det=4, want_scores=True)),
        "detect_same8": lambda: digest(*ctx.batch_detect(pcm, tm8, cfg_gate0, max_det=4)),         # matrix-core kernel, Max folded in, hot flags
        "detect_ref_only": lambda: digest(*ctx.batch_detect(pcm, tmt, cfg, max_det=4)),            # every window through dtw_ref_kernel
        "scores": lambda: digest(*ctx.dtw_scores(mf, tm8)),
        "scores_tiny": lambda: digest(*ctx.dtw_scores(mf_tiny, tm8)),                              # every pair listed and rescored
        "mlp": lambda: digest(ctx.mlp_forward(x, model)),
        "mlp_big": lambda: digest(ctx.mlp_forward(xb, model)),                                     # rows listed for the f32 pass
        "ingest": lambda: digest(*ctx.batch_detect_ingest(pcm, tm, cfg, max_det=4, block_streams=2)[:2]),
    }
    first = {k: f() for k, f in kinds.items()}
    assert first["detect"] == first["ingest"]
    names = list(kinds)
    for i in range(200):
        k = names[int(rng.integers(len(names)))]
        assert kinds[k]() == first[k], (i, k)


# ---------------------------------------------------------------------------------------------------------------------------
# mlp_windows_kernel: layer 1 over all windows of a stream from one staging of its frames (rp_mlp_forward_windows)
def _window_logits_oracle(mfcc, L, ws, bs):
    S, nf, K = mfcc.shape
    n_win = nf - L + 1
    out = np.empty((S, n_win, ws[-1].shape[0]), np.float32)
    for s in range(S):
        rows = np.empty((n_win, L * K), np.float32)
        for w in range(n_win):
            win = mfcc[s, w:w + L]
            acc = np.zeros(K, np.float32)
            for f in range(L):          # MfccNormalizer::normalize: the column sums in frame order
                acc = acc + win[f]
            rows[w] = (win - acc / np.float32(L)).reshape(-1)
        out[s] = orc.mlp_forward(rows, ws, bs)
    return out


def _window_model(rng, L, K, hidden, labels):
    dims = [L * K] + list(hidden) + [labels]
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    return ws, bs


@pytest.mark.parametrize("L,hidden,labels,nf,S", [
    (195, (32, 16), 2, 399, 3),     # the Small shape of BASELINE C5 on 4 s streams: 205 windows = 7 tiles, the last one ragged
    (195, (13,), 2, 450, 2),        # Tiny: one 16-wide output tile (bias / mean-correction rows past it do not exist)
    (50, (32, 32), 3, 81, 2),       # exactly 32 windows: the smallest call the kernel takes
    (251, (17, 8), 2, 550, 2),      # 300 windows: two workgroups per stream; 251 frames = the longest window, 51 weight groups (the last one ragged)
    (7, (), 30, 100, 1),            # a single layer
    (195, (32, 16), 2, 225, 2),     # 31 windows: stays with mlp_mfma_kernel (rows read in place)
    (195, (65, 32), 2, 399, 2),     # Medium: three 32-output tiles, one wave each (mlp_windows_wide_kernel)
    (195, (130, 32), 3, 430, 2),    # Large: five tiles; 236 windows = two workgroups of four row tiles per stream
    (60, (40, 20), 2, 100, 3),      # two output tiles, 41 windows = the two-row-tile form
    (100, (97,), 4, 200, 1),        # four output tiles, no hidden tail layer
    (30, (160,), 160, 70, 1),       # 160 outputs of a single layer written straight from the tiles
])
def test_window_logits_match_the_oracle(ra, ctx, L, hidden, labels, nf, S):
    """rp_mlp_forward_windows against the oracle's window-by-window forward (normalise, flatten, ModelImpl::forward): features on
    coefficient-dependent offsets ten times their spread, like real MFCCs, drifting along the stream (mlp_windows_kernel stages
    the frames minus its middle window's mean and takes the rest of each window's mean out after layer 1; mlp_mfma_kernel
    subtracts the window's own mean as it loads), gate 1e-5 relative to the larger of the logit and the largest CENTRED feature;
    equal to itself on a second call, independent of the other streams."""
    K = 16
    rng = np.random.default_rng(L * 1000 + nf)
    ws, bs = _window_model(rng, L, K, hidden, labels)
    model = ra.Model(ctx, ws, bs)
    # offsets ten times the spread, and a level that drifts along the stream (window means differ across a workgroup's windows)
    off = rng.uniform(-30.0, 30.0, K).astype(np.float32)
    drift = np.linspace(0.0, 1.0, nf)[None, :, None] * rng.uniform(-6.0, 6.0, (S, 1, K))
    mfcc = (rng.standard_normal((S, nf, K)) * rng.uniform(0.5, 2.0, (S, 1, 1)) + off + drift).astype(np.float32)
    got = ctx.mlp_forward_windows(mfcc, model)
    ref = _window_logits_oracle(mfcc, L, ws, bs).astype(np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all()
    # the gate scales with the CENTRED features (what the model sees), not with the offsets they sit on
    spread = max(np.abs(mfcc[s0, w:w + L] - mfcc[s0, w:w + L].mean(axis=0)).max() for s0 in range(S) for w in range(0, nf - L + 1, 16))
    tol = 1e-5 * np.maximum(np.abs(ref), spread)
    assert (np.abs(got - ref) <= tol).all(), float((np.abs(got - ref) / tol).max())
    assert ctx.mlp_forward_windows(mfcc, model).tobytes() == got.tobytes()
    one = ctx.mlp_forward_windows(mfcc[S - 1:], model)
    assert one.tobytes() == got[S - 1:].tobytes()
    os.environ["RP_MLP_WINDOWS"] = "0"      # and against mlp_mfma_kernel reading the rows in place: two summation orders of the same products
    try:
        old = ctx.mlp_forward_windows(mfcc, model)
    finally:
        os.environ.pop("RP_MLP_WINDOWS")
    assert (np.abs(old.astype(np.float64) - ref) <= tol).all() and (np.abs(old.astype(np.float64) - got) <= 2 * tol).all()


def test_window_logits_with_a_frame_beyond_the_f16_range(ra, ctx):
    """A frame holding 1e6 (an f16 part cannot): exactly the windows that contain it are listed and come from the f32 matrix
    instructions -- finite, right, and every other window keeps the bits it has without that frame in the batch."""
    K, L, nf = 16, 195, 399
    rng = np.random.default_rng(77)
    ws, bs = _window_model(rng, L, K, (32, 16), 2)
    model = ra.Model(ctx, ws, bs)
    mfcc = rng.standard_normal((3, nf, K)).astype(np.float32)
    clean = ctx.mlp_forward_windows(mfcc, model)
    hot = mfcc.copy()
    hot[1, 300, 5] = 1e6
    got = ctx.mlp_forward_windows(hot, model)
    ref = _window_logits_oracle(hot, L, ws, bs).astype(np.float64)
    assert np.isfinite(got).all()
    inside = np.zeros((3, nf - L + 1), bool)
    inside[1, 300 - L + 1:301] = True
    assert got[~inside].tobytes() == clean[~inside].tobytes()
    tol = 2e-5 * np.maximum(np.abs(ref), 1e6)
    assert (np.abs(got - ref)[inside] <= tol[inside]).all() and not np.array_equal(got[inside], clean[inside])
    assert ctx.last_mlp_kernel() != ""


@pytest.mark.parametrize("K,L,hidden,nf", [(8, 120, (32, 16), 300), (12, 70, (20,), 160), (20, 50, (65, 32), 130), (4, 200, (13,), 260)])
def test_window_logits_other_mfcc_sizes(ra, ctx, K, L, hidden, nf):
    """The same for mfcc sizes other than 16 (mlp_mfma_kernel with the rows read in place: the window's own mean leaves each feature as
    it is loaded, its place in the frame carried from k-group to k-group): offsets ten times the spread, gate 1e-5 of the centred spread."""
    rng = np.random.default_rng(K * 100 + L)
    ws, bs = _window_model(rng, L, K, hidden, 2)
    model = ra.Model(ctx, ws, bs)
    S = 2
    off = rng.uniform(-30.0, 30.0, K).astype(np.float32)
    drift = np.linspace(0.0, 1.0, nf)[None, :, None] * rng.uniform(-6.0, 6.0, (S, 1, K))
    mfcc = (rng.standard_normal((S, nf, K)) * rng.uniform(0.5, 2.0, (S, 1, 1)) + off + drift).astype(np.float32)
    got = ctx.mlp_forward_windows(mfcc, model)
    ref = _window_logits_oracle(mfcc, L, ws, bs).astype(np.float64)
    spread = max(np.abs(mfcc[s0, w:w + L] - mfcc[s0, w:w + L].mean(axis=0)).max() for s0 in range(S) for w in range(0, nf - L + 1, 16))
    tol = 1e-5 * np.maximum(np.abs(ref), spread)
    assert got.shape == ref.shape and (np.abs(got - ref) <= tol).all(), float((np.abs(got - ref) / tol).max())
    strict = ctx.mlp_forward_windows(mfcc, model, precision="f32_strict")
    assert (np.abs(strict - ref) <= tol).all()
    assert ctx.mlp_forward_windows(mfcc, model).tobytes() == got.tobytes()


def test_window_logits_edges(ra, ctx):
    """rp_mlp_forward_windows at its edges: a stream shorter than one window gives no rows; exactly one window; a model input that is not
    a whole number of frames of the given mfcc size is refused with the reference's words; device pointers give the same bits as host ones."""
    import torch
    K, L = 16, 40
    rng = np.random.default_rng(3)
    ws, bs = _window_model(rng, L, K, (20, 10), 2)
    model = ra.Model(ctx, ws, bs)
    assert ctx.mlp_forward_windows(np.zeros((2, L - 1, K), np.float32), model).shape == (2, 0, 2)
    one = rng.standard_normal((3, L, K)).astype(np.float32)
    got = ctx.mlp_forward_windows(one, model)
    ref = _window_logits_oracle(one, L, ws, bs)
    assert got.shape == (3, 1, 2) and np.allclose(got, ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(ra.RustpotterError, match="mfcc size"):
        ctx.mlp_forward_windows(np.zeros((1, 100, 7), np.float32), model)
    many = rng.standard_normal((5, 300, K)).astype(np.float32)
    host = ctx.mlp_forward_windows(many, model)
    dctx = ra.BatchContext(device=0, host_pointers=False)
    dmodel = ra.Model(dctx, ws, bs)
    x = torch.from_numpy(many).cuda()
    out = torch.empty((5, 300 - L + 1, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L_ = ra.load_library()
    assert L_.rp_mlp_forward_windows(dctx._h, dmodel._h, x.data_ptr(), 5, 300, K, 0, out.data_ptr()) == 0
    dctx.synchronize()
    assert out.cpu().numpy().tobytes() == host.tobytes()
