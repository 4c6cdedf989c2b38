"""dtw_ragged_kernel (rustpotter_amd/csrc/rp_dtw_ragged.hip): the banded DTW on the matrix cores for templates of UNEQUAL length --
the shape of every wakeword file the reference ships (tests/wakeword.rs:27-71) and of BASELINE config C1.  Each window is cut to
every template's own length before the column means are taken (wakeword_comp.rs:22-27,99-104).  The kernel is OPT-IN
(RP_DTW_RAGGED=1: measured 0-8 % faster than the register kernels, DESIGN.md 4.2b, which does not pay for the bit-equality with the
live path it would cost); these tests hold it to the same bar as every default path: against the oracle (1e-5, the gate of every
DTW test), against the register kernels (same scores to 4e-6 and not the same bits -- i.e. the kernel really runs;
rp_ctx_dtw_kernels names it), the reference's own fixtures, and its fallbacks (imprecise windows, norm range, score_ref floor)."""
import os

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

SEED = 0x5EED000000000001
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


def rel_close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-30)))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def _streams(S, n_frames, K=5, first=0):
    n = 480 * (n_frames // 3 + 2)
    mf = [orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(S)]
    assert all(m.shape[0] == n_frames for m in mf)
    return np.stack(mf)


def _ragged_on():
    """RP_ARITH_FAST_SPLIT + ragged_matrix (rp_ctx_set_arithmetic on every live context) for the calls inside: the opt-in kernel is admitted."""
    import rustpotter_amd
    return rustpotter_amd.arithmetic_all("fast_split", ragged_matrix=True)


def _ragged_templates(seed, lens, K=5):
    tt = orc.synth_templates(seed, len(lens), max(lens), K)
    return [np.ascontiguousarray(t[:L]) for t, L in zip(tt, lens)]


def _check(ra, ctx, templates, mf, band=5, score_ref=0.22, mode=None, expect_ragged=True, tol=1e-5):
    tm = ra.Templates(ctx, templates)
    kw = {} if mode is None else {"score_mode": mode[0]}
    ctx.dtw_kernels()
    with _ragged_on():
        scores, _, agg = ctx.dtw_scores(mf, tm, band_size=band, score_ref=score_ref, **kw)
    ran = ctx.dtw_kernels()
    assert ("dtw_ragged_kernel" in ran) == expect_ragged, ran
    worst = 0.0
    for s in range(mf.shape[0]):
        ref_s, ref_a = orc.score_stream(mf[s], templates, band=band, score_ref=score_ref, **({} if mode is None else {"mode": mode[1]}))
        worst = max(worst, rel_err(scores[s], ref_s))
        assert rel_close(scores[s], ref_s, tol), rel_err(scores[s], ref_s)
        assert rel_close(agg[s], ref_a, tol), rel_err(agg[s], ref_a)
    reg, _, _ = ctx.dtw_scores(mf, tm, band_size=band, score_ref=score_ref, **kw)   # the default path
    assert "dtw_ragged_kernel" not in ctx.dtw_kernels()
    assert rel_close(scores, reg, 4e-6), rel_err(scores, reg)
    assert np.array_equal(scores, reg) == (not expect_ragged)
    return worst


@pytest.mark.parametrize("lens", [(108, 96, 90, 93, 102), (117, 126, 99), (16,), (17, 16), (31, 32, 33, 47, 48, 49, 64, 65), (100, 100, 40),
                                  (20, 21, 22, 23, 24, 25, 26, 27, 28), (168, 144, 150)])
def test_scores_match_the_oracle(ra, ctx, lens):
    """The reference's own length sets (oye_casa_g 108/96/90/93/102, alexa 117/126/99, oye_casa_real 144..168), lengths around the
    16-column blocks, one and two templates, nine (two chunks), a pair of equal lengths beside a single; 70 windows per stream: the
    512-window tiles straddle the streams."""
    K, S = 5, 9
    n_win = 70
    templates = _ragged_templates(SEED + sum(lens), lens, K)
    mf = _streams(S, n_win + max(lens) - 1, K, first=max(lens))
    _check(ra, ctx, templates, mf)


@pytest.mark.parametrize("band", [3, 4])
def test_bands_three_and_four(ra, ctx, band):
    templates = _ragged_templates(SEED + band, (40, 57, 33, 64), 5)
    mf = _streams(4, 64 + 80, 5, first=300)
    _check(ra, ctx, templates, mf, band=band)
    six = ra.Templates(ctx, templates)
    ctx.dtw_kernels()
    with _ragged_on():
        ctx.dtw_scores(mf, six, band_size=6)
    assert "dtw_ragged_kernel" not in ctx.dtw_kernels()   # band 6 needs 14 row slots + the look-ahead: the register kernels keep it


def test_mixed_with_equal_length_chunks(ra, ctx):
    """8 templates of one length (dtw_mfma_kernel) + three ragged ones (dtw_ragged_kernel) in one call; Median over all of them."""
    K = 5
    templates = orc.synth_templates(SEED + 77, 11, 64, K)
    templates[8] = templates[8][:50].copy()
    templates[9] = templates[9][:41].copy()
    templates[10] = templates[10][:33].copy()
    mf = _streams(3, 64 + 90, K, first=40)
    tm = ra.Templates(ctx, templates)
    ctx.dtw_kernels()
    with _ragged_on():
        scores, _, agg = ctx.dtw_scores(mf, tm, score_mode=ra.ScoreMode.Median)
    ran = ctx.dtw_kernels()
    assert "dtw_ragged_kernel" in ran and "dtw_mfma_kernel" in ran, ran
    for s in range(3):
        ref_s, ref_a = orc.score_stream(mf[s], templates, mode="median")
        assert rel_close(scores[s], ref_s), rel_err(scores[s], ref_s)
        assert rel_close(agg[s], ref_a)


@pytest.mark.parametrize("score_ref", [0.1, 0.15, 0.22])
def test_score_ref_floor(ra, ctx, score_ref):
    templates = _ragged_templates(SEED + 9, (60, 72, 66), 5)
    mf = _streams(4, 64 + 71, 5, first=500)
    _check(ra, ctx, templates, mf, score_ref=score_ref)


def test_below_the_score_ref_floor_the_register_kernels_score(ra, ctx):
    templates = _ragged_templates(SEED + 9, (60, 72, 66), 5)
    mf = _streams(2, 64 + 71, 5, first=500)
    _check(ra, ctx, templates, mf, score_ref=0.05, expect_ragged=False)


# ---- the reference's own files and recordings, whole path (rp_batch_detect), with the matrix-core form switched on
import json          # noqa: E402

import rpw_py        # noqa: E402
import simstream     # noqa: E402

EXP = json.load(open(os.path.join(GOLDEN, "expectations.json")))


def _detector_config(ra, e, **over):
    cfg = ra.DetectorConfig()
    cfg.threshold = e.get("threshold", 0.5)
    cfg.avg_threshold = over.get("avg_threshold", 0.0)   # the gate off: every window is compared with every template
    cfg.score_mode = {"max": 1, "median": 2, "average": 0}[e.get("score_mode", "max")]
    cfg.min_scores = e.get("min_scores", 5)
    return cfg


@pytest.mark.parametrize("case", ["max", "median", "average"])
def test_reference_fixture_stream(ra, ctx, case):
    """tests/detector.rs:24-87: the recording the reference's tests build (simstream) against oye_casa_g.rpw -- five templates of 108 / 96 /
    90 / 93 / 102 frames: the detections of the opt-in path carry the scores the reference asserts (1e-5) and the same frames and counters as
    the default path; its per-window scores are within 4e-6 of the register kernels' and not the same bits."""
    e = EXP["simulation"][case]
    w = rpw_py.load_rpw(os.path.join(GOLDEN, e["rpw"]))
    templates = list(w["samples_features"].values())
    assert len({t.shape[0] for t in templates}) == len(templates)   # all lengths differ
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 7)])
    tm = ra.Templates(ctx, templates)
    cfg = _detector_config(ra, e)
    det0, n0, sc0, agg0 = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    ctx.dtw_kernels()
    with _ragged_on():
        det1, n1, sc1, agg1 = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    assert "dtw_ragged_kernel" in ctx.dtw_kernels()
    assert rel_close(sc1, sc0, 4e-6) and not np.array_equal(sc1, sc0)
    assert np.array_equal(n0, n1) and n1[0] == len(e["detections"])
    for s in range(2):
        for j in range(n1[s]):
            assert (det1[s][j]["frame"], det1[s][j]["window"], det1[s][j]["counter"]) == (det0[s][j]["frame"], det0[s][j]["window"], det0[s][j]["counter"])
    for j, (_, gscore) in enumerate(e["detections"]):
        assert abs(det1[0][j]["score"] - np.float32(gscore)) <= 1e-5 * gscore      # the value tests/detector.rs asserts


def test_windows_the_kernel_cannot_resolve_are_scored_again(ra, ctx):
    """Digital silence behind speech: the windows inside the silence centre to rounding residue, 1e-6 of the stream's offset -- far below
    what the f16 parts of the shared operand resolve.  The kernel lists them and the scale-invariant register kernels score them again (list
    mode behind rp_batch_detect, whose frame array ends with slack; dtw_ref_kernel behind rp_dtw_score_batch): every score at 1e-5 of the
    oracle, as if the kernel had never run on them."""
    lens = (40, 57, 33, 64)
    templates = _ragged_templates(SEED + 21, lens, 5)
    S, N = 3, 480 * 120
    pcm = np.stack([orc.synth_pcm(SEED, 70 + s, N) for s in range(S)])
    pcm[:, N // 2:] = 0.0                                  # the second half: digital silence
    pcm[2] = 0.0                                           # and a stream of nothing else
    tm = ra.Templates(ctx, templates)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    before = ctx.dtw_ref_pairs()
    ctx.dtw_kernels()
    with _ragged_on():
        _, _, sc, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    ran = ctx.dtw_kernels()
    assert "dtw_ragged_kernel" in ran and "register kernels" in ran, ran   # the list pass
    assert ctx.dtw_ref_pairs() == before                                   # ... not the reference-shaped kernel
    mf = ctx.mfcc(pcm, 5)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_close(sc[s], ref_s), rel_err(sc[s], ref_s)
        assert rel_close(agg[s], ref_a)
    # the operator-level call has no slack behind the caller's array: the same windows go to dtw_ref_kernel
    with _ragged_on():
        sc2, _, _ = ctx.dtw_scores(mf, tm)
    assert ctx.dtw_ref_pairs() > before
    for s in range(S):
        ref_s, _ = orc.score_stream(mf[s], templates)
        assert rel_close(sc2[s], ref_s), rel_err(sc2[s], ref_s)


def test_two_ragged_chunks_may_both_list_every_window(ra, ctx):
    """More than eight templates of unequal length = two ragged chunks, each listing on its own: digital silence makes BOTH list (nearly)
    every window, so the list holds up to two entries per window -- the list-mode launches must cover all of them (round-5 advice: a grid
    sized for one entry per window dropped the tail, and a dropped window kept the matrix kernel's unresolved score)."""
    lens = (40, 57, 33, 64, 45, 51, 38, 60, 47, 55)
    templates = _ragged_templates(SEED + 29, lens, 5)
    S, N = 4, 480 * 120
    pcm = np.stack([orc.synth_pcm(SEED, 170 + s, N) for s in range(S)])
    pcm[:, 480 * 12:] = 0.0                                # nine tenths of every stream: digital silence
    pcm[3] = 0.0
    tm = ra.Templates(ctx, templates)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.dtw_kernels()
    with _ragged_on():
        _, _, sc, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    ran = ctx.dtw_kernels()
    assert "dtw_ragged_kernel" in ran and "register kernels" in ran, ran
    mf = ctx.mfcc(pcm, 5)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_close(sc[s], ref_s), rel_err(sc[s], ref_s)
        assert rel_close(agg[s], ref_a)


def test_a_streams_bits_do_not_depend_on_the_batch(ra, ctx):
    """Offset and scale of the shared operand are functions of the stream alone (ragged_prep_kernel): a stream scored alone, first or last in
    a batch gets the same bits -- the 512-window tiles fall differently each time."""
    templates = _ragged_templates(SEED + 33, (70, 81, 64), 5)
    mf = _streams(5, 64 + 100, 5, first=900)
    tm = ra.Templates(ctx, templates)
    with _ragged_on():
        ctx.dtw_kernels()
        whole, _, _ = ctx.dtw_scores(mf, tm)
        assert "dtw_ragged_kernel" in ctx.dtw_kernels()
        alone, _, _ = ctx.dtw_scores(mf[3:4], tm)
        tail, _, _ = ctx.dtw_scores(mf[2:], tm)
    assert np.array_equal(whole[3], alone[0]) and np.array_equal(whole[2:], tail)


def test_detect_only_call_abandons_without_changing_a_detection(ra, ctx):
    """Early abandon in the opt-in kernel: a detect-only call (no per-window arrays) may stop template passes that can no longer reach the
    threshold; the detections are those of the full call."""
    e = EXP["simulation"]["max"]
    w = rpw_py.load_rpw(os.path.join(GOLDEN, e["rpw"]))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 5), np.roll(base[:n], -480 * 9)])
    tm = ra.Templates(ctx, templates)
    cfg = _detector_config(ra, e)
    with _ragged_on():
        det_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
        ctx.dtw_kernels()
        det_only, n_only = ctx.batch_detect(pcm, tm, cfg)[:2]
        assert "dtw_ragged_kernel" in ctx.dtw_kernels()
    assert np.array_equal(n_full, n_only) and n_full.sum() >= 3
    for s in range(3):
        for j in range(n_full[s]):
            assert all(det_full[s][j][k] == det_only[s][j][k] for k in ("frame", "window", "counter", "score"))


def test_randomised_ragged_sweep(ra, ctx):
    """16 random (reference of unequal lengths, config, streams) cases against the oracle's chunked detector, the opt-in kernel scoring the
    offline calls (tests/sweep_parity.py --ragged-cases; the long run is recorded in profiles/sweep_r05.txt)."""
    import sweep_parity
    n, total, ties = sweep_parity.run_sweep(ra, ctx, 16, seed=7, ragged=True)
    assert n == 16 and total >= 5 and ties == 0
