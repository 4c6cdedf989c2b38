"""dtw_mfma_kernel (rustpotter_amd/csrc/rp_dtw_mfma.hip): the banded DTW whose cosine costs come out of the matrix cores, taken for
mfcc_size 5 chunks of 3..8 same-length templates (5..8: band 3..5, eight template slots per wave; 3..4: band 5, four slots).  Against the oracle (1e-5, the gate of every DTW test), against the
register kernels it replaces (RP_ARITH_STRICT_F32: same scores to 2e-6, and not the same bits -- i.e. the kernel really runs), and in
its other modes: tiles that straddle streams, frames read from global memory (live-stream batches, the gate's list), early abandon.
Every test of this file runs in BOTH matrix arithmetics (the `ctx` fixture): RP_ARITH_F32_MATRIX (the default: three bf16 parts per
operand, f32-grade) and RP_ARITH_FAST_SPLIT (two f16 parts, 22-bit, opt-in); the mfcc_size 13 / 16 kernel exists for the latter only."""
import os

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module", params=["f32_matrix", "fast_split"])
def ctx(ra, request):
    return ra.BatchContext(device=0, host_pointers=True, arithmetic=request.param)


def _fast(ctx):
    return ctx.get_arithmetic()[0] == "fast_split"


def rel_close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-30)))


def _streams(S, n_frames, K=5, first=0):
    n = 480 * (n_frames // 3 + 2)
    mf = [orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(S)]
    assert all(m.shape[0] == n_frames for m in mf)
    return np.stack(mf)


def _registers_only():
    """RP_ARITH_STRICT_F32 (rp_ctx_set_arithmetic on every live context) for the calls inside: the f32 vector kernels only."""
    import rustpotter_amd
    return rustpotter_amd.arithmetic_all("strict_f32")


@pytest.mark.parametrize("L,T", [(12, 5), (13, 8), (23, 5), (24, 6), (25, 8), (36, 7), (59, 6), (100, 8), (126, 5),
                                 (16, 3), (17, 4), (31, 4), (32, 3), (33, 4), (47, 4), (48, 3), (100, 4), (126, 3), (12, 4), (15, 3), (40, 2)])
def test_scores_match_the_oracle(ra, ctx, L, T):
    """Template lengths around the column blocks (12 columns with eight template slots: 12, 13, 23..25, 36; 16 with four: 16, 17,
    31..33, 47, 48), chunk sizes 3..8, window counts that are not a multiple of the 32-window tile: the flattened tiles straddle the
    three streams.  Chunks of two templates and chunks of 3..4 templates shorter than 16 frames stay with the register kernels (the
    last three cases: the switch changes nothing there)."""
    K, S = 5, 3
    n_win = 45 if L < 60 else 39
    templates = orc.synth_templates(SEED + L, T, L, K)
    mf = _streams(S, n_win + L - 1, K, first=L)
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm)
    assert scores.shape == (S, n_win, T)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm)
    assert rel_close(scores, reg, 2e-6), np.abs(scores / reg - 1).max()
    matrix = (T >= 5 and L >= 12) or (3 <= T <= 4 and L >= 16)
    assert np.array_equal(scores, reg) == (not matrix), "which chunks take the matrix-core kernel"


@pytest.mark.parametrize("band,L,T", [(3, 37, 8), (4, 50, 6), (3, 12, 5), (4, 24, 7)])
def test_bands_three_and_four(ra, ctx, band, L, T):
    """band_size 3 and 4 use the same twelve row slots (2 band + 2 <= 12); band 6 would need a fourth tile and stays with the
    register kernel, as does every band beyond."""
    K, S, n_win = 5, 2, 41
    templates = orc.synth_templates(SEED + 31 * band + L, T, L, K)
    mf = _streams(S, n_win + L - 1, K, first=200 + L)
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm, band_size=band)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(mf[s], templates, band=band)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm, band_size=band)
    assert rel_close(scores, reg, 2e-6) and not np.array_equal(scores, reg)
    six, _, _ = ctx.dtw_scores(mf, tm, band_size=6)
    with _registers_only():
        six_reg, _, _ = ctx.dtw_scores(mf, tm, band_size=6)
    assert np.array_equal(six, six_reg)
    # four template slots per wave exist for band 5 only
    tm4 = ra.Templates(ctx, templates[:4])
    four, _, _ = ctx.dtw_scores(mf, tm4, band_size=band)
    with _registers_only():
        four_reg, _, _ = ctx.dtw_scores(mf, tm4, band_size=band)
    assert np.array_equal(four, four_reg)


def test_mixed_chunk_classes_and_an_averaged_template(ra, ctx):
    """21 templates: 8 + 8 of one length (two chunks for the matrix kernel), 3 of another (tc-4 register kernel), two ragged ones
    and the averaged template (two windows per lane) -- every column against the oracle."""
    K = 5
    templates = orc.synth_templates(SEED + 5, 21, 64, K)
    for i in (16, 17, 18):
        templates[i] = templates[i][:50].copy()
    templates[19] = templates[19][:41].copy()
    templates[20] = templates[20][:33].copy()
    avg = orc.synth_templates(SEED + 6, 1, 64, K)[0]
    mf = _streams(2, 64 + 70, K, first=40)
    tm = ra.Templates(ctx, templates, avg=avg)
    scores, avg_s, agg = ctx.dtw_scores(mf, tm, with_avg=True, score_mode=ra.ScoreMode.Median)
    for s in range(2):
        ref_s, ref_a = orc.score_stream(mf[s], templates, mode="median")
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)
        ref_avg = np.array([orc.score_window(mf[s][w:w + 64], avg) for w in range(0, scores.shape[1], 7)])
        assert rel_close(avg_s[s][::7], ref_avg)


def test_zero_rows_constant_windows_and_silence(ra, ctx):
    """`magnitude == 0 -> similarity 0` (comparator.rs:43-47) on both sides: all-zero template rows, windows whose frames all
    equal their mean (the centred frame is the zero vector) and windows that mix constant and live frames."""
    K, L, T = 5, 40, 8
    templates = orc.synth_templates(SEED + 9, T, L, K)
    templates[2][5:9] = 0.0
    templates[7][:] = 0.0
    templates[0][L - 1] = 0.0
    mf = _streams(2, 130, K, first=60)
    mf[0, 30:90] = mf[0, 30]          # 60 identical frames: windows 30..50 are constant, their neighbours partly so
    mf[1, :] = 0.0                     # digital silence after the extractor
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm)
    for s in range(2):
        ref_s, ref_a = orc.score_stream(mf[s], templates)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)
    # a constant window costs exactly 1 per cell against anything
    ref_const = orc.score_window(mf[0][35:35 + L], templates[1])
    assert rel_close(scores[0, 35, 1], ref_const, 1e-6)
    assert np.all(scores[1] == scores[1, 0])


def test_identical_window_and_template(ra, ctx):
    """A window that IS the template (after centring): costs of the matching cells are 1 - 1 = a few ulps around zero, possibly
    negative -- the score must still match the oracle's."""
    K, L, T = 5, 48, 6
    mf = _streams(2, 48 + 40, K, first=80)
    templates = orc.synth_templates(SEED + 11, T, L, K)
    w = mf[0, 17:17 + L]
    templates[1] = (w - w.mean(axis=0, keepdims=True)).astype(np.float32)
    tm = ra.Templates(ctx, templates)
    scores, _, _ = ctx.dtw_scores(mf, tm)
    for s in range(2):
        ref_s, _ = orc.score_stream(mf[s], templates)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
    assert scores[0, 17, 1] > 0.7


def test_long_templates_take_the_eight_wave_shape(ra, ctx):
    """250-frame templates (two-part form; 170 frames in the three-part form, whose A image is twice as large: beyond ~179 frames it does not
    fit beside eight waves' frame stages and the set keeps the register kernels): the A image and twelve waves' frame stages do not fit
    the CU's LDS together, the launch falls back to eight waves per workgroup."""
    K, L, T = 5, 250 if _fast(ctx) else 170, 5
    templates = orc.synth_templates(SEED + 13, T, L, K)
    mf = _streams(2, L + 33, K, first=90)
    tm = ra.Templates(ctx, templates)
    scores, _, _ = ctx.dtw_scores(mf, tm)
    assert scores.shape == (2, 34, T)
    for s in range(2):
        for w in (0, 1, 16, 31, 32, 33):
            for t in (0, 2, 4):
                assert rel_close(scores[s, w, t], orc.score_window(mf[s][w:w + L], templates[t]))
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm)
    assert rel_close(scores, reg, 2e-6) and not np.array_equal(scores, reg)


def test_templates_too_long_for_the_three_part_image_keep_the_register_kernels(ra):
    """RP_ARITH_F32_MATRIX, 250-frame templates: (250 + 16) x 512 bytes of A image + eight frame stages exceed 160 KB -- the register kernels
    score (the arithmetic switch changes nothing), against the oracle."""
    ctx = ra.BatchContext(device=0, host_pointers=True)
    K, L, T = 5, 250, 5
    templates = orc.synth_templates(SEED + 13, T, L, K)
    mf = _streams(2, L + 33, K, first=90)
    tm = ra.Templates(ctx, templates)
    ctx.dtw_kernels()
    scores, _, _ = ctx.dtw_scores(mf, tm)
    assert ctx.dtw_kernels() == ["register kernels"]
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm)
    assert np.array_equal(scores, reg)
    for w in (0, 16, 33):
        assert rel_close(scores[1, w, 2], orc.score_window(mf[1][w:w + L], templates[2]))


def test_long_templates_in_a_chunk_of_four(ra, ctx):
    """250-frame templates, four of them: twelve waves' frame stages do not fit beside the A image -- the four-slot form falls back to eight
    waves per workgroup in the two-part arithmetic (68 KB of image); the three-part image (136 KB) leaves no room for the stages and the
    chunk stays with the tc-4 register kernel (the switch changes nothing).  Scores against the oracle."""
    K, L, T = 5, 250, 4
    templates = orc.synth_templates(SEED + 23, T, L, K)
    mf = _streams(2, L + 33, K, first=95)
    tm = ra.Templates(ctx, templates)
    scores, _, _ = ctx.dtw_scores(mf, tm)
    with _registers_only():
        reg, _, _ = ctx.dtw_scores(mf, tm)
    assert rel_close(scores, reg, 2e-6) and np.array_equal(scores, reg) == (not _fast(ctx))
    for w in (0, 17, 33):
        assert rel_close(scores[1, w, 3], orc.score_window(mf[1][w:w + L], templates[3]))


@pytest.mark.parametrize("K,T,L,matrix", [(5, 8, 179, True), (5, 8, 180, False),      # eight slots, eight waves: (L + 16) x 512 + 8 stages <= 160 KB
                                         (5, 4, 147, True), (5, 4, 148, True), (5, 4, 179, True), (5, 4, 180, False),   # four slots: twelve waves up to 147 frames, then eight
                                         (16, 4, 180, True), (16, 4, 181, False)])   # dtw_mfma_wide3_kernel: (L + 16) x 832 <= 160 KB
def test_the_longest_templates_the_three_part_images_hold(ra, K, T, L, matrix):
    """RP_ARITH_F32_MATRIX at the edge of the CU's LDS: the longest template each shape's A image (and frame stages) still fits, and one frame
    more -- the set then keeps the register kernels.  Scores against the oracle at a few windows either way."""
    ctx = ra.BatchContext(device=0, host_pointers=True)
    S, N = 2, 480 * (L // 3 + 14)
    templates = orc.synth_templates(SEED + L, T, L, K)
    tm = ra.Templates(ctx, templates)
    pcm = np.stack([orc.synth_pcm(SEED, 900 + s, N) for s in range(S)])
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.dtw_kernels()
    _, _, scores, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    ran = ctx.dtw_kernels()
    assert (ran != ["register kernels"]) == matrix and ctx.last_dtw_products == (["bf16x3"] if matrix else []), (ran, ctx.last_dtw_products)
    n_win = scores.shape[1]
    assert n_win >= 20
    for s in range(S):
        mf = orc.mfcc_stream(pcm[s], K)
        for w in (0, 1, n_win // 2, n_win - 1):
            for t in (0, T - 1):
                assert rel_close(scores[s, w, t], orc.score_window(mf[w:w + L], templates[t]))


def test_many_streams_equal_their_single_stream_scores(ra, ctx):
    """Size-independent property at a size the oracle does not reach: 1 500 streams x 77 windows x 8 templates in one launch (every
    wave takes several tiles, most tiles straddle two streams) give, stream by stream, the bits of that stream scored alone."""
    K, L, T, S = 5, 30, 8, 1500
    templates = orc.synth_templates(SEED + 17, T, L, K)
    base = _streams(12, 106, K, first=120)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 12, S)
    gain = (0.5 + rng.random(S)).astype(np.float32)
    mf = base[pick] * gain[:, None, None]
    tm = ra.Templates(ctx, templates)
    scores, _, agg = ctx.dtw_scores(mf, tm)
    for s in (0, 1, 2, 700, 1498, 1499):
        one, _, one_agg = ctx.dtw_scores(mf[s], tm)
        assert np.array_equal(scores[s], one[0]) and np.array_equal(agg[s], one_agg[0])
    ref_s, _ = orc.score_stream(mf[1499], templates)
    assert rel_close(scores[1499], ref_s)


class _static_rounds:
    """RP_MFMA_STATIC_ROUNDS for the calls inside (read per launch): tiles a wave takes by index before it uses the atomic counter."""
    def __init__(self, n):
        self.n = n
    def __enter__(self):
        self.old = os.environ.get("RP_MFMA_STATIC_ROUNDS")
        os.environ["RP_MFMA_STATIC_ROUNDS"] = str(self.n)
    def __exit__(self, *a):
        if self.old is None:
            del os.environ["RP_MFMA_STATIC_ROUNDS"]
        else:
            os.environ["RP_MFMA_STATIC_ROUNDS"] = self.old


@pytest.mark.parametrize("K,T,S", [(5, 8, 1500), (5, 4, 700), (16, 8, 300), (5, 8, 40)])
def test_tile_hand_out_does_not_change_a_bit(ra, ctx, K, T, S):
    """Every tile is scored exactly once whichever way the waves come by it: all from the atomic counter (0), one / two / seven rounds
    by index (seven is more rounds than the launch has: the counter then hands out nothing), the host's rule (unset) -- same bits, and
    the counter is back at zero after every launch (a second call under the same setting repeats the first)."""
    L = 30
    templates = orc.synth_templates(SEED + 23 + K, T, L, K)
    base = _streams(8, 75 + L - 1, K, first=300)
    rng = np.random.default_rng(11)
    mf = base[rng.integers(0, 8, S)] * (0.5 + rng.random(S)).astype(np.float32)[:, None, None]
    tm = ra.Templates(ctx, templates)
    ref, _, ref_agg = ctx.dtw_scores(mf, tm)
    assert np.isfinite(ref).all() and ref.min() > 0.0
    for n in (0, 1, 2, 7):
        with _static_rounds(n):
            for _ in range(2):
                got, _, agg = ctx.dtw_scores(mf, tm)
                assert np.array_equal(got, ref) and np.array_equal(agg, ref_agg), n
    one, _, _ = ctx.dtw_scores(mf[S - 1], tm)
    assert np.array_equal(ref[S - 1], one[0])


class _aggregate_pass:
    """RP_DTW_NO_FUSED_MAX=1 for the calls inside (read per call): ScoreMode::Max by the aggregate pass, not inside the DTW kernel."""
    def __enter__(self):
        self.old = os.environ.get("RP_DTW_NO_FUSED_MAX")
        os.environ["RP_DTW_NO_FUSED_MAX"] = "1"
    def __exit__(self, *a):
        if self.old is None:
            del os.environ["RP_DTW_NO_FUSED_MAX"]
        else:
            os.environ["RP_DTW_NO_FUSED_MAX"] = self.old


@pytest.mark.parametrize("T,L", [(8, 30), (5, 40), (4, 33), (3, 48), (6, 12)])
def test_max_inside_the_dtw_kernel_equals_the_aggregate_pass(ra, ctx, T, L):
    """One chunk of 3..8 same-length templates, ScoreMode::Max, no averaged template: the matrix-core kernel writes the aggregate and the
    per-stream flags itself (DtwFusedAgg) and the aggregate pass is not launched.  Same detections, scores and aggregates bit for bit as
    with the pass (RP_DTW_NO_FUSED_MAX=1), with the per-window arrays and in detect-only calls, offline and chunk by chunk; the
    aggregate is the largest of a window's scores; the pass's timing slot counts no launch."""
    K, S = 5, 70
    templates = orc.synth_templates(SEED + 31 + T, T, L, K)
    tm = ra.Templates(ctx, templates)
    pcm = np.stack([orc.synth_pcm(SEED, 500 + s, 480 * 40) for s in range(S)])
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    scores0 = ctx.batch_detect(pcm, tm, cfg, want_scores=True)[2]
    cfg.threshold, cfg.min_scores = float(np.quantile(scores0.max(axis=2), 0.98)), 2   # a few streams fire, most stay quiet
    ctx.timing_enable(True)
    ctx.timing_reset()
    det, n_det, scores, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    n_fused = ctx.timing_read(2)[1]
    det_only, n_only = ctx.batch_detect(pcm, tm, cfg)
    with _aggregate_pass():
        ctx.timing_reset()
        det2, n2, scores2, agg2 = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
        n_pass = ctx.timing_read(2)[1]
        det_only2, n_only2 = ctx.batch_detect(pcm, tm, cfg)
    ctx.timing_enable(False)
    assert 0 < int(n_det.sum()) and int((n_det == 0).sum()) > S // 2
    assert np.array_equal(scores, scores2) and np.array_equal(agg, agg2) and np.array_equal(agg, scores.max(axis=2))
    assert np.array_equal(n_det, n2) and np.array_equal(det, det2)
    assert np.array_equal(n_only, n_only2) and np.array_equal(det_only, det_only2) and np.array_equal(n_only, n_det)
    # launches of the aggregate pass inside the two calls
    assert n_fused == 0 and n_pass == 1, (n_fused, n_pass)
    for cpc in (1, 4):
        out = []
        for fused in (True, False):
            sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=cpc)
            rows = []
            for i in range(0, pcm.shape[1], 480 * cpc):
                if fused:
                    rows.append(sb.process(pcm[:, i:i + 480 * cpc], want_agg=True))
                else:
                    with _aggregate_pass():
                        rows.append(sb.process(pcm[:, i:i + 480 * cpc], want_agg=True))
            out.append(rows)
        for a, b in zip(*out):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert sum(int(r[1].sum()) for r in out[0]) == int(n_det.sum())


@pytest.mark.parametrize("cpc", [1, 2, 5])
def test_one_stream_live_equals_offline(ra, ctx, cpc):
    """One stream alone is a batch too: fed one, two or five chunks per call (3 .. 15 new windows: the shapes the single-stream mirror
    hands to dtw_single_kernel) a live-stream batch of ONE stream takes the matrix-core kernel with frames from global memory and gives
    the aggregates of the offline call bit for bit -- a stream's bits do not depend on the batch it is scored in."""
    K, L, T = 5, 40, 8
    templates = orc.synth_templates(SEED + 19, T, L, K)
    tm = ra.Templates(ctx, templates)
    pcm = np.stack([orc.synth_pcm(SEED, 150, 480 * 70)])
    cfg = ra.DetectorConfig()
    cfg.threshold, cfg.min_scores = 0.3, 1
    _, _, scores, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    with _registers_only():
        _, _, reg, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    assert rel_close(scores, reg, 2e-6) and not np.array_equal(scores, reg)
    sb = ra.StreamBatch(ctx, tm, cfg, 1, max_chunks_per_call=cpc)
    compared = 0
    for i in range(0, pcm.shape[1], 480 * cpc):
        _, _, a = sb.process(pcm[:, i:i + 480 * cpc], want_agg=True)
        f0 = 3 * (i // 480) - 3  # frame index of the call's first aggregate column; window = frame - (L - 1)
        for k in range(a.shape[1]):
            wi = f0 + k - (L - 1)
            if 0 <= wi < agg.shape[1]:
                assert a[0, k] == agg[0, wi]
                compared += 1
    assert compared > 100


def test_matrix_core_sweep_few_cases(ra):
    """A few cases of the randomised sweep in the kernel's shapes (tests/sweep_parity.py --mfma-cases; 600 of them in
    profiles/sweep_r03.txt): oracle detections (chunk and counter exact, scores 1e-5), the gate-skip call against the full one and
    the live-stream batch against the offline batch bit for bit."""
    import sweep_parity
    ctx = ra.BatchContext(0)
    n, total, ties = sweep_parity.run_sweep(ra, ctx, 10, 3, mfma=True)
    assert n == 10 and ties == 0


@pytest.mark.parametrize("K,L,T", [(16, 40, 8), (16, 12, 3), (16, 61, 11), (13, 37, 5), (13, 24, 8), (13, 100, 3)])
def test_wide_frames_match_the_oracle(ra, ctx, K, L, T):
    """mfcc_size 13 / 16 (dtw_mfma_wide3_kernel, rp_dtw_mfma_wide3.hip: three bf16 parts, chunks of four, the default arithmetic;
    dtw_mfma_wide_kernel, rp_dtw_mfma_wide.hip: two f16 parts, chunks of eight, RP_ARITH_FAST_SPLIT): every length at least three times, band 5, through the batched
    detector (its frame rows end with slack: the kernel reads its frames from global memory).  Scores against the oracle, against the
    wide register kernels (2e-6, not the same bits), and the live-stream batch against the offline call bit for bit."""
    S, N = 3, 480 * 45
    templates = orc.synth_templates(SEED + K + L, T, L, K)
    tm = ra.Templates(ctx, templates)
    pcm = np.stack([orc.synth_pcm(SEED, 300 + s, N) for s in range(S)])
    cfg = ra.DetectorConfig()
    cfg.threshold, cfg.min_scores = 0.3, 1
    _, _, scores, agg = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    for s in range(S):
        ref_s, ref_a = orc.score_stream(orc.mfcc_stream(pcm[s], K), templates)
        assert rel_close(scores[s], ref_s), np.abs(scores[s] / ref_s - 1).max()
        assert rel_close(agg[s], ref_a)
    with _registers_only():
        _, _, reg, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    matrix = _fast(ctx) or L >= 16          # the three-part kernel's first block is 16 columns (the two-part one's 12)
    assert rel_close(scores, reg, 2e-6) and np.array_equal(scores, reg) == (not matrix)
    ctx.dtw_kernels()
    ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    if matrix:
        assert "dtw_mfma_wide_kernel" in ctx.dtw_kernels() and ctx.last_dtw_products == (["f16x2"] if _fast(ctx) else ["bf16x3"])
    sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=2)
    compared = 0
    for i in range(0, N, 960):
        _, _, a = sb.process(pcm[:, i:i + 960], want_agg=True)
        f0 = 3 * (i // 480) - 3
        for k in range(a.shape[1]):
            wi = f0 + k - (L - 1)
            if 0 <= wi < agg.shape[1]:
                assert np.array_equal(a[:, k], agg[:, wi])
                compared += 1
    assert compared >= 30


def test_wide_frames_with_a_rare_length_keep_the_register_kernels(ra, ctx):
    """A length that occurs fewer than three times: the matrix kernel would pay for eight template slots -- the whole set stays with
    the wide register kernels (the switch changes nothing)."""
    K = 16
    templates = orc.synth_templates(SEED + 71, 5, 40, K)
    templates[4] = templates[4][:33].copy()
    tm = ra.Templates(ctx, templates)
    pcm = np.stack([orc.synth_pcm(SEED, 320 + s, 480 * 40) for s in range(2)])
    cfg = ra.DetectorConfig()
    _, _, scores, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    with _registers_only():
        _, _, reg, _ = ctx.batch_detect(pcm, tm, cfg, want_scores=True)
    assert np.array_equal(scores, reg)
    ref_s, _ = orc.score_stream(orc.mfcc_stream(pcm[0], K), templates)
    assert rel_close(scores[0], ref_s)


def test_full_size_against_the_register_kernels(ra):
    """BASELINE config C3 at FULL size (65 536 streams x 297 windows x 8 templates = 155.7 M scores): the matrix-core kernel against the
    register kernels on every score -- the largest relative difference stays below 2e-6 (measured 8.95e-7 in round 4, 1.4e-6 in round 3
    before the template image rounded to nearest; parity gate against the reference: 1e-5), no
    score is further than that from the vector-only arithmetic, and the run is bit-reproducible."""
    import torch
    from test_gpu_parity import _full_size_run
    ctx, tmpl, cfg, templates, pcm, scores, agg, det, n_det = _full_size_run(ra, 65536, 8)
    S, N = 65536, 64000
    reg = torch.empty_like(scores)
    with _registers_only():
        ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, reg.data_ptr(), agg.data_ptr())
        torch.cuda.synchronize()
    again = torch.empty_like(scores)
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, again.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(again, scores)
    assert not torch.equal(reg, scores), "the matrix-core kernel did not run"
    worst = 0.0
    for i in range(0, S, 8192):  # in slices: the difference of two 623 MB arrays
        a, b = scores[i:i + 8192], reg[i:i + 8192]
        worst = max(worst, float(((a - b).abs() / b).max()))
    assert worst < 2e-6, worst
    print("largest relative difference over %d scores: %.3g" % (scores.numel(), worst))
