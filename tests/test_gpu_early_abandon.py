"""Early abandon of detect-only calls (ScoreMode::Max): a DTW whose cheapest band cell can no longer reach the threshold stops -- never a
changed detection, for every kernel family and at degenerate thresholds."""

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


# ------------------------------------------------------------------ early abandon in detect-only calls
@pytest.mark.parametrize("K,lens", [(5, [40] * 8), (5, [40, 36, 33, 40, 31]), (13, [40, 40, 35]), (16, [38, 40])])
def test_early_abandon_never_changes_a_detection(ra, K, lens):
    """Detect-only calls in ScoreMode::Max stop DTWs that can no longer reach `threshold` (launch_dtw, abandon_nc).  With
    the threshold placed INSIDE the score distribution (many windows just above and just below it) the detections must
    equal, bit for bit, those of the call that returns every score -- offline and fed chunk by chunk."""
    S, N = 400, 480 * 60
    templates = [t[:n].copy() for t, n in zip(orc.synth_templates(SEED, len(lens), max(lens), K), lens)]
    ctx = ra.BatchContext(0)
    pcm = ctx.synth_pcm(SEED, 0, S, N)
    tm = ra.Templates(ctx, templates)
    for q, min_scores in ((0.5, 1), (0.9, 2), (0.99, 1), (0.999, 1)):
        cfg = ra.DetectorConfig()
        cfg.avg_threshold, cfg.min_scores = 0.0, min_scores
        d0, n0, scores, agg = ctx.batch_detect(pcm, tm, cfg, max_det=8, want_scores=True)   # every score, nothing abandoned
        cfg.threshold = float(np.quantile(agg, q))
        d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=8, want_scores=True)
        d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=8)                            # detect-only: may abandon
        assert n_full.sum() > 20, "the case must produce detections"
        assert np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes(), (K, q)
    # live-stream batches take the same shortcut when the per-window aggregates are not asked for
    sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=3)
    live = [[] for _ in range(S)]
    for i in range(0, N, 480 * 3):
        d, nd = sb.process(np.ascontiguousarray(pcm[:, i:i + 480 * 3]), max_det=8)
        for s_ in range(S):
            live[s_] += [d[s_][j] for j in range(nd[s_])]
    for s_ in range(S):
        assert len(live[s_]) == n_full[s_]
        for a, b in zip(live[s_], d_full[s_][:min(n_full[s_], 8)]):
            assert (a["frame"], a["window"], a["counter"], a["score"]) == (b["frame"], b["window"], b["counter"], b["score"])


def test_early_abandon_degenerate_thresholds(ra):
    """threshold <= 0 (everything fires), >= 1 (nothing can), and a threshold so high that every DTW is abandoned at the
    first check: same detections as the full path."""
    S, N, K = 64, 480 * 40, 5
    templates = orc.synth_templates(SEED, 3, 30, K)
    ctx = ra.BatchContext(0)
    pcm = ctx.synth_pcm(SEED, 0, S, N)
    tm = ra.Templates(ctx, templates)
    for thr in (-1.0, 0.0, 0.999, 1.0, 1.5):
        cfg = ra.DetectorConfig()
        cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.0, thr, 1
        d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
        d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        assert np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes(), thr
    # (with every window above the threshold the countdown never reaches zero: eager mode makes those cases fire)
    cfg.threshold, cfg.eager = 0.0, True
    d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
    d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=4)
    assert n_full.sum() >= S and np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes()
