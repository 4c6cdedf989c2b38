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


class _env:
    def __init__(self, **kv):
        self.kv = kv
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _ragged_templates(seed, lens, K=5):
    tt = orc.synth_templates(seed, len(lens), max(lens), K)
    return [np.ascontiguousarray(t[:L]) for t, L in zip(tt, lens)]


def _check(ra, ctx, templates, mf, band=5, score_ref=0.22, mode=None, expect_ragged=True, tol=1e-5):
    tm = ra.Templates(ctx, templates)
    kw = {} if mode is None else {"score_mode": mode[0]}
    ctx.dtw_kernels()
    with _env(RP_DTW_RAGGED="1"):
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
    with _env(RP_DTW_RAGGED="1"):
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
    with _env(RP_DTW_RAGGED="1"):
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
