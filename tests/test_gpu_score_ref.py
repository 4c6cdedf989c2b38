"""DetectorConfig.score_ref (config.rs:172-209) well below the default 0.22 on the matrix-core DTW shapes, in both matrix arithmetics: the
relative score error grows like 1 / score_ref; the two-part f16 form has a floor (0.05) below which the register kernels score, the
three-part bf16 form has f32-grade products and none."""

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
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
    matrix = (T >= 5 and band <= 5) or (T >= 3 and band == 5)
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
