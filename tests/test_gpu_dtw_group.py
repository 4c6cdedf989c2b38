"""dtw_mfma_group_kernel (rustpotter_amd/csrc/rp_dtw_mfma_group.hip): references with several chunks of one template length -- BASELINE config
C4, 64 templates of 100 frames -- score a long batch with workgroups in which the four chunk-waves of a 32-window tile share the
column's B operand through an LDS ring.  Same operations on the same values as dtw_mfma_kernel: every test holds it to the BITS of that
kernel (RP_DTW_GROUP=0), asserts through rp_ctx_dtw_kernels that the group form really ran, and pins a sample against the oracle."""
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


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


class _env:
    """RP_DTW_GROUP for the calls inside (read per call): "0" dtw_mfma_kernel only, "2" the group form whatever the launch size."""
    def __init__(self, value):
        self.value = value
    def __enter__(self):
        import rustpotter_amd
        self.arith = rustpotter_amd.arithmetic_all("fast_split")   # the group form exists for the two-part f16 products only (four three-part images do not fit the LDS)
        self.arith.__enter__()
        self.old = os.environ.get("RP_DTW_GROUP")
        os.environ["RP_DTW_GROUP"] = self.value
    def __exit__(self, *a):
        self.arith.__exit__(*a)
        if self.old is None:
            del os.environ["RP_DTW_GROUP"]
        else:
            os.environ["RP_DTW_GROUP"] = self.old


def rel_close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-30)))


def _streams(S, n_frames, K=5, first=0, base=6):
    n = 480 * (n_frames // 3 + 2)
    mf = np.stack([orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(base)])
    rng = np.random.default_rng(first + S)
    return mf[rng.integers(0, base, S)] * (0.5 + rng.random(S)).astype(np.float32)[:, None, None]


def _templates(counts, K=5, seed=0):
    """counts: {length: number of templates}; templates of one length sit together (the library sorts by length anyway)."""
    out = []
    for L, c in counts.items():
        out += orc.synth_templates(SEED + seed + L, c, L, K)
    return out


def _both(ra, ctx, mf, templates, band=5, expect_plain=False):
    tm = ra.Templates(ctx, templates)
    with _env("0"):
        ctx.dtw_kernels()
        ref, _, ref_agg = ctx.dtw_scores(mf, tm, band_size=band)
        assert "dtw_mfma_group_kernel" not in ctx.dtw_kernels()
    with _env("2"):
        got, _, agg = ctx.dtw_scores(mf, tm, band_size=band)
        ran = ctx.dtw_kernels()
    assert ("dtw_mfma_group_kernel" in ran) != expect_plain, ran
    assert np.isfinite(ref).all() and ref.min() > 0.0
    assert np.array_equal(got, ref) and np.array_equal(agg, ref_agg)
    return tm, got


@pytest.mark.parametrize("counts,S,n_win,plain", [
    ({100: 64}, 24, 297, False),         # C4's reference: two groups of four chunks
    ({40: 32}, 50, 77, False),           # one group; tiles straddle streams
    ({36: 48}, 9, 101, False),           # a group and two rest chunks
    ({25: 21, 31: 29}, 20, 64, False),   # 21 = 8 + 8 + 5: rest chunks only; 29 = 8 + 8 + 8 + 5: a group
    ({64: 37}, 3, 33, False),            # a group and a rest chunk, three streams: most tile slots of the last workgroup are past the end
    ({30: 16}, 37, 45, True),            # two chunks: no group (the two-chunk shape measured slower than the plain kernel)
    ({110: 32}, 6, 80, True),            # too long for four images in LDS
])
def test_same_bits_as_the_plain_matrix_kernel(ra, ctx, counts, S, n_win, plain):
    templates = _templates(counts)
    L = max(counts)
    mf = _streams(S, n_win + L - 1, first=7 * L)
    tm, got = _both(ra, ctx, mf, templates, expect_plain=plain)
    for s in (0, S - 1):
        ref_s, _ = orc.score_stream(mf[s], templates)
        assert rel_close(got[s], ref_s), np.abs(got[s] / ref_s - 1).max()


@pytest.mark.parametrize("band", [3, 4])
def test_bands_three_and_four(ra, ctx, band):
    templates = _templates({37: 32, 50: 16}, seed=band)
    mf = _streams(11, 60 + 49, first=90 + band)
    tm, got = _both(ra, ctx, mf, templates, band=band)
    ref_s, _ = orc.score_stream(mf[4], templates, band=band)
    assert rel_close(got[4], ref_s)


def test_rest_chunks_other_classes_and_an_averaged_template(ra, ctx):
    """32 + 5 of one length (a group + a rest chunk), 4 of another (four-slot matrix kernel), two ragged ones and the averaged template
    beside the group: every launch of the call lands in the same score array."""
    K = 5
    templates = _templates({48: 37, 40: 4, 33: 1, 29: 1})
    avg = orc.synth_templates(SEED + 6, 1, 48, K)[0]
    mf = _streams(5, 48 + 70, first=40)
    tm = ra.Templates(ctx, templates, avg=avg)
    with _env("0"):
        ref, ref_avg, ref_agg = ctx.dtw_scores(mf, tm, with_avg=True, score_mode=ra.ScoreMode.Median)
    with _env("2"):
        ctx.dtw_kernels()
        got, got_avg, agg = ctx.dtw_scores(mf, tm, with_avg=True, score_mode=ra.ScoreMode.Median)
        assert "dtw_mfma_group_kernel" in ctx.dtw_kernels()
    assert np.array_equal(got, ref) and np.array_equal(agg, ref_agg) and np.array_equal(got_avg, ref_avg)
    ref_s, ref_a = orc.score_stream(mf[2], templates, mode="median")
    assert rel_close(got[2], ref_s) and rel_close(agg[2], ref_a)


def test_windows_outside_the_norm_range_are_listed_alike(ra, ctx):
    """Digital silence behind speech and a stream at a wild scale: the windows dtw_mfma_kernel hands to dtw_ref_kernel (a frame's norm outside the
    range test) are found by the tile's four waves together -- same list, same bits after the rescoring."""
    templates = _templates({30: 32})
    mf = _streams(6, 90, first=500)
    mf[1, 40:] = mf[1, 40]                 # a constant tail: zero vectors after centring
    mf[2] *= np.float32(1e-19)             # squared norms far below the range
    mf[3, 55] *= np.float32(3e18)          # one huge frame
    _, got = _both(ra, ctx, mf, templates)
    for s in (1, 2, 3):
        ref_s, _ = orc.score_stream(mf[s], templates)
        assert rel_close(got[s], ref_s), (s, np.abs(got[s] / ref_s - 1).max())


def test_the_size_rule_and_the_modes_that_keep_the_plain_kernel(ra, ctx):
    """Unset, the group form is for launches of hundreds of tile rounds: a small call keeps dtw_mfma_kernel.  Live-stream batches, the gate's
    list and detect-only calls (early abandon) never take it."""
    templates = _templates({30: 32})
    mf = _streams(4, 80, first=700)
    tm = ra.Templates(ctx, templates)
    ctx.dtw_kernels()
    with ctx.arithmetic("fast_split"):
        small, _, _ = ctx.dtw_scores(mf, tm)
    assert "dtw_mfma_group_kernel" not in ctx.dtw_kernels()
    with _env("2"):
        forced, _, _ = ctx.dtw_scores(mf, tm)
        assert "dtw_mfma_group_kernel" in ctx.dtw_kernels()
    assert np.array_equal(small, forced)


def test_c4_share_at_full_size(ra):
    """BASELINE config C4, one GPU's share (8 192 streams x 297 windows x 64 templates of 100 frames): by the size rule the group form runs
    unasked; every score equals dtw_mfma_kernel's bit for bit (156 M scores), a stream equals itself scored alone."""
    ctx = ra.BatchContext(device=0, host_pointers=True, arithmetic="fast_split")   # the group form is a two-part-f16 kernel
    templates = _templates({100: 64}, seed=3)
    mf = _streams(8192, 396, first=900, base=16)
    tm = ra.Templates(ctx, templates)
    ctx.dtw_kernels()
    got, _, agg = ctx.dtw_scores(mf, tm)
    assert "dtw_mfma_group_kernel" in ctx.dtw_kernels()
    with _env("0"):
        ref, _, ref_agg = ctx.dtw_scores(mf, tm)
        assert "dtw_mfma_group_kernel" not in ctx.dtw_kernels()
    assert np.array_equal(got, ref) and np.array_equal(agg, ref_agg)
    one, _, _ = ctx.dtw_scores(mf[8191], tm)
    assert np.array_equal(got[8191], one[0])
