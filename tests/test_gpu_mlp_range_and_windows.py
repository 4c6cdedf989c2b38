"""The wakeword-model forward (src/wakewords/nn/wakeword_nn.rs:101-106: candle f32) for features of any finite size -- RP_MLP_F32 (three bf16
parts: the f32 exponent range) and RP_MLP_F32_FAST (two f16 parts: rows beyond the f16 range go to the f32 matrix instructions) -- and the
forward over every window of a stream (rp_mlp_forward_windows) against the oracle: window means, other mfcc sizes, edges."""
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
