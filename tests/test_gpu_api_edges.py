"""Edges of the batched C ABI: several contexts on one device (large-LDS kernels per context), very long templates, NULL arguments with live handles."""
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
G = simstream.GOLDEN


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


def _wakeword(ra, ctx, name="oye_casa_g.rpw"):
    w = rpw_py.load_rpw(os.path.join(G, name))
    return ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])


def test_contexts_on_one_device_use_large_lds_kernels(ra):
    """The >64 KB dynamic-LDS attribute is set per (device, kernel), not once per process: a second context still
    launches the generic DTW kernel with a long template (LDS above 64 KB)."""
    K, L = 24, 600   # (64 + L - 1) * 25 * 4 B = 66 KB of frames alone
    rng = np.random.default_rng(1)
    templates = [rng.standard_normal((L, K)).astype(np.float32)]
    mf = rng.standard_normal((2, L + 70, K)).astype(np.float32)
    outs = []
    for _ in range(2):
        ctx = ra.BatchContext(0)
        scores, _, _ = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
        outs.append(scores)
    assert np.array_equal(outs[0], outs[1]) and np.isfinite(outs[0]).all()
    ref, _ = orc.score_stream(mf[0][:L + 3], templates)
    assert np.allclose(outs[0][0][:ref.shape[0]], ref, rtol=1e-5, atol=0)


def test_very_long_templates(ra):
    """Templates far beyond the usual 1-2 s: 25 s (the register kernels with > 64 KB of LDS), 50 s (too long for them: the
    generic kernel), and 90 s (refused with a text that names the limit, not a launch failure)."""
    ctx = ra.BatchContext(0)
    K = 5
    for L in (2500, 5000):
        rng = np.random.default_rng(L)
        templates = [rng.standard_normal((L, K)).astype(np.float32), rng.standard_normal((L - 7, K)).astype(np.float32)]
        mf = rng.standard_normal((2, L + 20, K)).astype(np.float32)
        sc, _, _ = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
        ref, _ = orc.score_stream(mf[1][:L + 2], templates)
        assert np.allclose(sc[1][:ref.shape[0]], ref, rtol=1e-5, atol=0)
    with pytest.raises(ra.RustpotterError, match="too long for the device kernels"):
        ra.Templates(ctx, [np.ones((9000, K), np.float32)])


# ------------------------------------------------------------------ NULL arguments with live handles
def test_null_arguments_with_live_handles(ra):
    """config / pcm / det == NULL next to a valid context and template set is an error return, not a crash."""
    import ctypes as C
    L = ra.load_library()
    ctx = ra.BatchContext(0)
    tm = _wakeword(ra, ctx)
    cfg = ra.DetectorConfig()._c()
    pcm = np.zeros((1, 4800), np.float32)
    det = np.zeros((1, 4), dtype=[("a", "<i4", 6)])
    n_det = np.zeros(1, np.int32)
    ok = L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, C.byref(cfg), det.ctypes.data, n_det.ctypes.data, 4, None, None)
    assert ok == 0
    assert L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, None, det.ctypes.data, n_det.ctypes.data, 4, None, None) == -1
    assert b"null" in L.rp_last_error()
    assert L.rp_batch_detect_fmt(ctx._h, None, 3, 1, 4800, 4800, tm._h, C.byref(cfg), det.ctypes.data, n_det.ctypes.data, 4, None, None) == -1
    assert L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, C.byref(cfg), None, n_det.ctypes.data, 4, None, None) == -1
    h = C.c_void_p()
    assert L.rp_stream_batch_new(ctx._h, tm._h, None, 4, 1, C.byref(h)) == -1
    assert L.rp_detect_scan(ctx._h, pcm.ctypes.data, None, 1, 10, 5, None, 0, None, 0, det.ctypes.data, n_det.ctypes.data, 4) == -1
