"""The arithmetic of the DTW cost's cosine products is part of the ABI (include/rustpotter_hip.h: RP_CTX_ARITH_* flags of rp_ctx_new,
rp_ctx_set_arithmetic / rp_ctx_arithmetic, RP_DTW_PRODUCTS_* in rp_ctx_dtw_kernels) -- the reference's "config is a struct" convention
(src/config.rs:172-219), not an environment variable.  What each mode launches, that the library reports it, and that a drop-in caller
who needs every product as an f32 FMA gets exactly that from a flag."""
import ctypes as C

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


def _streams(S, n_frames, K, first=0):
    n = 480 * (n_frames // 3 + 2)
    return np.stack([orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(S)])


def test_flags_of_rp_ctx_new(ra):
    L = ra.load_library()
    h = C.c_void_p()
    assert L.rp_ctx_new(0, 1 | 4 | 8, C.byref(h)) < 0          # RP_CTX_ARITH_STRICT_F32 | RP_CTX_ARITH_FAST_SPLIT
    L.rp_last_error.restype = C.c_char_p
    assert b"exclude each other" in L.rp_last_error()
    for name, ragged in (("f32_matrix", False), ("strict_f32", False), ("fast_split", False), ("fast_split", True)):
        ctx = ra.BatchContext(device=0, host_pointers=True, arithmetic=name, ragged_matrix=ragged)
        assert ctx.get_arithmetic() == (name, ragged)
    ctx = ra.BatchContext(device=0, host_pointers=True)
    assert ctx.get_arithmetic() == ("f32_matrix", False)       # the default: f32-grade products on the matrix cores
    assert L.rp_ctx_set_arithmetic(ctx._h, 7, 0) < 0 and b"RP_ARITH" in L.rp_last_error()
    assert ctx.get_arithmetic() == ("f32_matrix", False)
    with ctx.arithmetic("strict_f32"):
        assert ctx.get_arithmetic() == ("strict_f32", False)
    assert ctx.get_arithmetic() == ("f32_matrix", False)


def test_what_each_arithmetic_launches(ra):
    """mfcc_size 5, eight templates of one length (BASELINE C2 / C3's shape): dtw_mfma_kernel with three bf16 parts by default, with two
    f16 parts on request, the register kernels in strict mode -- one template set, three modes, all within 1e-5 of the oracle, and the
    three results are three different sets of bits (each mode really ran its own arithmetic)."""
    ctx = ra.BatchContext(device=0, host_pointers=True)
    K, T, L = 5, 8, 40
    templates = orc.synth_templates(SEED + 3, T, L, K)
    mf = _streams(3, 70 + L - 1, K, first=40)
    tm = ra.Templates(ctx, templates)
    out = {}
    for name, kernel, products in (("f32_matrix", "dtw_mfma_kernel", ["bf16x3"]), ("fast_split", "dtw_mfma_kernel", ["f16x2"]),
                                   ("strict_f32", "register kernels", [])):
        ctx.set_arithmetic(name)
        ctx.dtw_kernels()
        out[name], _, _ = ctx.dtw_scores(mf, tm)
        assert ctx.dtw_kernels() == [kernel] and ctx.last_dtw_products == products, name
        for s in range(3):
            ref_s, _ = orc.score_stream(mf[s], templates)
            assert np.all(np.abs(out[name][s] - ref_s) <= 1e-5 * ref_s), name
    assert not np.array_equal(out["f32_matrix"], out["fast_split"])
    assert not np.array_equal(out["f32_matrix"], out["strict_f32"])
    assert not np.array_equal(out["fast_split"], out["strict_f32"])


def test_wide_frames_and_ragged_sets_per_arithmetic(ra):
    """mfcc_size 16: dtw_mfma_wide3_kernel (three bf16 parts, chunks of four) by default, dtw_mfma_wide_kernel (two f16 parts, chunks of
    eight) in RP_ARITH_FAST_SPLIT, the wide register kernels in RP_ARITH_STRICT_F32.  Templates of unequal length (dtw_ragged_kernel)
    exist as a two-part f16 kernel only: the default arithmetic scores them with the f32 vector kernels -- bit for bit what
    RP_ARITH_STRICT_F32 gives -- and the opt-in brings the matrix kernel back."""
    ctx = ra.BatchContext(device=0, host_pointers=True)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    # mfcc_size 16, eight templates of 40 frames
    t16 = orc.synth_templates(SEED + 16, 8, 40, 16)
    pcm = np.stack([orc.synth_pcm(SEED, 600 + s, 480 * 45) for s in range(2)])
    tm16 = ra.Templates(ctx, t16)
    ctx.dtw_kernels()
    _, _, dflt, _ = ctx.batch_detect(pcm, tm16, cfg, want_scores=True)
    assert ctx.dtw_kernels() == ["dtw_mfma_wide_kernel"] and ctx.last_dtw_products == ["bf16x3"]
    with ctx.arithmetic("strict_f32"):
        _, _, strict, _ = ctx.batch_detect(pcm, tm16, cfg, want_scores=True)
        assert ctx.dtw_kernels() == ["register kernels"] and ctx.last_dtw_products == []
    assert not np.array_equal(dflt, strict) and np.all(np.abs(dflt - strict) <= 2e-6 * strict)
    with ctx.arithmetic("fast_split"):
        _, _, fast, _ = ctx.batch_detect(pcm, tm16, cfg, want_scores=True)
        assert "dtw_mfma_wide_kernel" in ctx.dtw_kernels() and ctx.last_dtw_products == ["f16x2"]
    assert not np.array_equal(fast, dflt) and np.all(np.abs(fast - strict) <= 2e-6 * strict)
    for s in range(2):
        ref_s, _ = orc.score_stream(orc.mfcc_stream(pcm[s], 16), t16)
        for got in (dflt, strict, fast):
            assert np.all(np.abs(got[s] - ref_s) <= 1e-5 * ref_s)
    # mfcc_size 5, five templates of unequal length (the shape of the reference's oye_casa_g.rpw), whole streams
    tt = orc.synth_templates(SEED + 5, 5, 108, 5)
    rag = [np.ascontiguousarray(t[:n]) for t, n in zip(tt, (108, 96, 90, 93, 102))]
    pcm5 = np.stack([orc.synth_pcm(SEED, 700 + s, 480 * 120) for s in range(2)])
    tm5 = ra.Templates(ctx, rag)
    ctx.dtw_kernels()
    _, _, dflt, _ = ctx.batch_detect(pcm5, tm5, cfg, want_scores=True)
    assert ctx.dtw_kernels() == ["register kernels"]
    with ctx.arithmetic("fast_split"):                         # without the ragged flag: still the register kernels
        _, _, fs, _ = ctx.batch_detect(pcm5, tm5, cfg, want_scores=True)
        assert ctx.dtw_kernels() == ["register kernels"]
    assert np.array_equal(fs, dflt)
    with ctx.arithmetic("fast_split", ragged_matrix=True):
        _, _, rg, _ = ctx.batch_detect(pcm5, tm5, cfg, want_scores=True)
        assert "dtw_ragged_kernel" in ctx.dtw_kernels()
    assert not np.array_equal(rg, dflt) and np.all(np.abs(rg - dflt) <= 4e-6 * dflt)
