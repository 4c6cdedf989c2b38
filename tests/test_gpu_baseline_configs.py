"""BASELINE.json's configs at their full sizes against the oracle: C2 (1 024 streams x 8 templates, every stream), C3 (65 536 x 8: sampled
streams, every window), C5 (65 536 rows through the wakeword-model forward)."""

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


# ------------------------------------------------------------------ BASELINE configs at their sizes
def test_c2_full_size_vs_oracle(ra):
    """BASELINE config C2 (1 024 streams x 8 templates, 4 s streams) at size: probabilities, aggregate = row maximum,
    no detection on noise, bit-reproducible, and 16 sampled streams x ALL 297 windows against the oracle at 1e-5."""
    import torch
    S, T, N, L, K = 1024, 8, 64000, 100, 5
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(SEED, T, L, K)
    tmpl = ra.Templates(ctx, templates)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
    n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert n_win == 297 and bool(torch.isfinite(scores).all()) and float(scores.min()) > 0.0 and float(scores.max()) < 1.0
    assert torch.equal(agg, scores.max(dim=2).values) and int(n_det.sum()) == 0
    s2 = torch.empty_like(scores)
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, s2.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(scores, s2)
    for s in list(range(0, S, 73)) + [S - 1]:
        ref_pcm = orc.synth_pcm(SEED, s, N)
        ref_s, _ = orc.score_stream(orc.mfcc_stream(ref_pcm, K), templates)
        got = scores[s].cpu().numpy()
        assert ref_s.shape == got.shape and np.all(np.abs(got - ref_s) <= 1e-5 * np.abs(ref_s)), s


def test_c3_sampled_streams_all_windows_vs_oracle(ra):
    """BASELINE config C3 at size (65 536 x 8): 16 sampled streams x all 297 windows against the oracle at 1e-5 (the
    size-independent properties are test_full_size_properties)."""
    import torch
    S, T, N, L, K = 65536, 8, 64000, 100, 5
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(SEED, T, L, K)
    tmpl = ra.Templates(ctx, templates)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
    n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    picks = [0, 1, 63, 64, 4095, 4096, 21845, 21846, 32767, 32768, 43690, 43691, 54321, 65000, 65534, 65535]
    for s in picks:
        ref_s, ref_a = orc.score_stream(orc.mfcc_stream(orc.synth_pcm(SEED, s, N), K), templates)
        got = scores[s].cpu().numpy()
        assert ref_s.shape == got.shape and np.all(np.abs(got - ref_s) <= 1e-5 * np.abs(ref_s)), s
        assert np.all(np.abs(agg[s].cpu().numpy() - ref_a) <= 1e-5 * np.abs(ref_a)), s


def test_c5_full_size_model_forward(ra):
    """BASELINE config C5 at size: B = 65 536 rows x 3 120 features through the Small stack 3120 -> 32 -> 16 -> 2.
    Finite logits, bit-reproducible, a row's logits do not depend on the batch it is in, 256 sampled rows against the f32
    oracle at 1e-5 (f32 callers) and against the bf16-rounding oracle at 1e-3 (bf16 MFMA)."""
    import torch
    B, dims = 65536, [3120, 32, 16, 2]
    rng = np.random.default_rng(5)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(ctx, ws, bs)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((B, dims[0]), dtype=torch.float32, device="cuda", generator=gen)
    rows = np.unique(np.concatenate([[0, 1, 15, 16, 127, 128, B - 1], np.random.default_rng(2).integers(0, B, 256)]))
    xs = x[torch.from_numpy(rows).cuda()].contiguous()
    for prec, tol, bf in (("f32", 1e-5, False), ("bf16", 1e-3, True)):
        out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
        out2 = torch.empty_like(out)
        ctx.mlp_dev(model, x.data_ptr(), B, prec, out.data_ptr())
        ctx.mlp_dev(model, x.data_ptr(), B, prec, out2.data_ptr())
        small = torch.empty((len(rows), 2), dtype=torch.float32, device="cuda")
        ctx.mlp_dev(model, xs.data_ptr(), len(rows), prec, small.data_ptr())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()) and torch.equal(out, out2)
        got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert np.array_equal(got, small.cpu().numpy())  # batch invariance
        ref = orc.mlp_forward(xs.cpu().numpy(), ws, bs, bf16_layer1=bf)
        assert np.allclose(got, ref, rtol=tol, atol=tol), (prec, np.abs(got - ref).max())
