"""mlp_stream_kernel, the line-streaming wakeword-model forward (rp_mlp_stream.hip): shapes and batch sizes with ragged last tiles, every
arithmetic of the ABI (RP_MLP_F32 = three bf16 parts, RP_MLP_F32_FAST = two f16 parts, RP_MLP_F32_STRICT, RP_MLP_BF16), unaligned row starts and
NaN neighbours."""
import os

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


# ------------------------------------------------------------------ the line-streaming MLP kernel
@pytest.mark.parametrize("dims", [(3120, 32, 16, 2), (3120, 13, 2), (1040, 32, 16, 3), (64, 13, 2), (4096, 20, 255, 4)])
@pytest.mark.parametrize("B", [1, 15, 257, 1300])
def test_mlp_stream_kernel_shapes_and_batch_sizes(ra, ctx, dims, B):
    """mlp_stream_kernel (layer-1 width <= 32, row pitch a multiple of 64 bytes) at batch sizes that leave ragged last tiles,
    with row pitches that put odd rows in the middle of a 128-byte line (3120, 1040 floats) and ones that do not (64, 4096),
    against the oracle (f32 1e-5; bf16 vs the bf16-rounding oracle 1e-3); bit-identical to itself on a second call and
    independent of where a row sits in the batch."""
    os.environ["RP_MLP_STREAM"] = "2"   # the stream kernel for f32 too (the library picks it for bf16 only by default)
    rng = np.random.default_rng(sum(dims) + B)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
    x = rng.standard_normal((B, dims[0])).astype(np.float32)
    model = ra.Model(ctx, ws, bs)
    got = ctx.mlp_forward(x, model)
    ref = orc.mlp_forward(x, ws, bs)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5), np.abs(got - ref).max()
    assert ctx.mlp_forward(x, model).tobytes() == got.tobytes()
    if B > 2:   # a row's logits do not depend on its position (even / odd rows start at different line phases)
        sub = ctx.mlp_forward(x[1:], model)
        assert sub.tobytes() == got[1:].tobytes()
    if len(dims) > 3 and dims[2] > 200:
        # a hidden layer this wide does not fit the fused kernels' LDS: the per-layer f32 kernel serves it, there is no bf16 form
        with pytest.raises(ra.RustpotterError, match="no bf16 MFMA kernel"):
            ctx.mlp_forward(x, model, precision="bf16")
        os.environ.pop("RP_MLP_STREAM")
        return
    got16 = ctx.mlp_forward(x, model, precision="bf16")
    ref16 = orc.mlp_forward(x, ws, bs, bf16_layer1=True)
    assert np.allclose(got16, ref16, rtol=1e-3, atol=1e-3), np.abs(got16 - ref16).max()
    os.environ.pop("RP_MLP_STREAM")
    # and as the library chooses by itself: bf16 streamed; f32 callers (RP_MLP_F32) streamed too, layer 1 from exact three-part bf16 splits
    # of inputs and weights (kMlpBf16x3) -- `got` above is that form already; RP_MLP_F32_FAST = f16 two-way splits (kMlpF16x2: within 1e-5 of
    # the f32 matrix instructions and not their bits), batch-invariant and bit-reproducible like the other forms; a row with a feature beyond
    # the f16 range comes from the f32 matrix instructions there
    assert ctx.mlp_forward(x, model).tobytes() == got.tobytes()
    assert "bf16x3" in ctx.last_mlp_kernel()
    split = ctx.mlp_forward(x, model, precision="f32_fast")
    assert "f16x2" in ctx.last_mlp_kernel()
    assert np.allclose(split, ref, rtol=1e-5, atol=1e-5)
    exact = ctx.mlp_forward(x, model, precision="f32_strict")
    assert "f32 matrix instructions" in ctx.last_mlp_kernel()
    assert np.allclose(split, exact, rtol=1e-5, atol=1e-5), np.abs(split - exact).max()
    assert np.allclose(got, exact, rtol=1e-5, atol=1e-5), np.abs(got - exact).max()
    assert np.allclose(split, got, rtol=1e-5, atol=1e-5), np.abs(split - got).max()
    if B > 200:   # three arithmetics, three sets of bits
        assert split.tobytes() != exact.tobytes() and split.tobytes() != got.tobytes() and got.tobytes() != exact.tobytes()
    assert ctx.mlp_forward(x, model, precision="f32_fast").tobytes() == split.tobytes()
    if B > 2:
        assert ctx.mlp_forward(x[1:], model, precision="f32_fast").tobytes() == split[1:].tobytes()
        # a feature beyond the f16 range: the two-part form has its row computed again by the f32 matrix instructions (round 4; it used to be
        # NaN); the three-part form has the f32 exponent range and needs no second pass
        xb = x.copy()
        xb[1, 7] = 7.0e4
        big = ctx.mlp_forward(xb, model, precision="f32_fast")
        strict = ctx.mlp_forward(xb, model, precision="f32_strict")
        assert big[1].tobytes() == strict[1].tobytes() and np.isfinite(big).all()
        assert np.delete(big, 1, axis=0).tobytes() == np.delete(split, 1, axis=0).tobytes()
        big3 = ctx.mlp_forward(xb, model)
        assert np.isfinite(big3).all() and np.allclose(big3[1], strict[1], rtol=2e-5, atol=2e-5 * 7.0e4)
        assert np.delete(big3, 1, axis=0).tobytes() == np.delete(got, 1, axis=0).tobytes()
    assert ctx.mlp_forward(x, model, precision="bf16").tobytes() == got16.tobytes()


def test_mlp_stream_kernel_unaligned_base_and_nan_neighbours(ra):
    """Device-pointer form with the rows starting at every 16-byte offset inside a 128-byte line (the kernel reads whole lines
    from the line start below a row: what lies before the first row and behind the last one must never reach a result),
    NaN planted around the array and in a neighbouring row: only that row's logits are NaN."""
    import torch
    dims = (3120, 32, 16, 2)
    os.environ["RP_MLP_STREAM"] = "2"
    rng = np.random.default_rng(9)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    dctx = ra.BatchContext(0, host_pointers=False)
    dctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(dctx, ws, bs)
    B = 300
    x = rng.standard_normal((B, dims[0])).astype(np.float32)
    ref = orc.mlp_forward(x, ws, bs)
    outs = []
    for off in range(0, 32, 4):   # float offsets 0, 4, .. 28 = 16-byte steps through a line
        buf = torch.full((off + B * dims[0] + 64,), float("nan"), dtype=torch.float32, device="cuda")
        buf[off:off + B * dims[0]] = torch.from_numpy(x.reshape(-1)).cuda()
        out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()   # torch's default stream is handle 0 = "the context's own stream" to rp_ctx_set_stream: order by hand
        dctx.mlp_dev(model, buf.data_ptr() + 4 * off, B, "f32", out.data_ptr())
        dctx.synchronize()
        o = out.cpu().numpy()
        assert np.isfinite(o).all() and np.allclose(o, ref, rtol=1e-5, atol=1e-5), off
        outs.append(o)
    assert all(o.tobytes() == outs[0].tobytes() for o in outs)   # the k order does not depend on the phase
    xn = x.copy()
    xn[17, 5] = np.nan
    buf = torch.from_numpy(xn.reshape(-1)).cuda()
    out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    dctx.mlp_dev(model, buf.data_ptr(), B, "f32", out.data_ptr())
    dctx.synchronize()
    o = out.cpu().numpy()
    assert np.isnan(o[17]).all() and np.isfinite(np.delete(o, 17, axis=0)).all()
    assert np.delete(o, 17, axis=0).tobytes() == np.delete(outs[0], 17, axis=0).tobytes()
    os.environ.pop("RP_MLP_STREAM")
