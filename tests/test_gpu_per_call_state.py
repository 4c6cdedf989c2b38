"""The few words of device state a context keeps between calls (tile counters, the out-of-range list, can-fire flags, the model forward's redo
list): 200 calls of different kinds in a scrambled order, each kind gives the same bits every time."""
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
G = simstream.GOLDEN
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


def _streams(S, n_frames, K=5, first=0):
    n = 480 * (n_frames // 3 + 2)
    mf = [orc.mfcc_stream(orc.synth_pcm(SEED, first + s, n), K)[:n_frames] for s in range(S)]
    assert all(m.shape[0] == n_frames for m in mf)
    return np.stack(mf)


# ------------------------------------------------------------------------------------------------ per-call device state
def test_per_call_words_survive_interleaved_calls(ra, ctx):
    """The context keeps a few words of device state that every call must leave at zero for the next one: the matrix-core kernels' tile
    counters, the list of out-of-range pairs (+ its ALL-mode overflow), the per-stream can-fire flags (cleared by the scan that reads
    them), the model forward's redo list.  200 calls of different kinds in a scrambled order on ONE context: each kind gives the same
    bits every time, whatever ran before it."""
    rng = np.random.default_rng(21)
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    pcm = np.stack([base[:n], np.roll(base[:n], 480 * 5), np.roll(base[:n], 480 * 17)])
    cfg = ra.DetectorConfig()
    cfg_gate0 = ra.DetectorConfig()
    cfg_gate0.avg_threshold = 0.0
    tm = ra.Templates(ctx, templates, avg=w["avg_features"])
    same = orc.synth_templates(SEED + 77, 8, 60, 5)            # one chunk for the matrix-core kernel (folded Max, tile counters)
    tm8 = ra.Templates(ctx, same)
    tiny = [(np.asarray(t, np.float64) * 3e-23).astype(np.float32) for t in templates]
    tmt = ra.Templates(ctx, tiny, avg=(np.asarray(w["avg_features"], np.float64) * 3e-23).astype(np.float32))   # ref_only set
    mf = _streams(4, 60 + 90, 5, first=60)
    mf_tiny = (mf.astype(np.float64) * 1e-12).astype(np.float32)
    dims = (320, 32, 16, 2)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    model = ra.Model(ctx, ws, bs)
    x = rng.standard_normal((300, dims[0])).astype(np.float32)
    xb = x.copy()
    xb[::7, 3] = 1e9

    def digest(*arrays):
        return tuple(np.ascontiguousarray(a).tobytes() for a in arrays)

    kinds = {
        "detect": lambda: digest(*ctx.batch_detect(pcm, tm, cfg, max_det=4)),                      # gated, detect-only
        "detect_scores": lambda: digest(*ctx.batch_detect(pcm, tm, cfg_gate0, max_det=4, want_scores=True)),
        "detect_same8": lambda: digest(*ctx.batch_detect(pcm, tm8, cfg_gate0, max_det=4)),         # matrix-core kernel, Max folded in, hot flags
        "detect_ref_only": lambda: digest(*ctx.batch_detect(pcm, tmt, cfg, max_det=4)),            # every window through dtw_ref_kernel
        "scores": lambda: digest(*ctx.dtw_scores(mf, tm8)),
        "scores_tiny": lambda: digest(*ctx.dtw_scores(mf_tiny, tm8)),                              # every pair listed and rescored
        "mlp": lambda: digest(ctx.mlp_forward(x, model)),
        "mlp_big": lambda: digest(ctx.mlp_forward(xb, model)),                                     # rows listed for the f32 pass
        "ingest": lambda: digest(*ctx.batch_detect_ingest(pcm, tm, cfg, max_det=4, block_streams=2)[:2]),
    }
    first = {k: f() for k, f in kinds.items()}
    assert first["detect"] == first["ingest"]
    names = list(kinds)
    for i in range(200):
        k = names[int(rng.integers(len(names)))]
        assert kinds[k]() == first[k], (i, k)
