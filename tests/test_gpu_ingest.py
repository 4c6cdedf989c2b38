"""rp_batch_detect_ingest: streams that start in HOST memory, copied block by block under the previous block's kernels -- the detections and
scores of one resident call."""
import os

import numpy as np
import pytest

import rpw_py
import simstream

pytestmark = pytest.mark.gpu
G = simstream.GOLDEN


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


# ------------------------------------------------------------------------------------------------ host ingest
@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_batch_detect_ingest_equals_one_resident_call(ra, ctx, dtype):
    """rp_batch_detect_ingest (streams in HOST memory, taken in blocks with the next block's copy under this block's kernels, results
    copied back per block) == rp_batch_detect over all streams at once: n_det and every detection record, global stream ids included --
    with a ragged last block, one block only, more blocks than streams; pageable and page-locked host memory; on a device-pointer
    context too (the entry point takes host arrays whatever the context's flag says)."""
    import torch
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    templates = list(w["samples_features"].values())
    base = simstream.simulation_stream_i16()
    n = (len(base) // 480) * 480
    rng = np.random.default_rng(3)
    streams = [np.roll(base[:n], 480 * int(rng.integers(0, 40))) for _ in range(21)]
    pcm = np.stack(streams)
    if dtype is np.float32:
        pcm = simstream.i16_to_f32(pcm)
    cfg = ra.DetectorConfig()
    for gate in (0.0, 0.2):
        cfg.avg_threshold = gate
        tm = ra.Templates(ctx, templates, avg=w["avg_features"])
        det1, n1 = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        assert n1.sum() >= 21
        for block in (8, 21, 64, 1):
            det, n_det, sec = ctx.batch_detect_ingest(pcm, tm, cfg, max_det=4, block_streams=block)
            assert np.array_equal(n_det, n1) and det.tobytes() == det1.tobytes() and sec > 0
        assert [int(d["stream"]) for s in range(21) for d in det[s][:n_det[s]]] == [s for s in range(21) for _ in range(n_det[s])]
    # page-locked memory, device-pointer context
    dctx = ra.BatchContext(device=0, host_pointers=False)
    tmd = ra.Templates(dctx, templates, avg=w["avg_features"])
    hp = torch.from_numpy(pcm).pin_memory()
    det_h = torch.zeros((21, 4, 6), dtype=torch.int32).pin_memory()
    n_h = torch.zeros((21,), dtype=torch.int32).pin_memory()
    dctx.batch_detect_ingest_ptr(hp.data_ptr(), 3 if dtype is np.float32 else 1, 21, n, n, tmd, cfg, det_h.data_ptr(), n_h.data_ptr(), 4, block_streams=5)
    assert np.array_equal(n_h.numpy(), n1) and det_h.numpy().tobytes() == det1.view(np.int32).reshape(21, 4, 6).tobytes()
    # nothing to do / bad arguments
    det0, n0, _ = ctx.batch_detect_ingest(pcm[:0], tm, cfg, max_det=4)
    assert det0.shape[0] == 0
