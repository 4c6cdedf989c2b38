"""The averaged-template gate behind dtw_generic_kernel (mfcc sizes / bands without a register kernel): a detect-only call skips what the
gate rules out and finds the detections of full scoring."""

import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


# ------------------------------------------------------------------ the averaged-template gate behind the generic DTW kernel
@pytest.mark.parametrize("K,band", [(7, 5), (5, 9), (12, 2)])
def test_avg_gate_skip_behind_the_generic_kernel(ra, K, band):
    """Template sets only dtw_generic_kernel serves (mfcc_size / band_size without a register kernel): a detect-only call skips
    the sample templates for every wave of 64 windows whose averaged-template scores are all below avg_threshold
    (wakeword_comp.rs:85-93).  Same detections, bit for bit, as RP_CTX_FULL_SCORES, at thresholds that reject nothing, about
    half, and almost everything; and the oracle's chunked detector agrees on chunk and counter."""
    rng = np.random.default_rng(K * 31 + band)
    L, n = 50, 480 * 260
    utt = (orc.synth_pcm(SEED + 5, 1, 480 * 20) * np.float32(0.4)).astype(np.float32)
    templates = []
    for i in range(3):
        v = utt + rng.standard_normal(len(utt)).astype(np.float32) * np.float32(0.004)
        m = orc.mfcc_stream(v, K)
        templates.append(np.ascontiguousarray((m - m.mean(axis=0, dtype=np.float32))[:L], np.float32))
    avg_t = np.ascontiguousarray(np.mean(templates, axis=0, dtype=np.float32), np.float32)
    streams = []
    for s in range(6):
        st = rng.standard_normal(n).astype(np.float32) * np.float32(0.003)
        if s % 2 == 0:
            p = 160 * (200 + 37 * s)
            st[p:p + len(utt)] += utt
        streams.append(st)
    pcm = np.stack(streams)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    tg, tf = ra.Templates(gated, templates, avg=avg_t), ra.Templates(full, templates, avg=avg_t)
    cfg = ra.DetectorConfig()
    cfg.band_size, cfg.threshold, cfg.min_scores = band, 0.45, 1
    some = False
    for avg_threshold in (0.05, 0.35, 0.6):
        cfg.avg_threshold = avg_threshold
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=4)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=4)
        assert np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes(), avg_threshold
        some = some or n_f.sum() > 0
        from collections import OrderedDict
        for s in (0, 1):
            d = orc.Detector(avg_threshold=avg_threshold, threshold=0.45, band_size=band, min_scores=1)
            d.add_ref({"name": "w", "samples_features": OrderedDict(("t%d" % i, t) for i, t in enumerate(templates)), "avg_features": avg_t,
                       "threshold": None, "avg_threshold": None, "rms_level": 0.0})
            want = [(i // 480, r) for i in range(0, n, 480) for r in [d.process_f32(pcm[s, i:i + 480])] if r is not None]
            assert len(want) == n_f[s]
            for j, (chunk, r) in enumerate(want):
                assert det_f[s][j]["frame"] // 3 + 1 == chunk and det_f[s][j]["counter"] == r["counter"]
    assert some
