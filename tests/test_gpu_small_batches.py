"""Small batches: the per-stream "can fire" flags between the aggregate pass and the scan (quiet streams skip the scan), and the tc-4 split of
small DTW launches (same scores as the tc-8 launch)."""
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = simstream.GOLDEN
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


# ------------------------------------------------------------------ small batches: hot flags + tc-4 split
def test_quiet_streams_skip_the_scan_and_loud_ones_do_not(ra, ctx):
    """The aggregate pass raises a flag per stream that has a window above the threshold; scan_kernel returns at once for the
    others.  Mixed batch: quiet noise streams between streams that hold the utterance -- detections equal a batch of the
    loud streams alone and the oracle's chunked detector; n_det of the quiet ones is 0 and their slots are zero."""
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    n = (len(base) // 480) * 480
    rng = np.random.default_rng(3)
    quiet = [rng.standard_normal(n).astype(np.float32) * np.float32(0.01) for _ in range(5)]
    loud = [base[:n], np.roll(base[:n], 480 * 7)]
    pcm = np.stack([quiet[0], loud[0], quiet[1], quiet[2], loud[1], quiet[3], quiet[4]])
    w = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    tm = ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])
    for avg_threshold in (0.0, 0.2):
        cfg = ra.DetectorConfig()
        cfg.threshold, cfg.avg_threshold = 0.45, avg_threshold
        det, n_det = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        det_l, n_l = ctx.batch_detect(np.stack(loud), tm, cfg, max_det=4)
        assert list(n_det) == [0, n_l[0], 0, 0, n_l[1], 0, 0] and n_l.min() >= 1
        for s, ls in ((1, 0), (4, 1)):
            for j in range(n_det[s]):
                assert all(det[s][j][f] == det_l[ls][j][f] for f in ("frame", "window", "counter", "score", "avg_score")) and det[s][j]["stream"] == s
        for s in (0, 2, 3, 5, 6):
            assert det[s].tobytes() == bytes(det[s].nbytes)
        # and with the per-window arrays requested (every window scored): same detections
        det2, n2, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
        assert np.array_equal(n2, n_det) and det2.tobytes() == det.tobytes()


@pytest.mark.parametrize("S", [64, 700, 1024])
def test_small_batches_tc4_split_gives_the_same_scores(ra, S):
    """A batch whose tc-8 DTW waves would fill the chip less than three times is scored by tc-4 half chunks instead
    (launch_dtw_k5).  Same operations per cell, so the scores are bit-identical to the tc-8 launch (RP_DTW_NO_SPLIT=1 in a
    child process) and within 1e-5 of the oracle."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import rustpotter_amd as ra
from oracle import rp_oracle as orc
S, SEED = %d, 0x5EED000000000001
ctx = ra.BatchContext(0)
templates = orc.synth_templates(SEED, 8, 60, 5)
pcm = ctx.synth_pcm(SEED, 0, S, 480 * 50)
mf = ctx.mfcc(pcm, 5)
scores, _, agg = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
np.save(sys.argv[1], scores)
""" % (ROOT, os.path.join(ROOT, "tests"), S)
    import tempfile
    outs = []
    for env_extra in ({}, {"RP_DTW_NO_SPLIT": "1"}):
        with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as f:
            path = f.name
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.unlink(path)
    assert outs[0].tobytes() == outs[1].tobytes()
    templates = orc.synth_templates(SEED, 8, 60, 5)
    for s in (0, S // 2, S - 1):
        ref_s, _ = orc.score_stream(orc.mfcc_stream(orc.synth_pcm(SEED, s, 480 * 50), 5), templates)
        assert np.all(np.abs(outs[0][s] - ref_s) <= 1e-5 * np.abs(ref_s))


def test_eight_and_twelve_wave_builds_of_the_three_part_kernel_give_the_same_bits(tmp_path):
    """dtw_mfma_kernel<5, 8 | 12, false, 8, true>: two builds of one arithmetic (219 registers, two waves per SIMD: the default; 168 registers,
    three per SIMD: RP_MFMA3_WAVES=12).  The scores of the same batch must not differ by a bit (child processes: the variable is read once)."""
    import subprocess
    import sys
    child = r"""
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np
import rustpotter_amd as ra
from oracle import rp_oracle as orc
SEED = 0x5EED000000000001
ctx = ra.BatchContext(device=0, host_pointers=True)
templates = orc.synth_templates(SEED + 9, 8, 60, 5)
pcm = ctx.synth_pcm(SEED, 500, 96, 480 * 70)
cfg = ra.DetectorConfig(); cfg.avg_threshold = 0.0
ctx.dtw_kernels()
_, _, scores, agg = ctx.batch_detect(pcm, ra.Templates(ctx, templates), cfg, want_scores=True)
assert ctx.dtw_kernels() == ["dtw_mfma_kernel"] and ctx.last_dtw_products == ["bf16x3"]
print("sha", hashlib.sha256(scores.tobytes() + agg.tobytes()).hexdigest(), scores.shape)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "waves_child.py"
    script.write_text(child)
    out = []
    for nw in ("8", "12"):
        e = dict(os.environ)
        e["RP_MFMA3_WAVES"] = nw
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("sha")][-1])
    assert out[0] == out[1], out
