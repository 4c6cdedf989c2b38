"""Pins the CPU oracle (oracle/rp_oracle.c) to the reference's own golden data.

G1: MFCC matrices inside tests/resources/*.rpw (written by tests/wakeword.rs:27-54)
G2: exact f32 detections asserted in tests/detector.rs:9-162
G3: avg_features inside the .rpw files (MfccAverager, src/mfcc/averager.rs)
G5: NN score formula values, tests/detector.rs:216-267
"""
import json
import os

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

G = simstream.GOLDEN
EXP = json.load(open(os.path.join(G, "expectations.json")))


def ulps(a, b):
    a, b = np.float32(a), np.float32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32)))


@pytest.mark.parametrize("rpw", ["oye_casa_g.rpw", "alexa.rpw"])
def test_g1_mfcc_matches_reference_rpw(rpw):
    w = rpw_py.load_rpw(os.path.join(G, rpw))
    assert w["kind"] == "ref" and w["mfcc_size"] == 5
    for name, ref in w["samples_features"].items():
        assert ref.shape == (EXP["rpw_shapes"][rpw][name], 5)
        pcm, sr = rpw_py.read_wav_i16(os.path.join(G, name))
        assert sr == 16000
        got = orc.normalize(orc.mfcc_stream(simstream.i16_to_f32(pcm), 5))
        assert got.shape == ref.shape
        # SURVEY §8(d) gate: |d| <= 1e-5 * max(|ref|, 1); measured: <= 1e-5 ABSOLUTE (|ref| up to 34)
        assert np.all(np.abs(got - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0))
        assert np.abs(got - ref).max() <= 1.2e-5


def test_g1_v2_file_holds_same_templates():
    a = rpw_py.load_rpw(os.path.join(G, "oye_casa_g.rpw"))
    b = rpw_py.load_rpw(os.path.join(G, "oye_casa_g_v2.rpw"))
    assert b["kind"] == "v2" and set(a["samples_features"]) == set(b["samples_features"])
    for k in a["samples_features"]:
        assert np.array_equal(a["samples_features"][k], b["samples_features"][k])


@pytest.mark.parametrize("rpw", ["oye_casa_g.rpw", "alexa.rpw"])
def test_g3_avg_features_bit_exact(rpw):
    """Unbanded DTW + back-trace + cosine distance are bit-exact: the reference's own
    avg_features are reproduced from its own samples_features with 0 ulp error."""
    w = rpw_py.load_rpw(os.path.join(G, rpw))
    avg = orc.average_templates(w["samples_features"])
    assert avg.shape == w["avg_features"].shape
    assert np.array_equal(avg, w["avg_features"])


def _run_sim(e):
    w = rpw_py.load_rpw(os.path.join(G, e["rpw"]))
    d = orc.Detector(avg_threshold=e["avg_threshold"], threshold=e["threshold"], min_scores=e.get("min_scores", 5),
                     score_mode=e["score_mode"], vad_mode=e.get("vad_mode"),
                     gain_normalizer=e.get("gain_normalizer", False), band_pass=e.get("band_pass", False),
                     low_cutoff=e.get("low_cutoff", 80.0), high_cutoff=e.get("high_cutoff", 400.0))
    d.add_ref(w)
    s = simstream.simulation_stream_i16(*e.get("gains", [1.0, 1.0]))
    assert len(s) == 274390
    dets = []
    for i in range(0, len(s) - 479, 480):
        r = d.process_i16(s[i:i + 480])
        if r is not None:
            dets.append(r)
    return dets


@pytest.mark.parametrize("case", sorted(EXP["simulation"].keys()))
def test_g2_detector_goldens(case):
    e = EXP["simulation"][case]
    dets = _run_sim(e)
    assert len(dets) == len(e["detections"])
    for got, (avg, score) in zip(dets, e["detections"]):
        # the reference asserts exact f32; the only non-restated arithmetic is rustfft's
        # rounding, which leaves at most 1 ulp on these values
        assert ulps(got["score"], score) <= 1, (got["score"], score)
        if avg is not None:
            assert ulps(got["avg_score"], avg) <= 1, (got["avg_score"], avg)


def test_g5_nn_score_formula():
    for key in ("nn_score_formula", "nn_score_formula_eager"):
        e = EXP[key]
        got = orc.calc_inverse_similarity(e["label_logit"], e["none_logit"], np.float32(e["score_ref"]) * np.float32(10))
        assert ulps(got, e["score"]) <= 1


def test_model_file_layout():
    m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
    assert m["kind"] == "model" and m["labels"] == ["none", "ok_casa"]
    assert m["train_size"] == 195 and m["mfcc_size"] == 16 and m["m_type"] == "Tiny"
    assert m["weights"]["ln1.weight"].shape == (13, 3120) and m["weights"]["ln2.weight"].shape == (2, 13)
    x = np.random.default_rng(0).standard_normal((3, 3120)).astype(np.float32)
    out = orc.mlp_forward(x, [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]],
                          [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]])
    h = np.maximum(x.astype(np.float64) @ m["weights"]["ln1.weight"].T.astype(np.float64) + m["weights"]["ln1.bias"], 0)
    ref = h @ m["weights"]["ln2.weight"].T.astype(np.float64) + m["weights"]["ln2.bias"]
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-4)


def test_filter_bank_and_frame_count():
    bank, cen = orc.mel_filter_bank(5)
    assert cen.tolist() == [0, 9, 22, 41, 68, 106, 161, 240]  # SURVEY §8(a5) [probe]
    assert bank.shape == (6, 240) and ((bank > 0).sum(axis=0) <= 2).all()
    assert orc.mfcc_stream(np.zeros(18150, np.float32), 5).shape[0] == 108  # 3*floor(N/480)-3
    assert orc.mfcc_stream(np.zeros(479, np.float32), 5).shape[0] == 0


def test_silence_normalises_to_exact_zero():
    """SURVEY §7 'Silence': identical frames -> normalised window exactly 0 -> every cell cost 1."""
    m = orc.mfcc_stream(np.zeros(480 * 60, np.float32), 5)
    n = orc.normalize(m[:100])
    assert np.all(n == 0.0)
    t = np.random.default_rng(1).standard_normal((100, 5)).astype(np.float32)
    # D[m-1][n] with unit cost per cell = m-1 steps along the diagonal + 1 => cost m... check via score
    cost = orc.dtw_banded(t, n)
    assert cost == 100.0
