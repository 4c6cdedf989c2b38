"""Scratch probe: error statistics of the HIP MFCC against the oracle and against an f64 DFT."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpotter_amd as ra
from oracle import rp_oracle as orc

SEED = 0x5EED000000000001
ctx = ra.BatchContext(0, True)

def f64_mfcc(pcm, K):
    """double-precision evaluation of the same pipeline (tables from the oracle in f32)"""
    ham = orc.hamming_window().astype(np.float64)
    fb, _ = orc.mel_filter_bank(K)
    dct = orc.dct_table(K).astype(np.float64)
    nch = len(pcm) // 480
    x = pcm[: nch * 480].astype(np.float64).reshape(-1, 160)
    pre = x.copy()
    pre[:, 1:] = x[:, 1:] - np.float64(np.float32(0.97)) * x[:, :-1]
    pre = np.float32(pre).astype(np.float64)   # the reference rounds pre-emphasis to f32
    pre = pre.reshape(-1)
    out = []
    for j in range(3 * nch - 3):
        fr = pre[(j + 1) * 160:(j + 4) * 160] * ham
        X = np.fft.fft(fr)[:240]
        P = np.abs(X) ** 2
        lg = np.log(fb.astype(np.float64) @ P + np.finfo(np.float32).tiny)
        out.append(2 * (dct @ lg)[1:])
    return np.array(out)

for K in (5, 16, 23):
    worst_o = worst_t = worst_ot = worst_f = 0
    for s in range(4):
        pcm = orc.synth_pcm(SEED, s, 480 * 40)
        got = ctx.mfcc(pcm, K)[0]
        ref = orc.mfcc_stream(pcm, K)
        tru = f64_mfcc(pcm, K)
        worst_o = max(worst_o, (np.abs(got - ref) / np.maximum(np.abs(ref), 1)).max())
        worst_f = max(worst_f, (np.abs(got - ref) / np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1)).max())
        worst_t = max(worst_t, np.abs(got - tru).max())
        worst_ot = max(worst_ot, np.abs(ref - tru).max())
    print("K=%d  gpu-vs-oracle elementwise %.2e  framescale %.2e | abs err vs f64: gpu %.2e oracle %.2e" % (K, worst_o, worst_f, worst_t, worst_ot))
