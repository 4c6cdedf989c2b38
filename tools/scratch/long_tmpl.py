"""Probe: wakeword templates far longer than the usual 1-2 s (do the register DTW kernels get the LDS they need?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(0)
K = 5
for L in (1500, 2500, 5000, 9000):
    rng = np.random.default_rng(L)
    templates = [rng.standard_normal((L, K)).astype(np.float32), rng.standard_normal((L - 7, K)).astype(np.float32), rng.standard_normal((L, K)).astype(np.float32)]
    mf = rng.standard_normal((2, L + 40, K)).astype(np.float32)
    try:
        sc, _, ag = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
        ref, _ = orc.score_stream(mf[0][:L + 2], templates)
        print(L, "ok", sc.shape, float(np.abs(sc[0][:ref.shape[0]] / ref - 1).max()))
    except Exception as e:
        print(L, "FAILED:", e)
