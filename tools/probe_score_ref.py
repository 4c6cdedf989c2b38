"""GPU box: largest relative score error of the DTW kernels against the oracle as a function of score_ref (the relative error of a
score grows like (1 - s) * d(nc) / score_ref: config.rs:172-209 lets a caller choose any value).  Prints one line per (shape, score_ref):
matrix-core kernel and register kernel (RP_DTW_MFMA=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpotter_amd as ra
from oracle import rp_oracle as orc
SEED = 0x5EED000000000001
ctx = ra.BatchContext(device=0, host_pointers=True)
S = int(os.environ.get("PROBE_STREAMS", "12"))
SHAPES = ((5, 8, 100, 5), (5, 4, 100, 5), (5, 8, 60, 3), (16, 8, 60, 5))
if os.environ.get("PROBE_SHAPES"):
    SHAPES = SHAPES[:int(os.environ["PROBE_SHAPES"])]
REFS = [float(x) for x in os.environ.get("PROBE_REFS", "0.22,0.15,0.1,0.07,0.05,0.03").split(",")]
for K, T, L, band in SHAPES:
    n_win = 120
    templates = orc.synth_templates(SEED + 7 * K + L, T, L, K)
    n = 480 * ((n_win + L - 1) // 3 + 2)
    mf = np.stack([orc.mfcc_stream(orc.synth_pcm(SEED, 900 + s, n), K)[:n_win + L - 1] for s in range(S)])
    tm = ra.Templates(ctx, templates)
    for ref in REFS:
        oracle = np.stack([orc.score_stream(mf[s], templates, band=band, score_ref=ref)[0] for s in range(S)]).astype(np.float64)
        res = []
        for env in (None, "0"):
            if env is None: os.environ.pop("RP_DTW_MFMA", None)
            else: os.environ["RP_DTW_MFMA"] = env
            sc, _, _ = ctx.dtw_scores(mf, tm, score_ref=ref, band_size=band)
            sgn = (sc - oracle) / np.maximum(oracle, 1e-300)
            rel = np.abs(sgn)
            res.append((rel.max(), np.sqrt((rel ** 2).mean()), sgn.mean()))
        os.environ.pop("RP_DTW_MFMA", None)
        print("K=%d T=%d L=%d band=%d score_ref=%.2f  matrix max %.2e rms %.2e mean %+.2e | register max %.2e rms %.2e mean %+.2e | min score %.3g" %
              (K, T, L, band, ref, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2], oracle.min()), flush=True)
