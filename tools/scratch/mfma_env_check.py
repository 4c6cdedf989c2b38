import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rustpotter_amd as ra
from oracle import rp_oracle as orc
SEED = 0x5EED000000000001
ctx = ra.BatchContext(device=0, host_pointers=True)
K, L, T, S = 5, 100, 8, 3
templates = orc.synth_templates(SEED, T, L, K)
mf = np.stack([orc.mfcc_stream(orc.synth_pcm(SEED, s, 480 * 60), K) for s in range(S)])
tm = ra.Templates(ctx, templates)
a, _, _ = ctx.dtw_scores(mf, tm)
os.environ["RP_DTW_MFMA"] = "0"
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
print("getenv:", libc.getenv(b"RP_DTW_MFMA"))
b, _, _ = ctx.dtw_scores(mf, tm)
del os.environ["RP_DTW_MFMA"]
c, _, _ = ctx.dtw_scores(mf, tm)
print("shape", a.shape, "on vs off max rel", np.abs(a / b - 1).max(), "on vs on again", np.abs(a / c - 1).max())
ref, _ = orc.score_stream(mf[0], templates)
print("vs oracle: on", np.abs(a[0] / ref - 1).max(), "off", np.abs(b[0] / ref - 1).max())
