import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rustpotter_amd as ra
from oracle import rp_oracle as orc
import test_gpu_round3 as t3
import rpw_py
SEED = 0x5EED000000000001
ctx = ra.BatchContext(0)
K = 16
rng = np.random.default_rng(5)
utt = orc.synth_pcm(SEED + 77, 3, 480 * 30) * np.float32(0.3)
wavs = {}
for i in range(3):
    v = utt + rng.standard_normal(len(utt)).astype(np.float32) * np.float32(0.003)
    wavs["u%d.wav" % i] = t3._wav_i16((np.clip(v, -1, 1) * 32767).astype(np.int16))
rpw = ctx.build_wakeword_ref("utt", wavs, K)
open("/tmp/w.rpw", "wb").write(rpw)
ref = rpw_py.load_rpw("/tmp/w.rpw")
tm = ra.Templates(ctx, list(ref["samples_features"].values()), avg=ref["avg_features"])
n = 480 * 500
s = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
s[120000:120000 + len(utt)] += utt
mf = ctx.mfcc(s[None, :], K)
scores, avg, agg = ctx.dtw_scores(mf, tm, with_avg=True) if "with_avg" in ctx.dtw_scores.__code__.co_varnames else ctx.dtw_scores(mf, tm)
print("templates", [v.shape for v in ref["samples_features"].values()], "max agg", agg.max(), "argmax", agg.argmax(), "avg max", None if avg is None else avg.max())
print(np.sort(agg.ravel())[-10:])
