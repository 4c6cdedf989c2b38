import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rustpotter_amd as ra
from oracle import rp_oracle as orc
import test_gpu_round3 as t3
import rpw_py
SEED = 0x5EED000000000001
G = t3.G
ctx = ra.BatchContext(0)
K = 16
m = rpw_py.load_rpw(os.path.join(G, "ok_casa-tiny.rpw"))
model = ra.Model(ctx, [m["weights"]["ln1.weight"], m["weights"]["ln2.weight"]], [m["weights"]["ln1.bias"], m["weights"]["ln2.bias"]])
none_index = m["labels"].index("none")
x48, sr, _ = rpw_py.read_wav(os.path.join(G, "ok_casa.wav"))
speech = orc.resample_stream(x48, sr)
rng = np.random.default_rng(5)
n = 480 * 500
cfg = ra.DetectorConfig(); cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.2, 0.6, 1
for a in (20000, 90000):
    st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
    st[a:a + len(speech)] += speech
    det, lab, nd = ctx.batch_detect_model(st[None, :], model, K, none_index, cfg, max_det=16)
    print(a, nd[0], [(int(d["frame"]), round(float(d["score"]), 4), int(d["counter"])) for d in det[0][:nd[0]]])
for seed in (77, 78, 79):
    cand = orc.synth_pcm(SEED + seed, 3, 480 * 30) * np.float32(0.3)
    st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
    st[120000:120000 + len(cand)] += cand
    det, lab, nd = ctx.batch_detect_model(st[None, :], model, K, none_index, cfg, max_det=16)
    print("burst", seed, nd[0], [(int(d["frame"]), round(float(d["score"]), 4), int(d["counter"])) for d in det[0][:nd[0]]])
st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
det, lab, nd = ctx.batch_detect_model(st[None, :], model, K, none_index, cfg, max_det=16)
print("noise only", nd[0], [(int(d["frame"]), round(float(d["score"]), 4)) for d in det[0][:nd[0]]])
