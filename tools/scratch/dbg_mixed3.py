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
rng = np.random.default_rng(5)
n = 480 * 500
L = 480 * 30
t = np.arange(L) / 16000.0
kinds = {
 "chirp": 0.3 * np.sin(2 * np.pi * (200 * t + 0.5 * 3000 * t * t)),
 "quiet_burst": orc.synth_pcm(SEED + 77, 3, L) * 0.012,
 "tones": 0.1 * (np.sin(2*np.pi*300*t) + np.sin(2*np.pi*900*t) + np.sin(2*np.pi*2100*t)),
 "am_noise": orc.synth_pcm(SEED + 78, 3, L) * 0.3 * (0.5 + 0.5 * np.sin(2*np.pi*4*t)),
 "steps": 0.2 * np.sign(np.sin(2*np.pi*(100 + 50*np.floor(t*10))*t)),
 "lowpass": np.convolve(orc.synth_pcm(SEED + 79, 3, L) * 0.5, np.ones(24)/24, mode="same"),
}
cfg = ra.DetectorConfig(); cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.2, 0.55, 1
for name, u in kinds.items():
    u = u.astype(np.float32)
    st = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
    st[120000:120000 + L] += u
    _, _, nm = ctx.batch_detect_model(st[None, :], model, K, none_index, cfg)
    wavs = {}
    for i in range(3):
        v = u + rng.standard_normal(L).astype(np.float32) * np.float32(0.003)
        wavs["u%d.wav" % i] = t3._wav_i16((np.clip(v, -1, 1) * 32767).astype(np.int16))
    rpw = ctx.build_wakeword_ref("utt", wavs, K)
    open("/tmp/w.rpw", "wb").write(rpw)
    ref = rpw_py.load_rpw("/tmp/w.rpw")
    tm = ra.Templates(ctx, list(ref["samples_features"].values()), avg=ref["avg_features"])
    cfg3 = ra.DetectorConfig(); cfg3.avg_threshold, cfg3.threshold, cfg3.min_scores = 0.2, 0.55, 3
    det, nr = ctx.batch_detect(st[None, :], tm, cfg3)
    print(name, "model windows:", nm[0], "ref detections:", nr[0], [float(d["score"]) for d in det[0][:nr[0]]])
