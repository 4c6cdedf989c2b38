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
rng = np.random.default_rng(5)
utt = orc.synth_pcm(SEED + 77, 3, 480 * 30) * np.float32(0.3)
wavs = {}
for i in range(3):
    v = utt + rng.standard_normal(len(utt)).astype(np.float32) * np.float32(0.003)
    wavs["u%d.wav" % i] = t3._wav_i16((np.clip(v, -1, 1) * 32767).astype(np.int16))
rpw = ctx.build_wakeword_ref("utt", wavs, K)
n = 480 * 500
s = rng.standard_normal(n).astype(np.float32) * np.float32(0.002)
s[120000:120000 + len(utt)] += utt
c = ra.RustpotterConfig.default()
c.detector.avg_threshold, c.detector.threshold, c.detector.min_scores = 0.2, 0.55, 3
for both in (False, True):
    rp = ra.Rustpotter.new(c)
    rp.add_wakeword_from_buffer("utt", rpw)
    if both:
        rp.add_wakeword_from_file("model", os.path.join(G, "ok_casa-tiny.rpw"))
    got = []
    best = 0
    for i in range(0, n, 480):
        d = rp.process_samples(s[i:i + 480].copy())
        p = rp.get_partial_detection()
        if p is not None: best = max(best, p.score)
        if d is not None:
            got.append((i // 480, d.name, d.score, d.counter))
    print("both" if both else "ref only", got, "best partial", best)
