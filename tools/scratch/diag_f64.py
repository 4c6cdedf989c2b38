import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rustpotter_amd as ra
from oracle import rp_oracle as orc
import test_gpu_mfcc_f64 as T
ctx = ra.BatchContext(device=0, host_pointers=True)
def stats(sig, K, framescale):
    got, ref, tru = [], [], []
    for x in sig:
        x = np.ascontiguousarray(x, np.float32)
        got.append(ctx.mfcc(x[None, :], K)[0].astype(np.float64)); ref.append(orc.mfcc_stream(x, K).astype(np.float64)); tru.append(T.f64_mfcc(x, K))
    got, ref, tru = np.concatenate(got), np.concatenate(ref), np.concatenate(tru)
    ek, eo = np.abs(got - tru), np.abs(ref - tru)
    rk, ro = np.sqrt((ek ** 2).mean(axis=0)), np.sqrt((eo ** 2).mean(axis=0))
    strict = 1e-5 * np.maximum(np.abs(ref), 1.0)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0) if framescale else np.maximum(np.abs(ref), 1.0)
    beyond = np.abs(got - ref) > strict
    b = (ek / np.broadcast_to(eo.max(axis=0), ek.shape))[beyond]
    return dict(K=K, rms_ratio_max=float((rk/ro).max()), rms_ratio_all=float(np.sqrt((ek**2).mean())/np.sqrt((eo**2).mean())), max_ratio=float(ek.max()/eo.max()),
                loose=float((np.abs(got-ref)/scale).max()), n_beyond=int(beyond.sum()), beyond_ratio=float(b.max()) if b.size else 0.0, eo_max=float(eo.max()), ek_max=float(ek.max()))
SEED = T.SEED
for K in (16, 23, 40):
    print("noise", stats([orc.synth_pcm(SEED, s, 480 * 60) for s in range(6)], K, True))
for seed in (2, 5, 9):
    rng = np.random.default_rng(seed)
    print("tones", seed, stats([T._tones(rng, 480 * 40) * 10.0 ** rng.uniform(-1.5, 0.3) for _ in range(8)], 5, False))
for seed in (3, 6):
    rng = np.random.default_rng(seed)
    print("speech", seed, stats([T._utterance(rng, 480 * 40) * 10.0 ** rng.uniform(-1.5, 0.3) for _ in range(8)], 5, False))
print("noise5", stats([orc.synth_pcm(SEED, 40 + s, 480 * 60) for s in range(4)], 5, False))
