import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import sweep_parity as sp
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(0)
case = sp.make_model_case(np.random.default_rng([202, 99, 874]))
c, m, x = case["cfg"], case["model"], case["x"]
K, L = m["mfcc_size"], m["train_size"]
nl = len([k for k in m["weights"] if k.endswith(".weight")])
ws = [m["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)]; bs = [m["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)]
model = ra.Model(ctx, ws, bs)
mf = ctx.mfcc(x[None, :], K)[0]
print("dims", [ws[0].shape[1]] + [w.shape[0] for w in ws], "frames", mf.shape)
wins = list(range(92, 102))
X = np.stack([(mf[w:w + L] - mf[w:w + L].mean(axis=0, dtype=np.float32)).reshape(-1) for w in wins]).astype(np.float32)
lg_gpu = ctx.mlp_forward(X, model)
lg_orc = orc.mlp_forward(X, ws, bs)
ref = 2.2
def score(l):
    b = int(np.argmax(l)); return b, 1 - 1 / (1 + np.exp(((l[b] - l[0]) - ref) / ref))
for w, a, b in zip(wins, lg_gpu, lg_orc):
    print(w, "explicit gpu", np.round(a, 4), score(a), "| oracle", np.round(b, 4), score(b))
print("mfcc mean |c| per coefficient of window 97:", np.round(mf[97:97 + L].mean(axis=0), 2))
