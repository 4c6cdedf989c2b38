import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(device=0, host_pointers=True)
for dims in [(3120, 32, 16, 2), (3120, 13, 2), (1040, 32, 16, 3), (64, 13, 2), (4096, 20, 255, 4), (3120, 80, 40, 3), (320, 32, 16, 2)]:
    rng = np.random.default_rng(sum(dims))
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
    model = ra.Model(ctx, ws, bs)
    for scale in (1.0, 10.0, 30.0):
        x = (rng.standard_normal((4096, dims[0])) * scale).astype(np.float32)
        ref = orc.mlp_forward(x, ws, bs).astype(np.float64)
        out = []
        for prec in ("f32", "f32_strict"):
            got = ctx.mlp_forward(x, model, precision=prec).astype(np.float64)
            d = np.abs(got - ref)
            out.append("%s: max abs %.2e, max d/(1e-5+1e-5|ref|) %.2f, |ref| max %.1f" % (prec, d.max(), (d / (1e-5 + 1e-5 * np.abs(ref))).max(), np.abs(ref).max()))
        print(dims, "x scale", scale, "|", " | ".join(out), "|", ctx.last_mlp_kernel()[:30])
