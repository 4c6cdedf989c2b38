import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rustpotter_amd as ra
from oracle import rp_oracle as orc
os.environ["RP_MLP_STREAM"] = "2"
dims = (3120, 32, 16, 2)
rng = np.random.default_rng(9)
ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
dctx = ra.BatchContext(0, host_pointers=False)
dctx.set_stream(torch.cuda.current_stream().cuda_stream)
model = ra.Model(dctx, ws, bs)
for B in (300, 600):
    x = rng.standard_normal((B, dims[0])).astype(np.float32)
    ref = orc.mlp_forward(x, ws, bs)
    for off in (0, 4, 12, 16, 28):
        for pad in (float("nan"), 0.0, 1e30):
            for rep in range(2):
                buf = torch.full((off + B * dims[0] + 64,), pad, dtype=torch.float32, device="cuda")
                buf[off:off + B * dims[0]] = torch.from_numpy(x.reshape(-1)).cuda()
                out = torch.full((B, 2), 12345.0, dtype=torch.float32, device="cuda")
                dctx.mlp_dev(model, buf.data_ptr() + 4 * off, B, "f32", out.data_ptr())
                torch.cuda.synchronize()
                o = out.cpu().numpy()
                unwritten = np.where((o == 12345.0).any(axis=1))[0]
                nan = np.where(~np.isfinite(o).all(axis=1))[0]
                wrong = np.where(~np.isclose(o, ref, rtol=1e-4, atol=1e-4).all(axis=1))[0]
                print("B", B, "off", off, "pad", pad, "unwritten", len(unwritten), "nan", len(nan), "wrong", len(wrong), wrong[:8])
print("done")
