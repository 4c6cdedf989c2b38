"""Throughput of the batched model detector (rp_batch_detect_model): S synthetic 4 s streams, Small model shape of
BASELINE config C5 (F=195 frames x K=16)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustpotter_amd as ra

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
N, F, K = 64000, 195, 16
# MODEL_TYPE = tiny | small (default: BASELINE config C5) | medium | large: the layer widths of wakeword_nn.rs:305-389
mt = os.environ.get("MODEL_TYPE", "small")
dims = {"tiny": [F * K, F // 15, 2], "small": [F * K, F // 6, F // 12, 2], "medium": [F * K, F // 3, F // 6, 2], "large": [F * K, F // 3 * 2, F // 6, 2]}[mt]
rng = np.random.default_rng(5)
ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
dev = torch.device("cuda", 0)
ctx = ra.BatchContext(device=0, host_pointers=False)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
model = ra.Model(ctx, ws, bs)
pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
ctx.synth_dev(0x5EED000000000001, 0, S, N, N, pcm.data_ptr())
det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
lab = torch.zeros((S, 4), dtype=torch.int32, device=dev)
n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
cfg = ra.DetectorConfig()
cfg.avg_threshold = 0.0
import ctypes as C
L = ra.load_library()
c = cfg._c()
def step():
    r = L.rp_batch_detect_model(ctx._h, pcm.data_ptr(), 3, S, N, N, model._h, K, 0, C.byref(c), {"f32": 0, "bf16": 1}[prec],
                                det.data_ptr(), lab.data_ptr(), n_det.data_ptr(), 4)
    assert r == 0
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
n_win = (3 * (N // 480) - 3) - F + 1
print("%s model %s, %d streams, %s: %.2f ms per pass, %.1f M window scorings/s (%d windows per stream)" % (mt, "->".join(map(str, dims)), S, prec, dt * 1e3, S * n_win / dt / 1e6, n_win))
