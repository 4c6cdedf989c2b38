"""One-off measurement of rp_frontend_batch at BASELINE C3 size (i16 in, f32 out)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustpotter_amd as ra
from rustpotter_amd.api import _FiltersCfg
S, N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 64000
ctx = ra.BatchContext(0, host_pointers=False); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
raw = (torch.randn((S, N), device="cuda") * 3000).to(torch.int16)
out = torch.empty((S, N), dtype=torch.float32, device="cuda")
rc = ra.RustpotterConfig(); rc.filters.gain_normalizer.enabled = True; rc.filters.band_pass.enabled = True
f = rc._filters_c()
L = ra.load_library()
def run():
    assert L.rp_frontend_batch(ctx._h, raw.data_ptr(), 1, S, N, N, C.byref(f), 0.05, 33, out.data_ptr(), N, None, None) == 0
run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("frontend S=%d: %.2f ms, %.2f G samples/s, %.2f TB/s (i16 in + f32 out)" % (S, dt * 1e3, S * N / dt / 1e9, S * N * 6 / dt / 1e12))
