"""Measurement of rp_frontend_batch at BASELINE C3 size (i16 in, f32 out): wall time per call and, with KERNELS=1, the
device time of each of its kernels from HIP events around single launches is not available through the C ABI -- use
rocprofv3 --kernel-trace --stats -- python3 tools/bench_frontend.py for the split.  argv: [S] [gain 0/1] [band_pass 0/1] [fmt i16|f32]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustpotter_amd as ra
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
GAIN = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
FMT = sys.argv[4] if len(sys.argv) > 4 else "i16"
N = 64000
ctx = ra.BatchContext(0, host_pointers=False); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
if FMT == "i16":
    raw = (torch.randn((S, N), device="cuda") * 3000).to(torch.int16); code, width = 1, 2
else:
    raw = torch.randn((S, N), device="cuda") * 0.1; code, width = 3, 4
out = torch.empty((S, N), dtype=torch.float32, device="cuda")
rc = ra.RustpotterConfig(); rc.filters.gain_normalizer.enabled = bool(GAIN); rc.filters.band_pass.enabled = bool(BP)
f = rc._filters_c()
L = ra.load_library()
def run():
    assert L.rp_frontend_batch(ctx._h, raw.data_ptr(), code, S, N, N, C.byref(f), 0.05, 33, out.data_ptr(), N, None, None) == 0
run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("frontend S=%d %s gain=%d band_pass=%d: %.2f ms, %.2f G samples/s, %.2f TB/s (%s in + f32 out)" %
      (S, FMT, GAIN, BP, dt * 1e3, S * N / dt / 1e9, S * N * (width + 4) / dt / 1e12, FMT))
