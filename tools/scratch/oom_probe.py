import sys, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(0, host_pointers=False)
tm = ra.Templates(ctx, orc.synth_templates(1, 8, 100, 5))
cfg = ra.RustpotterConfig.default().detector
L = ra.load_library()
for S in (1 << 34, 1 << 40):
    try:
        sb = ra.StreamBatch(ctx, tm, cfg, S)
        print("StreamBatch", S, "created?!")
    except Exception as e:
        print("StreamBatch", S, "->", type(e).__name__, str(e)[:120])
# absurd batch sizes through the C ABI with a null-ish device pointer: must fail with an error, not crash
h = C.c_void_p()
out = C.c_void_p(0x1000)
r = L.rp_mfcc_batch(ctx._h, C.c_void_p(0x1000), C.c_size_t(1 << 40), C.c_size_t(64000), C.c_size_t(64000), 5, out)
print("rp_mfcc_batch huge S ->", r, L.rp_last_error())
r = L.rp_mfcc_batch(ctx._h, C.c_void_p(0x1000), C.c_size_t(4), C.c_size_t(1 << 62), C.c_size_t(1 << 62), 5, out)
print("rp_mfcc_batch huge N ->", r, L.rp_last_error())
print("alive")
