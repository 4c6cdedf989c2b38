import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(0)
rng = np.random.default_rng(5)
for dt in (np.int32, np.int16, np.int8):
    for S, N in ((65, 11656), (1, 480 * 3), (3, 480 * 3 + 37)):
        for gain, bp in ((False, False), (True, False), (False, True)):
            x = rng.standard_normal((S, N)) * 0.1
            info = np.iinfo(dt)
            raw = np.clip(np.round(x * info.max), info.min, info.max).astype(dt)
            dec = raw.astype(np.float32) / np.float32({np.int16: 32767.0, np.int8: 127.0, np.int32: 2147483648.0}[dt])
            f = ra.FiltersConfig(); f.gain_normalizer.enabled = gain; f.band_pass.enabled = bp
            out, rms, gains = ctx.frontend(raw, f, 0.05, 5)
            ro, rr, rg = orc.frontend_stream(dec[0], gain_normalizer=gain, rms_level_ref=0.05, window_size=5, band_pass=bp)
            print(np.dtype(dt).name, S, N, gain, bp, 'rms', np.array_equal(rms[0], rr), 'gains', np.array_equal(gains[0], rg), 'out', np.array_equal(out[0], ro),
                  'first diff', (np.flatnonzero(out[0] != ro)[:3], out[0][out[0] != ro][:2], ro[out[0] != ro][:2]) if not np.array_equal(out[0], ro) else '')
