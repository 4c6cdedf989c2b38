"""Measures the resampler kernel against the oracle and against an f64 evaluation of the same linear map
(how much of the difference is the oracle's own f32 rounding)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpotter_amd as ra
from oracle import rp_oracle as orc

ctx = ra.BatchContext(device=0, host_pointers=True)
for fs in (48000, 44100, 22050, 8000):
    fi, fo = ra.resampler_frame_lengths(fs)
    rng = np.random.default_rng(fs)
    n = fi * 7
    t = np.arange(n)
    pcm = np.stack([rng.uniform(-0.5, 0.5, n), 0.3 * np.sin(2 * np.pi * 440.0 * t / fs) + 0.01 * rng.standard_normal(n),
                    np.where((t // 500) % 2 == 0, 0.25, -0.25)]).astype(np.float32)
    got = ctx.resample(pcm, fs)
    for s in range(3):
        ref = orc.resample_stream(pcm[s], fs)
        d = np.abs(got[s] - ref)
        print(fs, s, "max|d| %.3e  rms d %.3e  max|ref| %.3f" % (d.max(), np.sqrt((d * d).mean()), np.abs(ref).max()))
