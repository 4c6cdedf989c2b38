"""Single-stream drop-in API latency per 30 ms chunk (what tests/test_gpu_parity.py::test_single_stream_call_latency times)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpotter_amd as ra
from oracle import rp_oracle as orc

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
c = ra.RustpotterConfig.default()
c.detector.avg_threshold = float(os.environ.get("AVG", "0.0"))
rp = ra.Rustpotter.new(c)
rp.add_wakeword_from_file("w", os.path.join(G, "oye_casa_g.rpw"))
pcm = orc.synth_pcm(0x5EED000000000001, 5, 480 * 700) * np.float32(0.1)
if os.environ.get("SIGNAL", "noise") == "silence":    # digital silence: every window is rejected by the averaged-template gate
    pcm = np.zeros_like(pcm)
elif os.environ.get("SIGNAL") == "hum":                # a steady tone: rejected as well (avg_score ~0.1)
    pcm = (0.05 * np.sin(2 * np.pi * 120.0 * np.arange(len(pcm)) / 16000.0)).astype(np.float32)
chunks = [pcm[i:i + 480].copy() for i in range(0, len(pcm), 480)]
for ch in chunks[:150]:
    rp.process_samples(ch)
lat = []
for ch in chunks[150:]:
    t0 = time.perf_counter()
    rp.process_samples(ch)
    lat.append(time.perf_counter() - t0)
lat = np.array(lat) * 1e6
print("per call: median %.1f us, mean %.1f, p99 %.1f, min %.1f" % (np.median(lat), lat.mean(), np.percentile(lat, 99), lat.min()))
