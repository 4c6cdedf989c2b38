"""CPU restatement timings beside the GPU numbers: 1 thread and all granted cores on a slice of the
synthetic C3 set, and the reference's own fixture stream (C1) through the chunked detector."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import rp_oracle as orc
import rpw_py, simstream

SEED = 0x5EED000000000001
tmpl = orc.synth_templates(SEED, 8, 100, 5)
cores = len(os.sched_getaffinity(0))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        cores = max(1, min(cores, int(int(q) / int(p))))
except Exception:
    pass
for threads, S in ((1, 48), (cores, 48 * cores)):
    secs, sc, _ = orc.bench(SEED, S, 64000, tmpl, threads=threads)
    print("synthetic C3 slice: %d streams, %d thread(s): %.0f scorings/s (%.2f s)" % (S, threads, sc / secs, secs))
G = simstream.GOLDEN
w = rpw_py.load_rpw(os.path.join(G, "alexa.rpw"))
d = orc.Detector(avg_threshold=0.0, threshold=0.45, min_scores=0)
d.add_ref(w)
s = simstream.simulation_stream_i16()
t0 = time.perf_counter()
n = 0
for i in range(0, len(s) - 479, 480):
    d.process_i16(s[i:i + 480]); n += 1
dt = time.perf_counter() - t0
print("C1 fixture stream x alexa.rpw (T=3, L=126): %d chunks in %.3f s = %.0f us per 30 ms chunk, %.0f frame scorings/s, 1 thread" % (n, dt, dt / n * 1e6, 3 * n / dt))
