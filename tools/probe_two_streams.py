"""Probe: does the headline step gain from two halves of the batch running concurrently on two HIP streams of one process (MFCC of one half
beside the DTW of the other)?  usage: python tools/probe_two_streams.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustpotter_amd as ra
from oracle import rp_oracle as orc
SEED, N, L, K, T = 0x5EED000000000001, 64000, 100, 5, 8
dev = torch.device("cuda", 0)
templates = orc.synth_templates(SEED, T, L, K)
nf = ra.mfcc_num_frames(N); n_win = nf - L + 1
cfg = ra.DetectorConfig(); cfg.avg_threshold = 0.0

class Half:
    def __init__(self, S, first):
        self.S = S
        self.stream = torch.cuda.Stream()
        self.ctx = ra.BatchContext(device=0, host_pointers=False)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.tm = ra.Templates(self.ctx, templates)
        self.pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
        self.ctx.synth_dev(SEED, first, S, N, N, self.pcm.data_ptr())
        self.scores = torch.empty((S, n_win, T), dtype=torch.float32, device=dev)
        self.agg = torch.empty((S, n_win), dtype=torch.float32, device=dev)
        self.det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
        self.n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
        self.ctx.synchronize()
    def step(self):
        self.ctx.batch_detect_dev(self.pcm.data_ptr(), self.S, N, N, self.tm, cfg, self.det.data_ptr(), self.n_det.data_ptr(), 4, self.scores.data_ptr(), self.agg.data_ptr())

def run(halves, steps):
    def work(h):
        for _ in range(steps): h.step()
        h.ctx.synchronize()
    for h in halves: h.step(); h.ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(h,)) for h in halves]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

one = Half(65536, 0)
t1 = run([one], 10)
del one; torch.cuda.empty_cache()
two = [Half(32768, 0), Half(32768, 32768)]
t2 = run(two, 10)
print("one stream, 65 536 streams per step: %.2f ms; two HIP streams x 32 768 concurrently: %.2f ms per step of both" % (t1 * 1e3, t2 * 1e3))
