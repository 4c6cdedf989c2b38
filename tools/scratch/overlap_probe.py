#!/usr/bin/env python3
"""Does the MFCC kernel of one block of streams run UNDER the DTW kernel of another (two HIP streams, two contexts)?
N iterations of rp_mfcc_batch on stream 1 and N of rp_dtw_score_batch on stream 2, alone and together; S streams each.
  python tools/scratch/overlap_probe.py [S] [N]        (RP_LIB_PATH / RP_MFMA3_WAVES select the build under test)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rustpotter_amd as ra
from oracle import rp_oracle as orc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
SEED, NS, K, T, L = 0x5EED000000000001, 64000, 5, 8, 100
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cm, cd = ra.BatchContext(device=0, host_pointers=False), ra.BatchContext(device=0, host_pointers=False)
cm.set_stream(s1.cuda_stream); cd.set_stream(s2.cuda_stream)
templates = orc.synth_templates(SEED, T, L, K)
tm = ra.Templates(cd, templates)
nf = ra.mfcc_num_frames(NS); n_win = nf - L + 1
pcm = torch.empty((S, NS), dtype=torch.float32, device=dev)
cm.synth_dev(SEED, 0, S, NS, NS, pcm.data_ptr())
mf_a = torch.empty((S, nf, K), dtype=torch.float32, device=dev)      # written by the MFCC stream
mf_b = torch.empty((S + 1, nf, K), dtype=torch.float32, device=dev)  # read by the DTW stream (slack behind the last stream)
cd.set_stream(s1.cuda_stream); cm.mfcc_dev(pcm.data_ptr(), S, NS, NS, K, mf_b.data_ptr()); torch.cuda.synchronize(); cd.set_stream(s2.cuda_stream)
scores = torch.empty((S, n_win, T), dtype=torch.float32, device=dev)
agg = torch.empty((S, n_win), dtype=torch.float32, device=dev)

def mfcc(): cm.mfcc_dev(pcm.data_ptr(), S, NS, NS, K, mf_a.data_ptr())
def dtw(): cd.dtw_dev(mf_b.data_ptr(), S, nf, tm, 0.22, 5, 1, False, scores.data_ptr(), 0, agg.data_ptr())

def timed(fns, n):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(3):
    tm_, td_ = timed([mfcc], N), timed([dtw], N)
    tb1, tb2 = timed([mfcc, dtw], N), timed([dtw, mfcc], N)
    print("rep %d  S %d: mfcc alone %.3f ms  dtw alone %.3f ms  sum %.3f | both, mfcc enqueued first %.3f  dtw first %.3f  (kernels ran: %s)" % (
        rep, S, tm_, td_, tm_ + td_, tb1, tb2, cd.dtw_kernels()))
