"""Does running the whole path for two halves of the streams on two HIP streams overlap the MFCC kernel of one half with
the DTW kernel of the other (different resource profiles)?  Prints ms per full pass for 1, 2 and 4 concurrent queues."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rustpotter_amd as ra
import bench

S, N, T, L, K = 65536, 64000, 8, 100, 5
dev = torch.device("cuda", 0)
ctx0 = ra.BatchContext(0, host_pointers=False)
ctx0.set_stream(torch.cuda.current_stream().cuda_stream)
templates = bench.make_templates(ra, ctx0, torch, dev, [L] * T, K)
pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
ctx0.synth_dev(bench.SEED, 0, S, N, N, pcm.data_ptr())
nf = ra.mfcc_num_frames(N); n_win = nf - L + 1
cfg = ra.DetectorConfig(); cfg.avg_threshold = 0.0
for parts in (1, 2, 4, 8):
    ctxs = [ra.BatchContext(0, host_pointers=False) for _ in range(parts)]   # each has its own non-blocking stream
    tms = [ra.Templates(c, templates) for c in ctxs]
    Sp = S // parts
    outs = [(torch.empty((Sp, n_win, T), dtype=torch.float32, device=dev), torch.empty((Sp, n_win), dtype=torch.float32, device=dev),
             torch.zeros((Sp, 4, 6), dtype=torch.int32, device=dev), torch.zeros((Sp,), dtype=torch.int32, device=dev)) for _ in range(parts)]
    def step():
        for p in range(parts):
            sc, ag, det, nd = outs[p]
            ctxs[p].batch_detect_dev(pcm[p * Sp:(p + 1) * Sp].data_ptr(), Sp, N, N, tms[p], cfg, det.data_ptr(), nd.data_ptr(), 4, sc.data_ptr(), ag.data_ptr())
    def sync():
        for c in ctxs: c.synchronize()
        torch.cuda.synchronize()
    for _ in range(3): step()
    sync()
    t0 = time.perf_counter()
    for _ in range(10): step()
    sync()
    print("queues %d: %.3f ms per pass" % (parts, (time.perf_counter() - t0) / 10 * 1e3))
    del ctxs, tms, outs
