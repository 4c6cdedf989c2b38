"""GPU box, with a library built with -DRP_MFMA_TRACE (tools/ab.sh -x): where does a short dtw_mfma_kernel launch spend its time?
Runs the DTW stage alone on `rounds` x 3 072 tiles and prints the phase boundaries (constant 100 MHz clock, us) of wave 0 and the last
wave of the first 96 workgroups: kernel entry, A image staged, first tile's frames + means ready, first tile's columns done, first tile
written, all tiles of the wave done, workgroup done."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustpotter_amd as ra
from oracle import rp_oracle as orc
SEED = 0x5EED000000000001
L, T, K = 100, 8, 5
lib = ra.load_library()
for rounds in (1.0, 2.0, 3.09):
    S = int(round(rounds * 3072 * 32 / 297))
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(SEED, T, L, K)
    tm = ra.Templates(ctx, templates)
    N = 64000
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    mf = torch.empty((S * nf * K + 1024,), dtype=torch.float32, device="cuda")
    ctx.mfcc_dev(pcm.data_ptr(), S, N, N, K, mf.data_ptr())
    sc = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    ag = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    for _ in range(5):
        ctx.dtw_dev(mf.data_ptr(), S, nf, tm, 0.22, 5, 1, 0, sc.data_ptr(), None, ag.data_ptr())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ctx.dtw_dev(mf.data_ptr(), S, nf, tm, 0.22, 5, 1, 0, sc.data_ptr(), None, ag.data_ptr())
    b.record()
    torch.cuda.synchronize()
    words = 1024 + 96 * 2 * 8 * 2   # the stamps end where the counter block ends (2 * kDtwSchedChunks words): DtwWork::fix behind it is not touched
    buf = (C.c_uint32 * words)()
    lib.rp_debug_read_dtw_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.rp_debug_read_dtw_work(ctx._h, buf, words) == 0
    t = np.frombuffer(buf, dtype=np.uint32)[1024:].view(np.uint64).reshape(96, 2, 8).astype(np.float64) / 100.0  # us
    t0 = t[:, :, 0].min()
    t = t - t0
    names = ["entry", "A staged", "tile1 frames+means", "tile1 columns", "tile1 written", "all tiles", "workgroup done"]
    print("rounds %.2f: %d streams, %d tiles; events around the call (dtw + aggregate + list pass): %.1f us" % (rounds, S, S * n_win // 32, a.elapsed_time(b) * 1e3))
    for i, nme in enumerate(names):
        col = t[:, :, i]
        print("   %-20s wave 0: median %7.1f  min %7.1f  max %7.1f | last wave: median %7.1f  max %7.1f" %
              (nme, np.median(col[:, 0]), col[:, 0].min(), col[:, 0].max(), np.median(col[:, 1]), col[:, 1].max()))
    del ctx
