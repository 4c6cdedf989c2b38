#!/usr/bin/env python3
"""bench.py -- 10 ms-frame MFCC+DTW scorings/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (MFCC of every 10 ms frame, T banded DTWs +
score_mode aggregation per window, detection state machine) over S synthetic 16 kHz f32
streams that are already resident in HBM.  One "scoring" = one (stream, 10 ms frame)
step in steady state = one scored window.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over torch.distributed (backend nccl == RCCL on ROCm); streams are sharded by
rank (weak scaling: every GPU gets --streams streams), the only exchange is an RCCL all_gather of the
per-stream detection summary at the end of each step.  The ranks are either started by the caller
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is set) or, when
WORLD_SIZE is not set, by bench.py itself: the parent starts that same command as a child process BEFORE
anything touches the GPU, relays the child's output (rank 0 prints the one JSON line) and exits with its code.

What the one JSON line carries besides the contract's fields (round 4):
  roofline.frac      the largest of the dominant kernel's pipe fractions (executed work / pipe peak / measured time): VALU issue
                     cycles from the committed instruction mix of its hot loop (profiles/r04_*_isa_mix.json x
                     profiles/valu_rate_table.json), matrix flops, HBM bytes, LDS cycles -- each <= 1 by construction; the
                     reference-shaped flop rate of SURVEY 8d sits beside it as ref_flop_rate_vs_vector_peak
  vector_only        the same step with the cosine products on the vector pipe (RP_DTW_MFMA=0), 2 steps
  extra_configs      short timed runs of BASELINE configs C2, C5 (bf16) and C5 (f32), of the wakeword-model detector, of the filter front-end, of one GPU's
                     share of C4, of mfcc_size 16 and of five templates of unequal length in the same process
  h2d_included       a bounded sample of the same path with the PCM starting in pinned HOST memory (see --ingest)
  cpu_baseline       the oracle on all granted host cores, and on one thread (one_thread)
  config.*           host CPU model, build id of the library, and for N > 1 the world size RCCL reports and every rank's device UUID
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x5EED000000000001
HBM_PEAK = 8.0e12      # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
VALU_PEAK = 157.3e12   # FLOP/s fp32 vector, same table
MFMA_F16_PEAK = 2.5e15  # FLOP/s dense f16 / bf16 MFMA, same table
CLOCK_PEAK = 2.4e9     # Hz, peak shader clock, same table
# the arithmetic of the path: f32 throughout; the cosine products of the DTW cost are f32 vector FMAs ("strict_f32"), or formed on the matrix
# cores from exact three-part bf16 splits of both f32 operands with f32 accumulation ("f32_matrix", the library default: f32-grade, pinned by
# tests/test_gpu_dtw_f64.py), or -- opt-in, narrower than the reference -- from two-part f16 splits ("fast_split")
DTYPE_DTW = "f32"
DTYPE_DTW_FAST = "f32 (cosine products: f16x2-split MFMA, 22-bit -- RP_ARITH_FAST_SPLIT, narrower than the reference)"
DTYPE_DTW_VECTOR = "f32"
MIX_MFMA = {"bf16x3": "profiles/dtw_mfma_isa_mix.json", "f16x2": "profiles/dtw_mfma_f16x2_isa_mix.json"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process group.  Nothing in this
    process has initialised the GPU (torch.cuda.device_count() does not); the child is a separate program, not an exec."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if have < n:
        if os.environ.get("RP_BENCH_OVERSUBSCRIBE") != "1":
            sys.stderr.write("bench.py: --gpus %d but this node has %d GPU(s); set RP_BENCH_OVERSUBSCRIBE=1 for a dry run in which "
                             "ranks share devices over gloo (reported as oversubscribed, not a scaling number)\n" % (n, have))
            return 2
        env["RP_BENCH_BACKEND"] = "gloo"
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def host_cpu():
    """(threads this process may use -- affinity mask capped by a cgroup CPU quota --, CPU model string)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # honour a cgroup CPU quota (the GPU box grants 16 of its 256 hardware threads)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return cores, model


def load_json(rel):
    try:
        return json.load(open(os.path.join(ROOT, rel)))
    except Exception:
        return None


def n_simds(torch, dev):
    return 4 * torch.cuda.get_device_properties(dev).multi_processor_count


ROOFLINE_FIRST = ("bound", "kernel", "products", "arithmetic", "frac", "frac_at_architectural_rates", "valu_plus_matrix_issue_frac", "achieved", "peak", "unit",
                  "avg_launch_ms", "traffic", "traffic_over_algorithmic", "strict_f32_value", "strict_f32_ms_per_step", "strict_f32_steps", "strict_f32_dtw_ms",
                  "fast_split_value", "fast_split_ms_per_step", "path_hbm_frac", "path_ref_flop_rate_vs_vector_peak", "mfcc_ms", "dtw_ms", "i16_pcm_value",
                  "i16_pcm_mfcc_ms", "effective_clock_ghz", "valu_issue_frac_at_effective_clock", "strict_f32_mfcc_ms", "strict_f32_aggregate_ms", "fast_split_dtw_ms",
                  "aggregate_ms", "scan_ms", "algorithmic_bytes_per_launch", "ref_flop_rate_vs_vector_peak")


def order_for_the_record(res):
    """The driver's record of a line keeps a bounded number of scalar entries of `roofline` and `config` (round 4: 23 of roofline's,
    nested blocks dropped, strings cut): the numbers a reader needs come first, as scalars; blocks and notes follow."""
    r = res.get("roofline")
    if isinstance(r, dict):
        km = r.get("kernels_ms") or {}
        for k in ("mfcc", "dtw", "aggregate", "scan"):
            if k in km:
                r[k + "_ms"] = km[k]
        scal = {k: r[k] for k in ROOFLINE_FIRST if k in r}
        rest_s = {k: v for k, v in r.items() if k not in scal and not isinstance(v, (dict, list, str))}
        rest_o = {k: v for k, v in r.items() if k not in scal and isinstance(v, (dict, list, str))}
        res["roofline"] = {**scal, **rest_s, **rest_o}
    c = res.get("config")
    if isinstance(c, dict):
        first = {k: c[k] for k in ("workload",) if k in c}
        flat = {k: v for k, v in c.items() if k.endswith(("_value", "_frac"))}
        other = {k: v for k, v in c.items() if k not in first and k not in flat}
        scal = {k: v for k, v in other.items() if not isinstance(v, (dict, list))}
        blocks = {k: v for k, v in other.items() if isinstance(v, (dict, list))}
        res["config"] = {**first, **flat, **scal, **blocks}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (BASELINE config C3)")
    ap.add_argument("--templates", type=int, default=8)
    ap.add_argument("--samples", type=int, default=64000, help="samples per stream (4 s @ 16 kHz)")
    ap.add_argument("--template-len", type=int, default=100)
    ap.add_argument("--mfcc-size", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model-type", choices=["tiny", "small", "medium", "large"], default="small", help="--mode model: layer widths of wakeword_nn.rs:305-389 for 195 frames x 16 coefficients")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement: no vector_only / extra_configs / h2d_included blocks")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default=None,
                    help="BASELINE.json presets: C2 = 1 024 streams x 8 templates; C3 = 65 536 x 8 (the default workload); C4 = 65 536 streams x 64 "
                         "templates SPLIT over the --gpus ranks (strong scaling, RCCL gather of the per-stream results); C5 = --mode mlp")
    ap.add_argument("--mode", choices=["dtw", "mlp", "stream", "resample", "model"], default="dtw",
                    help="dtw: the headline MFCC+DTW path (default); mlp: BASELINE config C5, wakeword-model forward; "
                         "model: the wakeword-MODEL detector over whole streams (rp_batch_detect_model, --model-type); "
                         "stream: the same path fed --chunks-per-call 30 ms chunks per call (rp_stream_batch_process); "
                         "resample: 48 kHz -> 16 kHz front-end alone (rp_resample_batch)")
    ap.add_argument("--ingest", action="store_true", help="the headline path with the PCM starting in pinned HOST memory: streams in blocks of "
                    "--ingest-block, block k+1's copy on a copy stream under block k's kernels (SURVEY 8d: H2D included); reports scorings/s and PCIe GB/s")
    ap.add_argument("--ingest-block", type=int, default=8192, help="streams per block of the --ingest pipeline")
    ap.add_argument("--ingest-format", choices=["f32", "i16"], default="f32", help="sample format of the host PCM (i16: decoded inside mfcc_kernel)")
    ap.add_argument("--ingest-blocks", type=int, default=4, help="blocks of the --ingest run (its page-locked host buffer holds all of them: 2.1 GB per f32 block of 8 192 streams)")
    ap.add_argument("--chunks-per-call", type=int, default=1)
    ap.add_argument("--pcm-format", choices=["f32", "i16"], default="f32", help="--mode resample: sample format of the 48 kHz input")
    ap.add_argument("--channels", type=int, default=1, help="--mode resample: interleaved channels of the 48 kHz input")
    ap.add_argument("--mlp-precision", choices=["f32", "bf16", "f32_strict", "f32_fast"], default="bf16")
    ap.add_argument("--template-lens", default="", help="comma-separated template lengths in frames (overrides --templates / "
                    "--template-len), e.g. 108,96,90,93,102 = the shape of the reference's oye_casa_g.rpw")
    ap.add_argument("--score-mode", choices=["average", "max", "median", "p25", "p50", "p75", "p80", "p90", "p95"], default="max")
    ap.add_argument("--avg-gate", action="store_true", help="reference defaults: an averaged template and avg_threshold 0.2 -- windows "
                    "whose avg_score is below it are not compared with the sample templates (wakeword_comp.rs:85-93)")
    ap.add_argument("--avg-threshold", type=float, default=0.2, help="with --avg-gate: DetectorConfig.avg_threshold (reference default 0.2)")
    ap.add_argument("--detect-only", action="store_true", help="do not ask for the per-window score arrays: the call returns detections only and "
                    "(ScoreMode::Max) may abandon DTWs that can no longer reach the threshold (same detections; NOT the headline workload, which "
                    "scores every window against every template)")
    ap.add_argument("--full-scores", action="store_true", help="with --avg-gate: compare every window with every template anyway (RP_CTX_FULL_SCORES)")
    ap.add_argument("--arith", choices=["f32_matrix", "strict_f32", "fast_split"], default="f32_matrix",
                    help="rp_ctx arithmetic of the DTW cost's cosine products (include/rustpotter_hip.h RP_ARITH_*): f32_matrix = the library default "
                         "(matrix cores, three bf16 parts per operand: f32-grade), strict_f32 = f32 vector FMAs only, fast_split = two f16 parts (22-bit)")
    ap.add_argument("--ragged-matrix", action="store_true", help="with --arith fast_split: references of unequal template lengths on dtw_ragged_kernel (RP_CTX_RAGGED_MATRIX)")
    args = ap.parse_args()
    args.total_streams = None
    if args.config == "C2":
        args.streams, args.templates = 1024, 8
    elif args.config == "C3":
        args.streams, args.templates = 65536, 8
    elif args.config == "C4":   # strong scaling: the 65 536 streams are split over the ranks (shard by stream)
        args.total_streams, args.templates = 65536, 64
    elif args.config == "C5":
        args.mode = "mlp"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RP_BENCH_FORCE_PG=1: a ONE-rank run still creates the process group (RCCL on the one GPU a box has) and sends its per-stream block
    # through the collective: the first multi-GPU run of this code must not be the first ncclCommInit (tests/test_gpu_rccl.py)
    force_pg = world == 1 and os.environ.get("RP_BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if force_pg:
        os.environ.setdefault("MASTER_PORT", "29537")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    # RP_BENCH_BACKEND=gloo is a dry-run switch for boxes with fewer GPUs than ranks (ranks then share
    # devices); the driver's runs use the default, nccl == RCCL on ROCm, one rank per GPU.
    backend = os.environ.get("RP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif force_pg:
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"

    import rustpotter_amd as ra

    env = Env(args, ra, torch, dist, dev, world, rank, local_rank, backend)
    env.pg = world > 1 or force_pg
    if args.mode == "mlp":
        res = bench_mlp(env)
    elif args.mode == "model":
        res = bench_model(env)
    elif args.mode == "stream":
        res = bench_stream(env)
    elif args.mode == "resample":
        res = bench_resample(env)
    elif args.ingest:
        res = bench_ingest(env)
    else:
        res = bench_dtw(env)
    if rank == 0:
        print(json.dumps(order_for_the_record(res)))
    if env.pg:
        dist.destroy_process_group()


class Env:
    def __init__(self, args, ra, torch, dist, dev, world, rank, local_rank, backend):
        self.args, self.ra, self.torch, self.dist, self.dev = args, ra, torch, dist, dev
        self.world, self.rank, self.local_rank, self.backend = world, rank, local_rank, backend
        self.cores, self.cpu_model = host_cpu()
        self.pg = world > 1   # a process group exists (main() also sets it for RP_BENCH_FORCE_PG=1 one-rank runs)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.pg:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.pg:
            tt = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def common_config(self):
        """Fields every line carries: where it ran and with which build."""
        p = self.torch.cuda.get_device_properties(self.dev)
        c = {"host_cpu": self.cpu_model, "host_threads_granted": self.cores, "device": p.name, "compute_units": p.multi_processor_count,
             "build": self.ra.build_info(), "world_size": self.world, "backend": self.backend if self.pg else "none (one rank)"}
        if self.pg:
            c.update(self.rank_identities())
        return c

    def rank_identities(self):
        """Self-proving multi-GPU record: the world size the process group reports and every rank's device UUID; with nccl (RCCL)
        the UUIDs must be N distinct devices."""
        p = self.torch.cuda.get_device_properties(self.dev)
        from rustpotter_amd import sharding
        mine = {"rank": self.rank, "local_rank": self.local_rank, "device_index": self.dev.index, "uuid": str(getattr(p, "uuid", "")),
                "pci_bus_id": getattr(p, "pci_bus_id", None), "name": p.name, "pid": os.getpid()}
        return sharding.rank_identities(mine, self.backend)


def make_templates(ra, ctx, torch, dev, lens, K):
    """Template t = the CPU oracle's MFCC of a synthetic utterance seeded SEED+1+t (stream 0), whole-matrix mean normalisation, cut to
    lens[t] -- SURVEY 8d's recipe, exactly how wav_file_extractor.rs:59-67 makes a reference's templates.  The oracle only prepares this
    INPUT (like the synthetic PCM); nothing of the measured path runs through it, and the same arrays go to the cpu_baseline leg."""
    import numpy as np
    from oracle import rp_oracle as orc
    tt = orc.synth_templates(SEED, len(lens), max(lens), K)
    return [np.ascontiguousarray(t[:n], dtype=np.float32) for t, n in zip(tt, lens)]


# ------------------------------------------------------------------------------------------------ kernel models
def cells(Lt, W=5):
    return sum((min(Lt, r + W - 1) - max(1, r - W) + 1) for r in range(1, Lt))


def dtw_kernel_model(env, S, n_win, lens, K, dtw_s, pmc_dtw, ran=None, products=None):
    """Pipe fractions of the DTW kernel of one launch (S x n_win windows x templates `lens`) that took dtw_s seconds.
    valu_issue: SIMD issue cycles of the hot loop (committed ISA mix x rate table) x trips / (SIMDs x 2.4 GHz); mfma_f16: executed
    matrix flops / 2.5 PFLOP/s; hbm: algorithmic bytes / 8 TB/s; valu_flops (kernels without a committed mix): executed vector flops /
    157.3 TFLOP/s.  Every fraction is executed work / (pipe peak x measured time), so none can exceed 1."""
    torch, W, T = env.torch, 5, len(lens)
    rows = S * n_win
    by_len = {}
    for Lt in lens:
        by_len[Lt] = by_len.get(Lt, 0) + 1
    # what ran is what the library says ran (rp_ctx_dtw_kernels); products: "bf16x3" (two matrix instructions per tile) / "f16x2" (one)
    ran = ran or []
    prod = (products or ["f16x2"])[0] if any("mfma" in r or "ragged" in r for r in ran) else None
    ksteps5 = 2 if prod == "bf16x3" else 1
    mfma_wide = "dtw_mfma_wide_kernel" in ran
    mfma_used = mfma_wide or "dtw_mfma_kernel" in ran or "dtw_mfma_group_kernel" in ran
    kernel = "dtw_mfma_wide_kernel" if mfma_wide else "dtw_mfma_kernel" if mfma_used else "dtw_band_kernel" if K == 5 else "dtw_band_wide_kernel"
    # templates whose length occurs once or twice: dtw_ragged_kernel when the library says it ran (rp_ctx_dtw_kernels)
    rag_lens = [Lt for Lt, c in by_len.items() for _ in range(c if c <= 2 else (c % 8 if c % 8 <= 2 else 0))] if K == 5 else []
    ragged = bool(ran) and "dtw_ragged_kernel" in ran and bool(rag_lens)
    if ragged and not mfma_used:
        kernel = "dtw_ragged_kernel"
    f_dtw_ref = sum(cells(Lt) * (2 * K + 7) + 2 * Lt * 2 * K + 2 * Lt * K for Lt in lens)   # SURVEY 8d, reference-shaped
    f_exec_vec = sum((Lt - 1) * 2 * W * (2 * K + 3) for Lt in lens) + sum(-(-c // 8) * ((Lt - 1) * (4 * K + 1) + Lt * K) for Lt, c in by_len.items())
    f_exec_mat = 0.0

    def mfma_chunks(c):  # (chunks with eight template slots, chunks with four, templates left to the register kernels)
        full, rem = divmod(c, 8)
        return full + (1 if rem >= 5 else 0), 1 if 3 <= rem <= 4 else 0, rem if rem <= 2 else 0
    if mfma_used and not mfma_wide:
        f_exec_mat = ksteps5 * sum((mfma_chunks(c)[0] * 3 + mfma_chunks(c)[1] * 2) * (Lt + 1) * 32768 / 32.0 for Lt, c in by_len.items())
        f_exec_vec = sum((c - mfma_chunks(c)[2]) * Lt * 2 * W * 3 + (mfma_chunks(c)[0] + mfma_chunks(c)[1]) * Lt * 40 for Lt, c in by_len.items())
    wide3 = mfma_wide and prod == "bf16x3"   # dtw_mfma_wide3_kernel: chunks of up to FOUR templates, 2 tiles x 6 k-steps per column
    if mfma_wide:
        ksteps = 3   # (round 5: mfcc_size 16 starts its sum at the C operand instead of spending a fourth k-step on the constant slot)
        f_exec_mat = sum(-(-c // 8) * 3 * ksteps * (Lt + 1) * 32768 / 32.0 for Lt, c in by_len.items())
        f_exec_vec = sum(c * Lt * 2 * W * 3 + -(-c // 8) * Lt * 8 * K for Lt, c in by_len.items())
    if wide3:
        kernel = "dtw_mfma_wide3_kernel"
        f_exec_mat = sum(-(-c // 4) * 2 * 6 * (Lt + 1) * 32768 / 32.0 for Lt, c in by_len.items())
        f_exec_vec = sum(c * Lt * 2 * W * 3 + -(-c // 4) * Lt * 14 * K for Lt, c in by_len.items())
    if ragged:
        # per window and column: two v_mfma_f32_32x32x16_f16 per 64 windows, and 3 vector ops per band cell + the norm + the mean term
        f_exec_mat += sum((Lt + 1) * 2 * 32768 / 64.0 for Lt in rag_lens)
        if not mfma_used:
            f_exec_vec = sum(Lt * (2 * W * 4 + 15 + 2 * K) for Lt in rag_lens)
    alg_bytes = rows * (4 * K + 4 * (T + 2))
    pipes = {"hbm": alg_bytes / dtw_s / HBM_PEAK, "mfma_f16": rows * f_exec_mat / dtw_s / MFMA_F16_PEAK}
    extra = {}
    mix_path = MIX_MFMA.get(prod, MIX_MFMA["f16x2"])
    mix = load_json(mix_path)
    only_full8 = mfma_used and not mfma_wide and all(mfma_chunks(c)[1] == 0 and mfma_chunks(c)[2] == 0 for c in by_len.values())
    gmix = load_json("profiles/dtw_mfma_group_isa_mix.json")
    grouped = bool(ran) and "dtw_mfma_group_kernel" in ran and bool(gmix)
    if grouped:
        kernel = "dtw_mfma_group_kernel"
    if mix and only_full8 and n_win >= 32:
        # the hot loop is one block of 12 columns of one 32-window tile of one chunk: a template of L frames is L / 12 trips
        tiles = -(-rows // 32)
        trips = sum(mfma_chunks(c)[0] * Lt / 12.0 for Lt, c in by_len.items()) * tiles
        cyc = trips * mix["valu_issue_cycles_per_trip"]
        if grouped:   # runs of four chunks of one length share the frame work: the group kernel's own mix for those
            g_trips = sum((mfma_chunks(c)[0] // 4) * 4 * Lt / 12.0 for Lt, c in by_len.items()) * tiles
            cyc = (trips - g_trips) * mix["valu_issue_cycles_per_trip"] + g_trips * gmix["valu_issue_cycles_per_trip"]
        pipes["valu_issue"] = cyc / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        if "valu_issue_cycles_per_trip_architectural" in mix:   # the same mix at 2 / 4 / 8 cycles per instruction instead of the measured rates
            arch = trips * mix["valu_issue_cycles_per_trip_architectural"]
            if grouped:
                arch = (trips - g_trips) * mix["valu_issue_cycles_per_trip_architectural"] + g_trips * gmix["valu_issue_cycles_per_trip_architectural"]
            extra["frac_at_architectural_rates"] = arch / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra.update({"valu_issue_cycles_per_launch": cyc, "isa_mix": mix_path + " x profiles/valu_rate_table.json: %d VALU + %d MFMA "
                      "instructions, %.0f SIMD issue cycles per 12-column block of a 32-window tile" % (mix["classes"]["valu"], mix["classes"]["mfma"],
                                                                                                         mix["valu_issue_cycles_per_trip"])})
        # a matrix instruction takes ~19 issue cycles from the vector work beside it at three waves per SIMD (tools/scratch/mfma_valu_overlap_probe.hip:
        # 310 cycles for 108 vector instructions alone, 369 with the column's three matrix instructions); the three-part form runs two waves per SIMD
        # with six matrix instructions per column -- the same price is used (measured there: 17 from the kernel's own time, 25 in the probe)
        extra["valu_plus_matrix_issue_frac"] = (cyc + 19.0 * trips * mix["classes"]["mfma"]) / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        if grouped:
            extra["isa_mix"] = ("profiles/dtw_mfma_group_isa_mix.json x profiles/valu_rate_table.json: %d VALU + %d MFMA instructions, %.0f SIMD issue cycles per "
                                "12-column block of a 32-window tile and chunk (four chunks of one length share a column's operand; %.0f for a chunk outside "
                                "a group)" % (gmix["classes"]["valu"], gmix["classes"]["mfma"], gmix["valu_issue_cycles_per_trip"], mix["valu_issue_cycles_per_trip"]))

    elif ragged and not mfma_used and n_win >= 64 and load_json("profiles/dtw_ragged_isa_mix.json"):
        mix = load_json("profiles/dtw_ragged_isa_mix.json")
        # the hot loop is one block of 16 columns of one template for the 64 windows of a wave
        trips = sum(Lt / 16.0 for Lt in rag_lens) * -(-rows // 64)
        cyc = trips * mix["valu_issue_cycles_per_trip"]
        pipes["valu_issue"] = cyc / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        pipes["lds"] = trips * mix["lds_cycles_per_trip"] / (n_simds(torch, env.dev) / 4 * CLOCK_PEAK * dtw_s)
        if "valu_issue_cycles_per_trip_architectural" in mix:
            extra["frac_at_architectural_rates"] = trips * mix["valu_issue_cycles_per_trip_architectural"] / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra.update({"valu_issue_cycles_per_launch": cyc, "isa_mix": "profiles/dtw_ragged_isa_mix.json x profiles/valu_rate_table.json: %d VALU + %d MFMA "
                      "instructions, %.0f SIMD issue cycles per 16-column block of one template for the 64 windows of a wave" % (
                          mix["classes"]["valu"], mix["classes"]["mfma"], mix["valu_issue_cycles_per_trip"])})
    elif wide3 and K == 16 and n_win >= 32 and load_json("profiles/dtw_mfma_wide3_isa_mix.json"):
        mix = load_json("profiles/dtw_mfma_wide3_isa_mix.json")
        # the hot loop is one block of 16 columns of one 32-window tile of one chunk of up to four templates
        trips = sum(-(-c // 4) * Lt / 16.0 for Lt, c in by_len.items()) * -(-rows // 32)
        cyc = trips * mix["valu_issue_cycles_per_trip"]
        pipes["valu_issue"] = cyc / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra["frac_at_architectural_rates"] = trips * mix["valu_issue_cycles_per_trip_architectural"] / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra["valu_plus_matrix_issue_frac"] = trips * (mix["valu_issue_cycles_per_trip"] + 25.0 * mix["classes"]["mfma"]) / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra.update({"valu_issue_cycles_per_launch": cyc, "isa_mix": "profiles/dtw_mfma_wide3_isa_mix.json x profiles/valu_rate_table.json: %d VALU + %d MFMA "
                      "instructions, %.0f SIMD issue cycles per 16-column block of a 32-window tile of four templates (two waves per SIMD)" % (
                          mix["classes"]["valu"], mix["classes"]["mfma"], mix["valu_issue_cycles_per_trip"])})
    elif mfma_wide and not wide3 and K == 16 and n_win >= 32 and load_json("profiles/dtw_mfma_wide_isa_mix.json"):
        mix = load_json("profiles/dtw_mfma_wide_isa_mix.json")
        # the hot loop is one block of 12 columns of one 32-window tile of one chunk of up to eight templates
        trips = sum(-(-c // 8) * Lt / 12.0 for Lt, c in by_len.items()) * -(-rows // 32)
        cyc = trips * mix["valu_issue_cycles_per_trip"]
        pipes["valu_issue"] = cyc / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra["frac_at_architectural_rates"] = trips * mix["valu_issue_cycles_per_trip_architectural"] / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        # a matrix instruction takes ~25 issue cycles away from the vector work beside it (tools/scratch/mfma_valu_overlap_probe.hip, two waves per SIMD)
        extra["valu_plus_matrix_issue_frac"] = trips * (mix["valu_issue_cycles_per_trip"] + 25.0 * mix["classes"]["mfma"]) / (n_simds(torch, env.dev) * CLOCK_PEAK * dtw_s)
        extra.update({"valu_issue_cycles_per_launch": cyc, "isa_mix": "profiles/dtw_mfma_wide_isa_mix.json x profiles/valu_rate_table.json: %d VALU + %d MFMA "
                      "instructions, %.0f SIMD issue cycles per 12-column block of a 32-window tile (two waves per SIMD: 256 registers)" % (
                          mix["classes"]["valu"], mix["classes"]["mfma"], mix["valu_issue_cycles_per_trip"])})
    else:
        pipes["valu_flops"] = rows * f_exec_vec / dtw_s / VALU_PEAK
    if pmc_dtw and "effective_clock_ghz" in pmc_dtw and "valu_issue" in pipes:
        extra["effective_clock_ghz"] = pmc_dtw["effective_clock_ghz"]
        extra["valu_issue_frac_at_effective_clock"] = pipes["valu_issue"] * 2.4 / pmc_dtw["effective_clock_ghz"]
    bound = max(pipes, key=pipes.get)
    r = {"bound": bound, "kernel": kernel, "frac": pipes[bound], "pipes": pipes,
         "avg_launch_ms": dtw_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
         "ref_flop_rate_vs_vector_peak": rows * f_dtw_ref / dtw_s / VALU_PEAK,
         "ref_flops_per_launch": rows * f_dtw_ref, "executed_vector_flops_per_launch": rows * f_exec_vec,
         "executed_matrix_flops_per_launch": rows * f_exec_mat}
    if bound == "valu_issue":
        r.update({"achieved": extra["valu_issue_cycles_per_launch"] / dtw_s / 1e9, "peak": n_simds(torch, env.dev) * CLOCK_PEAK / 1e9, "unit": "G SIMD-issue-cycles/s"})
    elif bound == "hbm":
        r.update({"achieved": alg_bytes / dtw_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s"})
    elif bound == "mfma_f16":
        r.update({"achieved": rows * f_exec_mat / dtw_s / 1e12, "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s"})
    else:
        r.update({"achieved": rows * f_exec_vec / dtw_s / 1e12, "peak": VALU_PEAK / 1e12, "unit": "TFLOP/s"})
    r.update(extra)
    if prod:
        r["products"] = prod
    r["note"] = ("frac = the largest pipe fraction of the kernel; every entry of `pipes` is executed work / (pipe peak x measured launch time).  "
                 "dtw_mfma_kernel forms the cosine products of a band column on the matrix cores -- products bf16x3: both f32 operands as three bf16 parts "
                 "(exact), two v_mfma_f32_32x32x16_bf16 per tile and column, f32 accumulate (f32-grade, the default); products f16x2: two f16 parts, one "
                 "v_mfma_f32_32x32x16_f16 (22-bit, opt-in) -- and runs the min-recurrence on the vector pipe: VALU issue binds it.  ref_flop_rate_vs_vector_peak prices the "
                 "REFERENCE-shaped flop count (SURVEY 8d: 2K+7 flops per band cell) against the 157.3 TFLOP/s vector peak as earlier rounds' `frac` "
                 "did -- it exceeds 1 because those products left the vector pipe, it is not a fraction of a roof")
    return r


def mfcc_kernel_model(env, S, nf, K, mfcc_s, pmc_mfcc=None):
    torch = env.torch
    alg_bytes, flops = S * nf * (640 + 4 * K), S * nf * 13.2e3
    pipes = {"hbm": alg_bytes / mfcc_s / HBM_PEAK, "valu_flops_ref": flops / mfcc_s / VALU_PEAK}
    extra = {}
    mix = load_json("profiles/mfcc_isa_mix.json")
    if mix and K == 5:
        trips = S * nf / 4.0   # one trip of the tile loop = 4 frames of one wave
        pipes["valu_issue"] = trips * mix["valu_issue_cycles_per_trip"] / (n_simds(torch, env.dev) * CLOCK_PEAK * mfcc_s)
        pipes["lds"] = trips * mix["lds_cycles_per_trip"] / (n_simds(torch, env.dev) / 4 * CLOCK_PEAK * mfcc_s)
        if "valu_issue_cycles_per_trip_architectural" in mix:
            extra["frac_at_architectural_rates"] = trips * mix["valu_issue_cycles_per_trip_architectural"] / (n_simds(torch, env.dev) * CLOCK_PEAK * mfcc_s)
        extra["isa_mix"] = ("profiles/mfcc_isa_mix.json x profiles/valu_rate_table.json: %d VALU + %d LDS instructions, %.0f SIMD issue cycles and %d "
                            "LDS-array cycles per 4-frame tile of a wave" % (mix["classes"]["valu"], mix["classes"]["lds"], mix["valu_issue_cycles_per_trip"],
                                                                             mix["lds_cycles_per_trip"]))
    if pmc_mfcc and "effective_clock_ghz" in pmc_mfcc and "valu_issue" in pipes:   # the chip clocks below its 2.4 GHz peak under this load
        extra["effective_clock_ghz"] = pmc_mfcc["effective_clock_ghz"]
        extra["valu_issue_frac_at_effective_clock"] = pipes["valu_issue"] * 2.4 / pmc_mfcc["effective_clock_ghz"]
    bound = max((k for k in pipes if k != "valu_flops_ref"), key=pipes.get)
    r = {"bound": bound, "kernel": "mfcc_kernel", "frac": pipes[bound], "pipes": pipes, "avg_launch_ms": mfcc_s * 1e3,
         "algorithmic_bytes_per_launch": alg_bytes, "hbm_gbps": alg_bytes / mfcc_s / 1e9}
    if bound == "hbm":
        r.update({"achieved": alg_bytes / mfcc_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s"})
    elif bound == "valu_issue":
        r.update({"achieved": S * nf / 4.0 * mix["valu_issue_cycles_per_trip"] / mfcc_s / 1e9, "peak": n_simds(torch, env.dev) * CLOCK_PEAK / 1e9,
                  "unit": "G SIMD-issue-cycles/s"})
    else:
        r.update({"achieved": S * nf / 4.0 * mix["lds_cycles_per_trip"] / mfcc_s / 1e9, "peak": n_simds(torch, env.dev) / 4 * CLOCK_PEAK / 1e9,
                  "unit": "G LDS-array-cycles/s"})
    r.update(extra)
    r["note"] = "660 B per frame (640 B of new PCM + 4K B out) against 8 TB/s; valu_flops_ref = 13.2 kflop per frame (SURVEY 8d) against 157.3 TFLOP/s"
    return r


def pmc_for(S, N, T, L, K, lens, args):
    """HBM bytes / instruction counts per launch from the committed rocprofv3 --pmc passes of this same command (a profiler cannot
    run inside the timed process); empty for any other workload."""
    pmc, src = {}, None
    if args.avg_gate or args.score_mode != "max" or len(set(lens)) != 1:
        return pmc, src
    tj = load_json("profiles/pmc_traffic_latest.json")
    try:
        w = tj["workload"]
        if (w["streams"], w["samples"], w["templates"], w["template_len"], w["mfcc_size"]) == (S, N, T, L, K):
            for name, d in tj["kernels"].items():
                for kn in ("mfcc", "dtw"):
                    if kn + "_" in name and "hbm_bytes_per_launch_corrected" in d and (kn != "dtw" or "dtw_mfma" in name or "dtw" not in pmc):
                        pmc[kn] = d
            src = "profiles/pmc_traffic_latest.json (committed rocprofv3 --pmc passes of this command, not this run)"
    except Exception:
        pmc = {}
    return pmc, src


# ------------------------------------------------------------------------------------------------ the headline path
class DtwCase:
    """S synthetic streams x templates `lens` resident in HBM and the one C call that is a step."""

    def __init__(self, env, S, lens, K, N, first_stream=0, want_arrays=True, avg_gate=False, avg_threshold=0.2, score_mode="max", full_scores=False,
                 ctx=None, templates=None):
        import numpy as np
        ra, torch, dev = env.ra, env.torch, env.dev
        self.env, self.S, self.lens, self.K, self.N = env, S, lens, K, N
        self.T, self.L = len(lens), max(lens)
        self.nf = ra.mfcc_num_frames(N)
        self.n_win = self.nf - self.L + 1
        self.ctx = ctx or ra.BatchContext(device=env.local_rank, host_pointers=False, full_scores=full_scores, arithmetic=getattr(env.args, "arith", "f32_matrix"),
                                          ragged_matrix=getattr(env.args, "ragged_matrix", False))
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        # templates: T synthetic utterances, MFCC by the CPU oracle, whole-matrix mean normalisation, cut to their
        # length.  Identical arrays are handed to the CPU baseline.
        self.templates = templates or make_templates(ra, self.ctx, torch, dev, lens, K)
        avg_t = None
        if avg_gate:
            # synthetic averaged template: the frame-wise mean of the templates cut to the shortest one (the reference's
            # MfccAverager aligns them by DTW first; only the amount of work matters here)
            lm = min(lens)
            avg_t = np.ascontiguousarray(np.mean([t[:lm] for t in self.templates], axis=0, dtype=np.float32), dtype=np.float32)
        self.tmpl = ra.Templates(self.ctx, self.templates, avg=avg_t)
        self.pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
        self.ctx.synth_dev(SEED, first_stream, S, N, N, self.pcm.data_ptr())
        self.want_arrays = want_arrays
        self.scores = torch.empty((S, self.n_win, self.T), dtype=torch.float32, device=dev) if want_arrays else None
        self.agg = torch.empty((S, self.n_win), dtype=torch.float32, device=dev) if want_arrays else None
        self.max_det = 4
        self.det = torch.zeros((S, self.max_det, 6), dtype=torch.int32, device=dev)
        self.n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
        self.cfg = ra.DetectorConfig()
        self.cfg.score_mode = {"average": 0, "max": 1, "median": 2, "p25": 3, "p50": 4, "p75": 5, "p80": 6, "p90": 7, "p95": 8}[score_mode]
        self.cfg.avg_threshold = avg_threshold if avg_gate else 0.0  # gate off: exactly T DTWs per scoring

    def call(self):
        # one C call: mfcc_kernel -> dtw kernel(s) -> aggregate kernel -> scan_kernel on the launch stream
        self.ctx.batch_detect_dev(self.pcm.data_ptr(), self.S, self.N, self.N, self.tmpl, self.cfg, self.det.data_ptr(), self.n_det.data_ptr(),
                                  self.max_det, self.scores.data_ptr() if self.want_arrays else None, self.agg.data_ptr() if self.want_arrays else None)

    def kernel_times(self, reps):
        """HIP events on the launch stream around every launch (rp_ctx_timing_*): avg ms and launches per kernel."""
        self.ctx.timing_enable(True)
        self.ctx.timing_reset()
        for _ in range(reps):
            self.call()
        self.env.torch.cuda.synchronize()
        k = {name: self.ctx.timing_read(i) for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
        self.ctx.timing_enable(False)
        return k

    def time_steps(self, warmup, steps, after_call=None):
        env = self.env
        for _ in range(warmup):
            self.call()
            if after_call:
                after_call()
        env.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.call()
            if after_call:
                after_call()
        env.fence()
        return env.max_over_ranks(time.perf_counter() - t0)


def bench_dtw(env):
    args, ra, torch, dist, dev, world, rank = env.args, env.ra, env.torch, env.dist, env.dev, env.world, env.rank
    from rustpotter_amd import sharding
    first_stream = None
    if args.total_streams is not None:   # strong scaling: this rank's contiguous block of the fixed stream set
        lo, hi = sharding.shard_bounds(args.total_streams, world, rank)
        args.streams, first_stream = hi - lo, lo
    S, N, K = args.streams, args.samples, args.mfcc_size
    lens = [int(x) for x in args.template_lens.split(",") if x] or [args.template_len] * args.templates
    want_arrays = not (args.avg_gate or args.detect_only)  # the per-window arrays are defined for every window: asking for them keeps every DTW
    case = DtwCase(env, S, lens, K, N, first_stream=sharding.weak_first_stream(S, rank) if first_stream is None else first_stream,
                   want_arrays=want_arrays, avg_gate=args.avg_gate, avg_threshold=args.avg_threshold, score_mode=args.score_mode, full_scores=args.full_scores)
    T, L, nf, n_win = case.T, case.L, case.nf, case.n_win

    # final per-stream result gather (RCCL over xGMI); shards of a fixed stream set may differ by one stream
    gather_ms = []
    gathered = [None]

    def gather():
        if not env.pg:
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        # the per-stream result block, T + 2 floats per stream (best score per template, best aggregate, detections);
        # detect-only / gated runs have no score arrays: their block is the detection count alone
        block = sharding.stream_summary(case.scores, case.agg, case.n_det) if case.want_arrays else case.n_det
        gathered[0] = (sharding.gather_per_stream(block, world, force_collective=env.pg) if first_stream is None
                       else sharding.gather_ragged(block, world, force_collective=env.pg))
        b.record()
        gather_ms.append((a, b))

    dt = case.time_steps(args.warmup, args.steps, after_call=gather)
    scorings_per_step = (S * world if args.total_streams is None else args.total_streams) * n_win
    value = scorings_per_step * args.steps / dt

    # ---- rooflines, measured live with HIP events on the launch stream (rp_ctx_timing_*: one event pair per launch)
    case.ctx.dtw_kernels()
    k_ms = case.kernel_times(max(2, min(args.steps, 5)))
    ran = case.ctx.dtw_kernels()   # the DTW kernel families those calls launched (rp_ctx_dtw_kernels)
    pmc, pmc_src = pmc_for(S, N, T, L, K, lens, args)
    dtw_s, mfcc_s = k_ms["dtw"][0] * 1e-3, k_ms["mfcc"][0] * 1e-3
    products = list(case.ctx.last_dtw_products)
    r_dtw = dtw_kernel_model(env, S, n_win, lens, K, dtw_s, pmc.get("dtw"), ran, products)
    r_dtw["dtw_kernels_launched"] = ", ".join(ran)
    r_dtw["arithmetic"] = args.arith
    r_mfcc = mfcc_kernel_model(env, S, nf, K, mfcc_s, pmc.get("mfcc"))
    for r, kn in ((r_dtw, "dtw"), (r_mfcc, "mfcc")):
        r["launches_timed"] = k_ms[kn][1]
        d = pmc.get(kn, {})
        r["traffic"] = d.get("hbm_bytes_per_launch_corrected")
        r["traffic_source"] = pmc_src if kn in pmc else None
        if r["traffic"]:
            r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes_per_launch"]
        if "instructions_per_launch" in d:
            r["pmc_valu_insts_per_launch"] = d["instructions_per_launch"]["SQ_INSTS_VALU"]
        for kf, lab in (("valu_busy_frac", "pmc_valu_busy_frac_derived"), ("mfma_busy_frac", "pmc_mfma_busy_frac")):
            if kf in d:
                r[lab] = d[kf]
        if "pmc_valu_busy_frac_derived" in r:
            r["pmc_note"] = ("pmc_valu_busy_frac_derived is a construct, not a counter: (4 x SQ_ACTIVE_INST_VALU - SQ_VALU_MFMA_BUSY_CYCLES) / (SIMDs x shader "
                             "cycles), tools/collect_profiles.py; the raw ratio 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles) is above 1")
    if r_dtw.get("traffic_over_algorithmic", 0) > 1.3:
        r_dtw["traffic_note"] = ("the LDS-staged 32-window tiles restage L + 3 frames of MFCC per tile: FETCH is ~2.4x the 0.52 GB MFCC array; at ~170 GB/s "
                                 "this kernel is nowhere near HBM-bound, the re-reads cost nothing today")
    dom = max(("mfcc", "dtw"), key=lambda n: k_ms[n][0])
    roofline = dict(r_dtw if dom == "dtw" else r_mfcc)
    roofline["kernels_ms"] = {k: round(v[0], 4) for k, v in k_ms.items()}
    if k_ms["aggregate"][1] == 0:  # no launch of the aggregate pass: ScoreMode::Max ran inside the DTW kernel
        roofline["aggregate_inside_dtw_kernel"] = True
    work_skipped = args.detect_only or (args.avg_gate and not args.full_scores)
    f_dtw_ref = sum(cells(Lt) * (2 * K + 7) + 2 * Lt * 2 * K + 2 * Lt * K for Lt in lens)
    roofline["path_hbm_frac"] = (value / world) * (640 + 4 * (T + 2)) / HBM_PEAK
    roofline["path_ref_flop_rate_vs_vector_peak"] = (value / world) * (13.2e3 * nf / n_win + f_dtw_ref) / VALU_PEAK

    by_len = {}
    for Lt in lens:
        by_len[Lt] = by_len.get(Lt, 0) + 1
    n_chunks = sum(-(-c // 8) for c in by_len.values())
    tag = {(65536, 8): "C3", (8192, 64): "C4 (per-GPU share)", (1024, 8): "C2"}.get((S, T), "custom") if len(set(lens)) == 1 and lens[0] == 100 else "custom"
    if args.total_streams is not None:
        tag = "C4"
    mfma_kernel = "mfma" in r_dtw["kernel"] or "ragged" in r_dtw["kernel"]
    config = {"workload": "%s: %s synthetic 16 kHz f32 streams x %d templates%s (%g s streams, L=%s, K=%d, band 5, "
                          "ScoreMode::%s, %s)" % (tag, ("%d" % S) if args.total_streams is None else ("%d" % args.total_streams), T,
                                                  " per GPU" if args.total_streams is None else " split over %d rank(s) by stream" % world,
                                                  N / 16000.0, lens[0] if len(set(lens)) == 1 else "/".join(map(str, lens)), K,
                                                  args.score_mode.capitalize(),
                                                  ("averaged template + avg_threshold %g%s, " % (args.avg_threshold, " (reference default)" if args.avg_threshold == 0.2 else "") +
                                                   ("every window scored anyway" if args.full_scores else "gated windows skipped"))
                                                  if args.avg_gate else ("avg gate off" + (", detect-only call: DTWs that cannot reach threshold 0.5 "
                                                                                          "any more are abandoned" if args.detect_only else ""))),
              "streams_per_gpu": S, "templates": T, "samples_per_stream": N, "frames_per_stream": nf,
              "windows_per_stream": n_win, "template_chunks": n_chunks,
              "parallelism": "streams sharded x%d, RCCL all_gather of the per-stream result block (T + 2 floats)" % world}
    config.update(env.common_config())
    if env.pg:
        torch.cuda.synchronize()
        g = [a.elapsed_time(b) for a, b in gather_ms[args.warmup:]]
        rows, cols = gathered[0].shape[0], (gathered[0].shape[1] if gathered[0].dim() > 1 else 1)
        config["gather_ms_per_step"] = {"mean": sum(g) / len(g), "max": max(g), "what": "per-stream result block (best score per template over the stream's "
                                        "windows, best aggregate, detections: T + 2 floats) reduced from the score arrays and all_gathered, HIP events on the "
                                        "launch stream around both (every step; the first %d are warm-up)" % args.warmup,
                                        "gathered_shape": [rows, cols], "bytes_sent_per_gpu": int(rows // world * cols * 4), "bytes_total": int(rows * cols * 4)}
        assert rows == (S * world if args.total_streams is None else args.total_streams), "the gather must return every stream of the job"
    out = {
        "metric": "10ms-frame MFCC+DTW scorings/sec", "value": value, "unit": "scorings/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.total_streams is None else "strong", "vs_baseline": None,
        "dtype": DTYPE_DTW_FAST if "f16x2" in products else DTYPE_DTW, "data": "synthetic", "config": config,
        # work is skipped by design in detect-only / gated runs: pricing the full flop count against the shorter time would
        # print a fraction above 1, so those lines carry the kernel times only
        "roofline": roofline if not work_skipped else None,
        "roofline_other": (r_mfcc if dom == "dtw" else r_dtw) if not work_skipped else None,
    }
    if work_skipped:
        out["kernels_ms"] = roofline["kernels_ms"]
        out["note"] = "work is skipped by design in this mode (gated windows / abandoned DTWs): no roofline fraction is quoted"
    if env.backend != "nccl" and world > 1:
        out["oversubscribed"] = {"devices": torch.cuda.device_count(), "backend": env.backend,
                                 "note": "ranks share GPUs: a launch-path dry run, not a scaling measurement"}
    pairs = case.ctx.dtw_ref_pairs()
    out["dtw_reference_cell_pairs"] = pairs   # windows whose frame norms left the scale-invariant range (rp_ctx_dtw_ref_pairs): 0 on this input
    if args.avg_gate:
        # how many windows pass the gate on this input (one extra pass over the averaged template, not timed)
        mf = torch.empty((S, nf, K), dtype=torch.float32, device=dev)
        case.ctx.mfcc_dev(case.pcm.data_ptr(), S, N, N, K, mf.data_ptr())
        sc_ = torch.empty((S, n_win, T), dtype=torch.float32, device=dev)
        av_ = torch.empty((S, n_win), dtype=torch.float32, device=dev)
        ag_ = torch.empty((S, n_win), dtype=torch.float32, device=dev)
        case.ctx.dtw_dev(mf.data_ptr(), S, nf, case.tmpl, case.cfg.score_ref, case.cfg.band_size, case.cfg.score_mode, 1, sc_.data_ptr(), av_.data_ptr(), ag_.data_ptr())
        torch.cuda.synchronize()
        out["config"]["avg_threshold"] = case.cfg.avg_threshold
        out["config"]["gate_pass_fraction"] = float((~(av_ < case.cfg.avg_threshold)).float().mean().item())
        qs = torch.quantile(av_.flatten()[:: max(1, av_.numel() // 4000000)], torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999], device=dev))
        out["config"]["avg_score_quantiles"] = {"q": [0.001, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999], "avg_score": [round(float(x), 4) for x in qs]}
        del mf, sc_, av_, ag_

    plain = world == 1 and not work_skipped and not args.avg_gate and len(set(lens)) == 1 and not args.no_extras
    # ---- the same step with the cosine products on the vector pipe (the register kernels): what "f32" in the strict sense costs.  Same
    # step count as the headline (at least 10); the scalars sit inside `roofline`, which the driver's record keeps
    if world == 1 and not work_skipped and not args.avg_gate and mfma_kernel and not args.no_extras:
        steps_v = max(10, args.steps)
        with case.ctx.arithmetic("strict_f32"):   # rp_ctx_set_arithmetic: read per call by the library
            dtv = case.time_steps(2, steps_v)
            kv = case.kernel_times(3)
        strict = {"what": "RP_ARITH_STRICT_F32: every DTW kernel of the register family (dtw_band_kernel: the five multiply-adds of a cell as "
                          "v_pk_fma_f32; every product, sum and norm in f32)", "dtype": DTYPE_DTW_VECTOR, "value": S * n_win * steps_v / dtv, "unit": "scorings/s",
                  "steps": steps_v, "ms_per_step": dtv / steps_v * 1e3, "kernels_ms": {k: round(v[0], 4) for k, v in kv.items()},
                  "executed_vector_flop_frac": S * n_win * (sum((Lt - 1) * 10 * (2 * K + 3) for Lt in lens) + n_chunks * ((L - 1) * (4 * K + 1) + L * K)) /
                  (kv["dtw"][0] * 1e-3) / VALU_PEAK}
        out["vector_only"] = strict
        roofline["strict_f32"] = strict
        roofline.update({"strict_f32_value": strict["value"], "strict_f32_ms_per_step": strict["ms_per_step"], "strict_f32_steps": steps_v,
                         "strict_f32_mfcc_ms": strict["kernels_ms"]["mfcc"], "strict_f32_dtw_ms": strict["kernels_ms"]["dtw"],
                         "strict_f32_aggregate_ms": strict["kernels_ms"]["aggregate"]})
        if args.arith == "f32_matrix":   # for the record: the opt-in two-part f16 products (22-bit: NOT the reference's precision) on the same inputs
            with case.ctx.arithmetic("fast_split", args.ragged_matrix):
                dtf = case.time_steps(2, steps_v)
                kf = case.kernel_times(3)
            roofline.update({"fast_split_value": S * n_win * steps_v / dtf, "fast_split_ms_per_step": dtf / steps_v * 1e3, "fast_split_dtw_ms": kf["dtw"][0],
                             "fast_split_note": "RP_ARITH_FAST_SPLIT (two f16 parts per operand, 22-bit products): opt-in, narrower than the reference's f32"})
    # ---- the same step fed i16 PCM (the sample format of every recording the reference's tests hold, tests/detector.rs:361-368): mfcc_kernel
    # decodes in registers, the stage reads half the bytes
    if world == 1 and plain:
        try:
            pcm16 = torch.empty((S, N), dtype=torch.int16, device=dev)
            for s0 in range(0, S, 8192):   # in slices: the f32 -> i16 temporaries of the whole array would not fit beside it
                pcm16[s0:s0 + 8192] = torch.round(case.pcm[s0:s0 + 8192] * 32767.0).to(torch.int16)

            def call16():
                case.ctx.batch_detect_fmt_dev(pcm16.data_ptr(), 1, S, N, N, case.tmpl, case.cfg, case.det.data_ptr(), case.n_det.data_ptr(), case.max_det,
                                              case.scores.data_ptr(), case.agg.data_ptr())
            for _ in range(2):
                call16()
            env.fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                call16()
            env.fence()
            dt16 = time.perf_counter() - t0
            case.ctx.timing_enable(True)
            case.ctx.timing_reset()
            for _ in range(3):
                call16()
            torch.cuda.synchronize()
            k16 = {name: case.ctx.timing_read(i) for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
            case.ctx.timing_enable(False)
            roofline.update({"i16_pcm_value": S * n_win * args.steps / dt16, "i16_pcm_ms_per_step": dt16 / args.steps * 1e3, "i16_pcm_mfcc_ms": k16["mfcc"][0],
                             "i16_pcm_mfcc_hbm_frac": S * nf * (320 + 4 * K) / (k16["mfcc"][0] * 1e-3) / HBM_PEAK})
            del pcm16
        except Exception as e:   # an extra must never take the headline line with it
            roofline["i16_pcm_error"] = repr(e)
    roofline["path"] = {"hbm_frac": roofline["path_hbm_frac"], "ref_flop_rate_vs_vector_peak": roofline["path_ref_flop_rate_vs_vector_peak"],
                        "bytes_per_scoring": 640 + 4 * (T + 2), "ref_flops_per_scoring": 13.2e3 * nf / n_win + f_dtw_ref}

    # ---- CPU baseline: the oracle's restatement of the reference algorithm on this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.avg_gate and len(set(lens)) == 1:
        out["cpu_baseline"] = cpu_baseline_dtw(env, N, case.templates, n_win, T)

    if plain and tag == "C3":
        # ---- the other BASELINE configs, short, in the same process (round-3 review: the driver only ever timed C3)
        extras = {}
        try:
            extras["C2"] = extra_c2(env, case, lens, K, N)
        except Exception as e:   # an extra must never take the headline line with it
            extras["C2"] = {"error": repr(e)}
        del case.scores, case.agg
        torch.cuda.empty_cache()
        for prec in ("bf16", "f32"):
            try:
                extras["C5_" + prec] = extra_c5(env, prec)
            except Exception as e:
                extras["C5_" + prec] = {"error": repr(e)}
        for name, fn in (("model_detector", extra_model_detector), ("front_end", extra_front_end)):
            try:
                extras[name] = fn(env)
            except Exception as e:
                extras[name] = {"error": repr(e)}
        try:
            extras["C4_share"] = extra_c4_share(env, K, N)
        except Exception as e:
            extras["C4_share"] = {"error": repr(e)}
        # SURVEY 8d's second frame size and the shape of the reference's own wakeword files, default arithmetic
        for name, S_, lens_, K_, what in (
                ("K16", 8192, [100] * 8, 16, "8192 synthetic streams x 8 templates of 100 frames, mfcc_size 16 (the wakeword-model front-end's frame size)"),
                ("ragged5", 65536, [108, 96, 90, 93, 102], K, "65536 synthetic streams x 5 templates of 108/96/90/93/102 frames (the shape of the reference's oye_casa_g.rpw)")):
            try:
                extras[name] = extra_dtw_shape(env, S_, lens_, K_, N, what)
            except Exception as e:
                extras[name] = {"error": repr(e)}
        out["extra_configs"] = extras
        # the other BASELINE configs where the driver's record keeps them: scalars first, then the same numbers as one block
        oc = {}
        for name in ("C2", "C4_share", "C5_bf16", "C5_f32", "K16", "ragged5"):
            e = extras.get(name, {})
            if "value" in e:
                oc[name] = {"value": e["value"], "unit": e["unit"], "ms_per_step": e["ms_per_step"], "frac": e["roofline"]["frac"], "bound": e["roofline"]["bound"]}
                config[name + "_value"] = e["value"]
                config[name + "_frac"] = e["roofline"]["frac"]
        config["other_configs"] = oc
        try:
            out["h2d_included"] = ingest_measure(env, case, blocks=2, block_streams=min(8192, S), fmt="f32")
        except Exception as e:
            out["h2d_included"] = {"error": repr(e)}
    return out


def cpu_baseline_dtw(env, N, templates, n_win, T):
    from oracle import rp_oracle as orc
    args, cores = env.args, env.cores
    secs, sc, _ = orc.bench(SEED, cores, N, templates, threads=cores)  # calibration: 1 stream per core
    rate = sc / secs
    s_cpu = int(max(cores, min(4096 * cores, args.cpu_seconds * rate / n_win)))
    secs, sc, _ = orc.bench(SEED, s_cpu, N, templates, threads=cores)
    res = {"value": sc / secs, "unit": "scorings/s", "cores": cores, "kind": "port", "host_cpu": env.cpu_model,
           "sample": "%d of the same synthetic streams x %d templates, %d scorings in %.1f s; C restatement of "
                     "the reference algorithm (complex FFT-480 per frame, dense mel, 3-dot cosine per DTW cell), "
                     "not the Rust crate" % (s_cpu, T, sc, secs)}
    # one thread
    s1 = int(max(1, min(64, 2.0 * (rate / cores) / n_win)))
    secs1, sc1, _ = orc.bench(SEED, s1, N, templates, threads=1)
    res["one_thread"] = {"value": sc1 / secs1, "unit": "scorings/s", "cores": 1, "sample": "%d streams, %d scorings in %.1f s on one thread" % (s1, sc1, secs1)}
    return res


def extra_c2(env, case, lens, K, N):
    """BASELINE config C2 = the first 1 024 streams of the same input, same templates, same context."""
    torch = env.torch
    S2 = 1024
    c2 = DtwCase.__new__(DtwCase)
    c2.__dict__.update(case.__dict__)
    c2.S = S2
    c2.pcm = case.pcm[:S2]
    c2.scores, c2.agg = case.scores[:S2], case.agg[:S2]
    c2.det, c2.n_det = case.det[:S2], case.n_det[:S2]
    dt = c2.time_steps(5, 50)
    c2.ctx.dtw_kernels()
    k = c2.kernel_times(5)
    ran2 = c2.ctx.dtw_kernels()
    r = dtw_kernel_model(env, S2, case.n_win, lens, K, k["dtw"][0] * 1e-3, None, ran2, list(c2.ctx.last_dtw_products))
    return {"workload": "C2: 1024 synthetic 16 kHz f32 streams x %d templates (same input, templates and context as the headline run)" % case.T,
            "value": S2 * case.n_win * 50 / dt, "unit": "scorings/s", "steps": 50, "warmup": 5, "ms_per_step": dt / 50 * 1e3, "dtype": DTYPE_DTW,
            "kernels_ms": {kk: round(v[0], 4) for kk, v in k.items()},
            "roofline": {"bound": r["bound"], "kernel": r["kernel"], "frac": r["frac"], "pipes": r["pipes"], "note": "a launch of 9 504 tiles over 3 072 resident "
                         "waves is 3.1 rounds: the last round runs a tenth full"},
            "path_hbm_frac": S2 * case.n_win * 50 / dt * (640 + 4 * (case.T + 2)) / HBM_PEAK}


def extra_c4_share(env, K, N):
    """One GPU's share of BASELINE config C4 (65 536 streams x 64 templates over 8 GPUs): 8 192 streams x 64 templates."""
    torch = env.torch
    S4, T4 = 8192, 64
    lens4 = [100] * T4
    c4 = DtwCase(env, S4, lens4, K, N, first_stream=0)
    dt = c4.time_steps(2, 10)
    k = c4.kernel_times(3)
    ran4 = c4.ctx.dtw_kernels()
    r = dtw_kernel_model(env, S4, c4.n_win, lens4, K, k["dtw"][0] * 1e-3, None, ran4, list(c4.ctx.last_dtw_products))
    res = {"workload": "C4 share: 8192 synthetic 16 kHz f32 streams x 64 templates (one GPU's part of 65 536 x 64 over 8 GPUs)",
           "value": S4 * c4.n_win * 10 / dt, "unit": "scorings/s", "steps": 10, "warmup": 2, "ms_per_step": dt / 10 * 1e3, "dtype": DTYPE_DTW,
           "kernels_ms": {kk: round(v[0], 4) for kk, v in k.items()},
           "roofline": {"bound": r["bound"], "kernel": r["kernel"], "frac": r["frac"], "pipes": r["pipes"]},
           "path_hbm_frac": S4 * c4.n_win * 10 / dt * (640 + 4 * (T4 + 2)) / HBM_PEAK}
    del c4
    torch.cuda.empty_cache()
    return res


def extra_dtw_shape(env, S, lens, K, N, what):
    """One more shape of the same path in the default arithmetic, short: (S streams, templates of `lens` frames, mfcc_size K)."""
    torch = env.torch
    c = DtwCase(env, S, lens, K, N, first_stream=0)
    dt = c.time_steps(2, 10)
    k = c.kernel_times(3)
    ran = c.ctx.dtw_kernels()
    r = dtw_kernel_model(env, S, c.n_win, lens, K, k["dtw"][0] * 1e-3, None, ran, list(c.ctx.last_dtw_products))
    res = {"workload": what, "value": S * c.n_win * 10 / dt, "unit": "scorings/s", "steps": 10, "warmup": 2, "ms_per_step": dt / 10 * 1e3, "dtype": DTYPE_DTW,
           "kernels_ms": {kk: round(v[0], 4) for kk, v in k.items()}, "dtw_kernels_launched": ", ".join(ran), "products": list(c.ctx.last_dtw_products),
           "roofline": {"bound": r["bound"], "kernel": r["kernel"], "frac": r["frac"], "pipes": r["pipes"]}}
    if "valu_plus_matrix_issue_frac" in r:
        res["roofline"]["valu_plus_matrix_issue_frac"] = r["valu_plus_matrix_issue_frac"]
    del c
    torch.cuda.empty_cache()
    return res


class MlpCase:
    """BASELINE config C5: B rows x 3 120 features (F=195 frames x K=16), Small model 3120 -> 32 -> 16 -> 2
    (src/wakewords/nn/wakeword_nn.rs:325-345), rows resident in HBM."""

    def __init__(self, env, B, precision):
        import numpy as np
        ra, torch, dev = env.ra, env.torch, env.dev
        F = 195
        self.env, self.B, self.precision = env, B, precision
        self.dims = [F * 16, F // 6, F // 12, 2]
        rng = np.random.default_rng(5)
        self.ws = [(rng.standard_normal((self.dims[i + 1], self.dims[i])) / np.sqrt(self.dims[i])).astype(np.float32) for i in range(3)]
        self.bs = [(rng.standard_normal(self.dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
        self.ctx = ra.BatchContext(device=env.local_rank, host_pointers=False)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.model = ra.Model(self.ctx, self.ws, self.bs)
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        self.x = torch.randn((B, self.dims[0]), dtype=torch.float32, device=dev, generator=g)
        self.out = torch.empty((B, self.dims[-1]), dtype=torch.float32, device=dev)

    def call(self):
        self.ctx.mlp_dev(self.model, self.x.data_ptr(), self.B, self.precision, self.out.data_ptr())

    def measure(self, warmup, steps):
        env, torch = self.env, self.env.torch
        for _ in range(warmup):
            self.call()
        env.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.call()
        env.fence()
        dt = env.max_over_ranks(time.perf_counter() - t0)
        self.ctx.timing_enable(True)
        self.ctx.timing_reset()
        for _ in range(5):
            self.call()
        torch.cuda.synchronize()
        ms, n = self.ctx.timing_read(4)
        self.ctx.timing_enable(False)
        return dt, ms, n

    def roofline(self, ms, n):
        B, dims = self.B, self.dims
        alg = B * (dims[0] * 4 + dims[-1] * 4)
        kname = self.ctx.last_mlp_kernel()
        traffic, traffic_src = None, None
        tj = load_json("profiles/pmc_c5_latest.json" if self.precision == "bf16" else "profiles/pmc_c5_f32_latest.json")
        try:
            w = tj["workload"]
            if (w["rows"], w["features"], w["precision"]) == (B, dims[0], self.precision):
                for name, d in tj["kernels"].items():
                    if name.startswith(kname.split("<")[0]) and "hbm_bytes_per_launch_corrected" in d:
                        traffic = d["hbm_bytes_per_launch_corrected"]
                        traffic_src = "profiles/%s (committed rocprofv3 --pmc passes of this command, not this run)" % ("pmc_c5_latest.json" if self.precision == "bf16" else "pmc_c5_f32_latest.json")
        except Exception:
            pass
        flops = B * 2.0 * sum(dims[i] * dims[i + 1] for i in range(3))
        mult = 3.0 if "f16x2" in kname else 1.0   # the split form issues three matrix products per layer-1 product
        r = {"bound": "hbm", "kernel": kname, "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK,
             "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": ms, "launches_timed": n, "algorithmic_bytes_per_launch": alg,
             "pipes": {"hbm": alg / (ms * 1e-3) / HBM_PEAK, "mfma_f16": mult * flops / (ms * 1e-3) / MFMA_F16_PEAK},
             "note": "12 480 B of features in + 8 B of logits out per row (SURVEY.md 8d) against 8 TB/s; avg_launch_ms covers every launch of the forward "
                     "(the split form is followed by a pass over the listed out-of-range rows: none here)"}
        if traffic:
            r["traffic_over_algorithmic"] = traffic / alg
        return r

    def cpu_baseline(self, seconds):
        import threading
        from oracle import rp_oracle as orc
        cores = self.env.cores
        xh = self.x[:4096].cpu().numpy()
        t0 = time.perf_counter()
        orc.mlp_forward(xh[:256], self.ws, self.bs)
        per_row = (time.perf_counter() - t0) / 256
        n_cpu = int(max(cores, seconds * cores / per_row))   # rows of the same input, cycled per thread
        per_thread = max(1, n_cpu // cores)

        def work():
            left = per_thread
            while left > 0:
                n = min(left, xh.shape[0])
                orc.mlp_forward(xh[:n], self.ws, self.bs)   # ctypes releases the GIL: the threads run on separate cores
                left -= n
        th = [threading.Thread(target=work) for _ in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        secs = time.perf_counter() - t0
        n1 = int(max(64, min(4096, 1.0 / per_row)))
        t0 = time.perf_counter()
        orc.mlp_forward(xh[:n1], self.ws, self.bs)
        s1 = time.perf_counter() - t0
        return {"value": per_thread * cores / secs, "unit": "rows/s", "cores": cores, "kind": "port", "host_cpu": self.env.cpu_model,
                "sample": "%d of the same rows through the same model in %.1f s on %d threads; C restatement of the reference "
                          "forward (f32 Linear -> ReLU chain), not the Rust crate / candle" % (per_thread * cores, secs, cores),
                "one_thread": {"value": n1 / s1, "unit": "rows/s", "cores": 1, "sample": "%d rows in %.2f s on one thread" % (n1, s1)}}


MLP_DTYPE = {"bf16": "bf16 inputs, f32 accumulate", "f32": "f32 (layer-1 products on the matrix cores from exact three-part bf16 splits of both f32 operands, f32 accumulate)",
             "f32_strict": "f32 (f32 matrix instructions)",
             "f32_fast": "f32 (layer-1 products: f16x2-split MFMA, 22-bit -- RP_MLP_F32_FAST, narrower than the reference; rows beyond the f16 range: f32 MFMA)"}


def extra_c5(env, precision):
    case = MlpCase(env, 65536, precision)
    dt, ms, n = case.measure(5, 50)
    r = case.roofline(ms, n)
    res = {"workload": "C5: 65536 rows x 3120 features, MLP 3120->32->16->2, %s" % precision, "value": case.B * 50 / dt, "unit": "rows/s", "steps": 50, "warmup": 5,
           "ms_per_step": dt / 50 * 1e3, "dtype": MLP_DTYPE[precision], "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "pipes")}}
    if not env.args.no_cpu_baseline:   # the oracle's f32 forward is the reference for both precisions; timed once, on the f32 entry
        res["cpu_baseline"] = case.cpu_baseline(2.0) if precision == "f32" else "the same rows and model as extra_configs.C5_f32.cpu_baseline"
    del case
    env.torch.cuda.empty_cache()
    return res


def _timed(env, call, warmup, steps):
    for _ in range(warmup):
        call()
    env.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    env.torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def model_detector_measure(env, S, m_type, warmup, steps):
    """The wakeword-MODEL detector over whole streams (rp_batch_detect_model: MFCC of mfcc_size 16 -> every window of 195 frames through
    the model -> scores -> detection state machine), SURVEY 8a row a17 in the form the detector runs it."""
    import ctypes as C
    import numpy as np
    ra, torch, dev = env.ra, env.torch, env.dev
    prec_code = 3 if getattr(env.args, "mlp_precision", "bf16") == "f32_fast" else 0   # RP_MLP_F32_FAST / RP_MLP_F32 (the model detector's input is MFCC: no bf16-input form)
    N, F, K = 64000, 195, 16
    dims = {"tiny": [F * K, F // 15, 2], "small": [F * K, F // 6, F // 12, 2], "medium": [F * K, F // 3, F // 6, 2],
            "large": [F * K, F // 3 * 2, F // 6, 2]}[m_type]
    rng = np.random.default_rng(5)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    ctx = ra.BatchContext(device=env.local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(ctx, ws, bs)
    pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
    ctx.synth_dev(SEED, env.rank * S, S, N, N, pcm.data_ptr())
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
    lab = torch.zeros((S, 4), dtype=torch.int32, device=dev)
    n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    c = cfg._c()
    L = ra.load_library()

    def call():
        if L.rp_batch_detect_model(ctx._h, pcm.data_ptr(), 3, S, N, N, model._h, K, 0, C.byref(c), prec_code, det.data_ptr(), lab.data_ptr(), n_det.data_ptr(), 4) != 0:
            raise RuntimeError("rp_batch_detect_model failed")
    for _ in range(warmup):
        call()
    env.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    env.fence()
    dt = env.max_over_ranks(time.perf_counter() - t0) / steps
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    k_ms = {"mfcc": ctx.timing_read(0)[0], "forward": ctx.timing_read(4)[0]}   # averages over the three timed calls (one timed region per call each)
    ctx.timing_enable(False)
    n_win = ra.mfcc_num_frames(N) - F + 1
    d1p = -(-dims[1] // 32) * 32
    # executed matrix work of layer 1: six bf16 products per feature and output in the default precision (three parts per operand, i + j <= 2),
    # three f16 ones under RP_MLP_F32_FAST (x0 w0 + x1 w0 + x0 w1); outputs padded to tiles of 32, windows to tiles of 32 rows
    rows_exec = S * (-(-n_win // 32) * 32)
    nprod = 3 if prec_code == 3 else 6
    return {"dims": dims, "n_win": n_win, "dt": dt, "kernels_ms": {k: round(v, 4) for k, v in k_ms.items()}, "kernel": ctx.last_mlp_kernel(),
            "algorithmic_flops": S * n_win * 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)),
            "executed_matrix_flops": rows_exec * 2.0 * nprod * dims[0] * d1p, "products_per_product": nprod,
            "dtype": MLP_DTYPE["f32_fast" if prec_code == 3 else "f32"]}


def extra_model_detector(env, S=8192):
    m = model_detector_measure(env, S, "small", 2, 10)
    return {"workload": "%d synthetic 4 s streams, Small model %s on every window of 195 frames x 16 coefficients (%d windows per stream), f32 callers"
                        % (S, "->".join(map(str, m["dims"])), m["n_win"]),
            "value": S * m["n_win"] / m["dt"], "unit": "window scorings/s", "steps": 10, "warmup": 2, "ms_per_step": m["dt"] * 1e3,
            "dtype": m["dtype"], "kernel": m["kernel"], "kernels_ms": m["kernels_ms"],
            "roofline": {"bound": "mfma", "kernel": "the forward (mlp_windows_kernel)", "achieved": m["executed_matrix_flops"] / (m["kernels_ms"]["forward"] * 1e-3) / 1e12,
                         "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": m["executed_matrix_flops"] / (m["kernels_ms"]["forward"] * 1e-3) / MFMA_F16_PEAK,
                         "note": "executed 16-bit matrix flops (%d partial products per layer-1 product, padded tiles) over the forward's launch time; PMC and the clock under the two-part form: profiles/r04_model_detect.txt" % m["products_per_product"]}}


def bench_model(env):
    args, world = env.args, env.world
    S = args.streams if args.streams else 8192
    m = model_detector_measure(env, S, args.model_type, args.warmup, args.steps)
    config = {"workload": "wakeword-model detector: %d synthetic 4 s streams per GPU, %s model %s on every window of 195 frames x 16 coefficients (%d windows per stream)"
                          % (S, args.model_type, "->".join(map(str, m["dims"])), m["n_win"])}
    config.update(env.common_config())
    fwd_s = m["kernels_ms"]["forward"] * 1e-3
    return {"metric": "wakeword-model window scorings/sec (rp_batch_detect_model)", "value": S * world * m["n_win"] / m["dt"], "unit": "window scorings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["dt"] * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": m["dtype"], "data": "synthetic", "config": config,
            "kernels_ms": m["kernels_ms"], "kernel": m["kernel"],
            "roofline": {"bound": "mfma", "kernel": "the forward (%s)" % m["kernel"].split("<")[0], "achieved": m["executed_matrix_flops"] / fwd_s / 1e12,
                         "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": m["executed_matrix_flops"] / fwd_s / MFMA_F16_PEAK, "traffic": None,
                         "algorithmic_flop_rate_tflops": m["algorithmic_flops"] / m["dt"] / 1e12,
                         "note": "executed 16-bit matrix flops (six bf16 partial products per layer-1 product in the default precision, three f16 ones with --mlp-precision f32_fast; outputs and windows padded to tiles of 32) over the forward's "
                                 "launch time (HIP events on the launch stream); the clock holds ~1.8 GHz under this kernel: profiles/r04_model_detect.txt"}}


def extra_front_end(env, S=65536):
    """sample decode + GainNormalizerFilter + BandPassFilter over whole streams (rp_frontend_batch), i16 in, f32 out."""
    import ctypes as C
    ra, torch, dev = env.ra, env.torch, env.dev
    N = 64000
    ctx = ra.BatchContext(device=env.local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    raw = (torch.randn((S, N), device=dev, generator=g) * 3000).to(torch.int16)
    out = torch.empty((S, N), dtype=torch.float32, device=dev)
    rc = ra.RustpotterConfig()
    rc.filters.gain_normalizer.enabled = True
    rc.filters.band_pass.enabled = True
    f = rc._filters_c()
    L = ra.load_library()

    def call():
        if L.rp_frontend_batch(ctx._h, raw.data_ptr(), 1, S, N, N, C.byref(f), 0.05, 33, out.data_ptr(), N, None, None) != 0:
            raise RuntimeError("rp_frontend_batch failed")
    dt = _timed(env, call, 2, 10)
    del raw, out
    torch.cuda.empty_cache()
    alg = S * N * 6.0
    # (a lane owns a stream: the call needs S / 64 >= 4 workgroups per CU to fill the chip, i.e. C3's 65 536 streams; 16 384 run at half the rate)
    return {"workload": "%d synthetic 4 s streams, i16 in, f32 out, gain normaliser (window 33) + band-pass on" % S, "value": S * N / dt, "unit": "samples/s",
            "steps": 10, "warmup": 2, "ms_per_step": dt * 1e3, "dtype": "f32 (bit-exact against the oracle)",
            "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / dt / HBM_PEAK,
                         "algorithmic_bytes": "2 B in + 4 B out per sample; the call reads the PCM twice (chunk RMS, then the filters): 8 B moved per sample",
                         "note": "the filter kernel runs at the measured copy ceiling of its 1 : 2 read : write mix (4.8-5.0 TB/s): profiles/r04_frontend.txt"}}


def bench_mlp(env):
    args, world = env.args, env.world
    case = MlpCase(env, args.streams, args.mlp_precision)
    dt, ms, n = case.measure(args.warmup, args.steps)
    config = {"workload": "C5: %d rows x %d features, MLP %s" % (case.B, case.dims[0], "->".join(map(str, case.dims)))}
    config.update(env.common_config())
    res = {"metric": "wakeword-model rows/sec (BASELINE config C5)", "value": case.B * world * args.steps / dt, "unit": "rows/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": MLP_DTYPE[args.mlp_precision], "data": "synthetic",
           "config": config, "roofline": case.roofline(ms, n)}
    # ---- CPU baseline: the oracle's restatement of the reference forward (candle's Linear -> ReLU chain, f32) on this host's cores
    if env.rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = case.cpu_baseline(args.cpu_seconds)
    return res


# ------------------------------------------------------------------------------------------------ H2D included
def ingest_measure(env, case, blocks, block_streams, fmt):
    """The headline path with the PCM starting in pinned HOST memory (SURVEY 8d "H2D included"), through the product's own pipelined
    entry point rp_batch_detect_ingest: the streams are taken in blocks of `block_streams`, block k+1's hipMemcpyAsync runs on the
    context's copy stream while block k's kernels run, two device blocks are cycled, every block's detections are copied back.
    Host memory: `blocks` x `block_streams` streams, page-locked (torch pin_memory = hipHostMalloc)."""
    ra, torch, dev = env.ra, env.torch, env.dev
    Sb, N = block_streams, case.N
    tdt = torch.float32 if fmt == "f32" else torch.int16
    sfmt = 3 if fmt == "f32" else 1
    src = case.pcm[:Sb] if fmt == "f32" else (case.pcm[:Sb] * 32767.0).round().clamp(-32768, 32767).to(torch.int16)
    S = Sb * blocks
    host = torch.empty((S, N), dtype=tdt, pin_memory=True)
    for k in range(blocks):
        host[k * Sb:(k + 1) * Sb].copy_(src)
    det_h = torch.zeros((S, case.max_det, 6), dtype=torch.int32, pin_memory=True)
    ndet_h = torch.zeros((S,), dtype=torch.int32, pin_memory=True)
    torch.cuda.synchronize()
    bytes_block = Sb * N * (4 if fmt == "f32" else 2)

    def run(n_blocks):
        return case.ctx.batch_detect_ingest_ptr(host.data_ptr(), sfmt, Sb * n_blocks, N, N, case.tmpl, case.cfg, det_h.data_ptr(), ndet_h.data_ptr(),
                                                case.max_det, block_streams=Sb)

    run(min(2, blocks))   # warm-up: the context's workspaces for this block size, the copy stream
    t0 = time.perf_counter()
    lib_seconds = run(blocks)
    dt = time.perf_counter() - t0
    # the same block with the input already resident (no copies): what the kernels alone take
    dbuf = torch.empty((Sb, N), dtype=tdt, device=dev)
    dbuf.copy_(host[:Sb])
    det = torch.zeros((Sb, case.max_det, 6), dtype=torch.int32, device=dev)
    n_det = torch.zeros((Sb,), dtype=torch.int32, device=dev)
    case.ctx.timing_enable(True)
    case.ctx.timing_reset()
    for _ in range(2):
        case.ctx.batch_detect_fmt_dev(dbuf.data_ptr(), sfmt, Sb, N, N, case.tmpl, case.cfg, det.data_ptr(), n_det.data_ptr(), case.max_det)
    torch.cuda.synchronize()
    k = {name: case.ctx.timing_read(i)[0] for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
    case.ctx.timing_enable(False)
    kern = sum(k.values())
    # one copy alone
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dbuf.copy_(host[:Sb], non_blocking=True)
    b.record()
    torch.cuda.synchronize()
    copy_ms = a.elapsed_time(b)
    scor = Sb * case.n_win * blocks
    res = {"what": "rp_batch_detect_ingest: PCM in pinned host memory (%s), %d blocks of %d streams, block k+1's hipMemcpyAsync on the context's copy "
                   "stream under block k's kernels, detections copied back per block; detect-only calls (no per-window score arrays leave the device)" % (fmt, blocks, Sb),
           "value": scor / dt, "unit": "scorings/s", "ms_per_block": dt / blocks * 1e3, "pcie_gbps_achieved": bytes_block * blocks / dt / 1e9,
           "library_wall_seconds": lib_seconds, "copy_alone_ms_per_block": copy_ms, "copy_alone_gbps": bytes_block / (copy_ms * 1e-3) / 1e9,
           "kernels_ms_per_block": {kk: round(v, 4) for kk, v in k.items()}, "kernels_sum_ms_per_block": kern,
           "overlap": "a block takes max(copy, kernels) when the two overlap: copy %.2f ms, kernels %.2f ms, measured %.2f ms per block" % (copy_ms, kern, dt / blocks * 1e3),
           "bytes_per_scoring_over_pcie": bytes_block / (Sb * case.n_win), "detections": int(ndet_h.sum().item())}
    del host, dbuf
    return res


def bench_ingest(env):
    args, ra, torch = env.args, env.ra, env.torch
    assert env.world == 1, "--ingest is a single-GPU measurement"
    S, N, K = args.streams, args.samples, args.mfcc_size
    lens = [int(x) for x in args.template_lens.split(",") if x] or [args.template_len] * args.templates
    Sb = min(args.ingest_block, S)
    case = DtwCase(env, Sb, lens, K, N, want_arrays=False)
    blocks = max(2, min(args.ingest_blocks, S // Sb))
    r = ingest_measure(env, case, blocks=blocks, block_streams=Sb, fmt=args.ingest_format)
    config = {"workload": "%d synthetic 16 kHz %s streams x %d templates from pinned host memory in %d blocks of %d" % (Sb * blocks, args.ingest_format, case.T, blocks, Sb)}
    config.update(env.common_config())
    out = {"metric": "10ms-frame MFCC+DTW scorings/sec (H2D included)", "value": r["value"], "unit": "scorings/s", "n_gpus": 1, "steps": blocks, "warmup": 2,
           "ms_per_step": r["ms_per_block"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_DTW, "data": "synthetic",
           "config": config, "h2d_included": r,
           "roofline": {"bound": "pcie", "kernel": "hipMemcpyAsync H2D", "achieved": r["pcie_gbps_achieved"], "peak": 64.0, "unit": "GB/s", "frac": r["pcie_gbps_achieved"] / 64.0,
                        "traffic": None, "note": "the link binds: PCIe 5.0 x16 = 64 GB/s per direction nominal; the kernels of a block take %.2f ms of its %.2f ms" %
                        (r["kernels_sum_ms_per_block"], r["ms_per_block"])}}
    return out


# ------------------------------------------------------------------------------------------------ other modes
def bench_stream(env):
    """Live serving shape of the same path: S streams per GPU, every call brings --chunks-per-call new
    30 ms chunks per stream (f32, resident in HBM) and returns that call's detections; extractor history,
    MFCC window and detector state stay on the device between calls."""
    args, ra, torch, dist, dev, world = env.args, env.ra, env.torch, env.dist, env.dev, env.world
    S, T, n = args.streams, args.templates, args.chunks_per_call
    ctx = ra.BatchContext(device=env.local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lens = [int(x) for x in args.template_lens.split(",") if x] or [args.template_len] * args.templates
    T = len(lens)
    tmpl = ra.Templates(ctx, make_templates(ra, ctx, torch, dev, lens, args.mfcc_size))
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    sb = ra.StreamBatch(ctx, tmpl, cfg, S, max_chunks_per_call=n)
    pcm = torch.empty((S, 480 * n * 4), dtype=torch.float32, device=dev)  # 4 distinct calls' worth, cycled
    ctx.synth_dev(SEED, 0, S, pcm.shape[1], pcm.shape[1], pcm.data_ptr())
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
    n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
    calls = [0]

    def step():
        off = (calls[0] % 4) * 480 * n
        calls[0] += 1
        sb.process_dev(pcm.data_ptr() + 4 * off, 3, n, pcm.shape[1], det.data_ptr(), n_det.data_ptr(), 4)

    # fill the window first so that every timed call scores complete windows
    for _ in range(-(-max(lens) // (3 * n)) + 1 + args.warmup):
        step()
    env.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    env.fence()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    k_ms = {name: round(ctx.timing_read(i)[0], 4) for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
    ms = dt / args.steps * 1e3
    config = {"workload": "%d live streams x %d templates (%s frames) per GPU, %d chunk(s) of 30 ms per call" % (S, T, "/".join(str(x) for x in sorted(set(lens))), n),
              "real_time_factor": 30.0 * n / ms, "kernels_ms": k_ms}
    config.update(env.common_config())
    return {"metric": "10ms-frame MFCC+DTW scorings/sec (streaming calls)", "value": S * 3 * n * world * args.steps / dt,
            "unit": "scorings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_DTW, "data": "synthetic", "config": config}


def bench_resample(env):
    """The sample-rate converter in front of the path: S streams of 4 s at 48 kHz f32 -> 16 kHz."""
    args, ra, torch, dev = env.args, env.ra, env.torch, env.dev
    if env.world > 1:
        raise SystemExit("--mode resample is a single-GPU measurement (launch it with --gpus 1)")
    fs = 48000
    S = min(args.streams, 16384)  # 16384 x 4 s x 48 kHz f32 = 12.6 GB in, 4.2 GB out, + the staged copy
    fi, fo = ra.resampler_frame_lengths(fs)
    n = args.samples * 3
    nch = n // fi
    ctx = ra.BatchContext(device=env.local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ch = args.channels
    pcm = torch.empty((S, n), dtype=torch.float32, device=dev)
    ctx.synth_dev(SEED, 0, S, n, n, pcm.data_ptr())
    if args.pcm_format == "i16" or ch > 1:
        mono = (pcm * 32767.0).round().to(torch.int16) if args.pcm_format == "i16" else pcm
        pcm = mono.unsqueeze(2).expand(S, n, ch).contiguous().view(S, n * ch)
        del mono
    fmt = 1 if args.pcm_format == "i16" else 3
    out = torch.empty((S, nch * fo), dtype=torch.float32, device=dev)

    def step():
        ctx.resample_dev(pcm.data_ptr(), fmt, ch, fs, S, n, n * ch, out.data_ptr(), nch * fo)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms, _n = ctx.timing_read(5)
    alg = S * nch * (fi * ch * (2 if fmt == 1 else 4) + fo * 4)
    gemm = os.environ.get("RP_RESAMPLE_GEMM") == "1"
    if gemm:   # the general kernel: one [2*fi x fo] product per output frame on the f32 matrix cores
        flops = S * nch * 2.0 * (2 * fi) * fo
        roof = {"bound": "mfma", "kernel": "resample_mfma_kernel", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3,
                "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 157.3e12, "traffic": None, "avg_launch_ms": ms}
    else:      # 48 kHz: pruned-FFT kernel, ~110 kflop per frame (8 FFT-240 + twiddle / untangle passes)
        roof = {"bound": "hbm", "kernel": "resample48_fft_kernel", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK, "traffic": None, "avg_launch_ms": ms,
                "valu_frac_fp32": S * nch * 110e3 / (ms * 1e-3) / VALU_PEAK}
    config = {"workload": "%d streams x %d samples at 48 kHz %s, %d channel(s)" % (S, n, args.pcm_format, ch)}
    config.update(env.common_config())
    return {"metric": "resampled 10ms output frames/sec (48 kHz -> 16 kHz)", "value": S * nch * 3 / dt, "unit": "frames/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "roofline": roof}


if __name__ == "__main__":
    main()
