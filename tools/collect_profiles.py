"""Turns the rocprofv3 output of `tools/prof.sh round` (tools/prof.sh stats / pmcset; tags final, final_fetch, final_write, final_sq,
final_inst under gpurun_out/) and the bench lines under gpurun_out/final/ into the committed files under profiles/."""
import csv, glob, json, collections, statistics, shutil, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = sys.argv[1] if len(sys.argv) > 1 else "r06"   # round tag of the files written
short = lambda n: n.split("(")[0]

def agg(tag):
    path = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: statistics.median(v) for c, v in d.items()} for k, d in a.items()}

fe, wr, sq, ins = agg("final_fetch"), agg("final_write"), agg("final_sq"), agg("final_inst")
try:  # VALU / matrix pipe occupancy (its own pass)
    vp = agg("final_valu")
except Exception:
    vp = {}
try:  # effective shader clock: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the launch's duration in the same pass
    clk = agg("final_clk")
    trc = glob.glob("gpurun_out/pmc_final_clk/**/*kernel_trace.csv", recursive=True)[0]
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(trc)):
        durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    clk = {k: (v["GRBM_GUI_ACTIVE"], statistics.median(durs[k])) for k, v in clk.items() if k in durs}
except Exception:
    clk = {}
out = {"command": "rocprofv3 --kernel-trace --pmc <CTRS> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "
                  "(separate passes: FETCH_SIZE; WRITE_SIZE; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY; "
                  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES; SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES; GRBM_GUI_ACTIVE -- tools/prof.sh pmcset, tools/prof.sh round)",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch as reported by rocprofv3 (TCC_EA0 request counters); FETCH_SIZE of "
                "16-byte-per-lane streaming reads under-reports by 2x on gfx950 (MI355X_MICROARCH.md HBM section); values are the "
                "MEDIAN over the launches of a kernel in the run (the run also holds one tiny mfcc launch for the templates)",
       "workload": {"streams": 65536, "samples": 64000, "templates": 8, "template_len": 100, "mfcc_size": 5}, "kernels": {}}
for k in fe:
    f, w = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
    d = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w}
    if "mfcc_kernel" in k:
        d["hbm_bytes_per_launch_corrected"] = (2 * f + w) * 1024
        d["note"] = "reads are 16 B/lane: FETCH_SIZE doubled per the guide"
    elif any(x in k for x in ("dtw_mfma_kernel", "dtw_band_kernel", "dtw_band2_kernel", "aggregate_kernel", "scan_kernel")):
        d["hbm_bytes_per_launch_corrected"] = (f + w) * 1024
        d["note"] = "4-byte reads: FETCH_SIZE taken at face value"
    if k in sq and sq[k].get("SQ_WAVE_CYCLES"):
        d["sq_fractions_of_wave_cycles"] = {n: sq[k][n] / sq[k]["SQ_WAVE_CYCLES"] for n in sq[k] if n != "SQ_WAVE_CYCLES"}
    if k in ins:
        d["instructions_per_launch"] = ins[k]
    if k in clk and clk[k][1] > 0:
        d["grbm_gui_active_per_launch"] = clk[k][0]
        d["effective_clock_ghz"] = clk[k][0] / 8.0 / clk[k][1]   # cycles per ns
        if k in vp and vp[k].get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the waves; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the SIMDs
            simd_cycles = 1024.0 * clk[k][0] / 8.0
            d["valu_pipe_counters_per_launch"] = vp[k]
            mfma_busy = vp[k].get("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0
            # SQ_ACTIVE_INST_VALU charges an MFMA its whole execution time (32 cycles) although vector instructions of other waves
            # issue beside part of it: with MFMAs the raw ratio can exceed 1; the vector share is what is left without them
            d["valu_active_frac_raw"] = 4.0 * vp[k]["SQ_ACTIVE_INST_VALU"] / simd_cycles
            d["valu_busy_frac"] = (4.0 * vp[k]["SQ_ACTIVE_INST_VALU"] - mfma_busy) / simd_cycles
            d["mfma_busy_frac"] = mfma_busy / simd_cycles
    out["kernels"][k] = d
json.dump(out, open("profiles/pmc_traffic_latest.json", "w"), indent=1)
json.dump(out, open("profiles/%s_final_pmc.json" % R, "w"), indent=1)
shutil.copy(glob.glob("gpurun_out/prof_final/**/*kernel_stats.csv", recursive=True)[0], "profiles/%s_final_kernel_stats.csv" % R)
line = [l for l in open("gpurun_out/prof_final.log") if l.startswith("{")][-1]
open("profiles/%s_final_bench_under_rocprof.json" % R, "w").write(line)
names = {"bench_default": R + "_final_bench", "stream1": "bench_%s_stream_1chunk" % R, "stream8": "bench_%s_stream_8chunks" % R,
         "rs_fft": "bench_%s_resample_fft" % R, "rs_gemm": "bench_%s_resample_gemm" % R, "rs_fft_i16_stereo": "bench_%s_resample_fft_i16_stereo" % R,
         "c5_bf16": "bench_%s_c5_bf16" % R, "c5_f32": "bench_%s_c5_f32" % R, "k16": "bench_%s_k16_8192streams" % R, "k13": "bench_%s_k13_8192streams" % R,
         "c4": "bench_%s_c4_per_gpu" % R, "c2": "bench_%s_c2" % R, "c4_one_gpu": "bench_%s_c4_preset_one_gpu" % R,
         "c4_two_ranks_one_gpu": "bench_%s_c4_preset_two_ranks_one_gpu_dry_run" % R, "ragged5": "bench_%s_ragged5_templates" % R, "t3": "bench_%s_t3_len126" % R,
         "median": "bench_%s_median" % R, "gate_default": "bench_%s_avg_gate_default" % R, "gate_default_full": "bench_%s_avg_gate_default_full_scores" % R,
         "gate_04": "bench_%s_avg_gate_04" % R, "gate_04_full": "bench_%s_avg_gate_04_full_scores" % R,
         "two_ranks_one_gpu": "bench_%s_two_ranks_one_gpu_dry_run" % R, "detect_only": "bench_%s_detect_only" % R,
         "detect_only_ragged5": "bench_%s_detect_only_ragged5" % R, "model_small": "bench_%s_model_detector_small" % R, "model_medium": "bench_%s_model_detector_medium" % R, "model_large": "bench_%s_model_detector_large" % R,
         "ingest_f32": "bench_%s_ingest_f32" % R, "ingest_i16": "bench_%s_ingest_i16" % R,
         "ragged5_matrix": "bench_%s_ragged5_templates_matrix_optin" % R, "ragged3_alexa": "bench_%s_ragged3_alexa_lens" % R,
         "ragged3_alexa_matrix": "bench_%s_ragged3_alexa_lens_matrix_optin" % R,
         "c3_fast_split": "bench_%s_c3_fast_split" % R, "c3_strict_f32": "bench_%s_c3_strict_f32" % R, "c2_fast_split": "bench_%s_c2_fast_split" % R,
         "c4_fast_split": "bench_%s_c4_per_gpu_fast_split" % R, "k16_fast_split": "bench_%s_k16_8192streams_fast_split" % R,
         "c5_f32_fast": "bench_%s_c5_f32_fast" % R, "c5_f32_strict": "bench_%s_c5_f32_strict" % R,
         "k16_strict_f32": "bench_%s_k16_8192streams_strict_f32" % R, "k13_strict_f32": "bench_%s_k13_8192streams_strict_f32" % R,
         "t4": "bench_%s_t4" % R, "t4_strict_f32": "bench_%s_t4_strict_f32" % R, "t4_fast_split": "bench_%s_t4_fast_split" % R}
for a, b in names.items():
    src = "gpurun_out/final/%s.json" % a
    if os.path.exists(src) and os.path.getsize(src) > 10:
        shutil.copy(src, "profiles/%s.json" % b)
        x = json.loads(open(src).read().strip().splitlines()[-1])
        print(b, "%.4g %s" % (x["value"], x["unit"]), "%.3f ms" % x["ms_per_step"], (x.get("roofline") or {}).get("kernels_ms", x.get("kernels_ms", "")))
txt = []
for f in ("model_detect.txt", "latency.txt", "frontend.txt", "c2_rounds.txt", "mfma_timeline.txt"):
    p = "gpurun_out/final/" + f
    if os.path.exists(p):
        txt.append(open(p).read().strip())
open("profiles/%s_final_misc.txt" % R, "w").write("\n".join(txt) + "\n")
tr = glob.glob("gpurun_out/prof_final/**/*kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in dur.items():
    if max(v) > 0.1:
        print("%-44s %d launches, median %.3f ms" % (k[:44], len(v), statistics.median(v)))
for k, d in out["kernels"].items():
    if "hbm_bytes_per_launch_corrected" in d:
        print("%-44s %.3f GB" % (k[:44], d["hbm_bytes_per_launch_corrected"] / 1e9),
              {a: round(b, 3) for a, b in d.get("sq_fractions_of_wave_cycles", {}).items()})
