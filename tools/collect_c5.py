"""Turns the output of `tools/prof.sh c5 <precision>` (gpurun_out/<dir>/: bench lines, rocprofv3 --kernel-trace --stats, PMC passes of
`bench.py --config C5`) into the committed evidence under profiles/: <round>_c5_kernel_stats.csv, <round>_c5_pmc.json
(= pmc_c5_latest.json, which bench.py --mode mlp reads for roofline.traffic) and the bench lines.
Usage: python tools/collect_c5.py [round tag, default r05] [gpurun_out sub-directory, default c5_bf16] [precision bf16 | f32]
(f32 writes pmc_c5_f32_latest.json)"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
D = "gpurun_out/" + (sys.argv[2] if len(sys.argv) > 2 else "c5_bf16")
P = sys.argv[3] if len(sys.argv) > 3 else "bf16"
SUF = "" if P == "bf16" else "_" + P
short = lambda n: n.split("(")[0].replace("void ", "").replace("rp::", "")


def pmc(tag):
    f = glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (D, tag), recursive=True)
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "mlp" in r["Kernel_Name"]:
            a[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: statistics.median(v) for c, v in d.items()} for k, d in a.items()}


def durations(tag):
    f = glob.glob("%s/%s/**/*kernel_trace.csv" % (D, tag), recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mlp" in r["Kernel_Name"]:
            d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


fe, wr, sq, ins, clk, lds = (pmc(t) for t in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT"))
dur = durations("prof_c5")
clk_dur = durations("pmc_GRBM_GUI_ACTIVE")
line = json.loads([l for l in open(D + "/c5_%s.json" % P) if l.startswith("{")][-1])
out = {"command": "rocprofv3 --kernel-trace --pmc <CTRS> --output-format csv -- python3 bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline "
                  "(--mlp-precision %s; separate passes: FETCH_SIZE; WRITE_SIZE; SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY; SQ_INSTS_VALU SQ_INSTS_LDS "
                  "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA; GRBM_GUI_ACTIVE; SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- tools/prof.sh c5)" % P,
       "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch (TCC_EA0 request counters); FETCH_SIZE of 16-byte-per-lane streaming reads -- global_load "
                "and LDS-DMA alike -- under-reports by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): doubled below; medians over the launches of the run",
       "workload": {"rows": 65536, "features": 3120, "precision": P, "model": "3120->32->16->2"}, "kernels": {}}
for k in fe:
    f, w = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
    d = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024,
         "note": "reads are 16 B/lane LDS-DMA: FETCH_SIZE doubled per the guide",
         "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"]}
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch_corrected"] / d["algorithmic_bytes_per_launch"]
    if k in dur:
        d["kernel_trace_launches"] = len(dur[k])
        d["kernel_trace_avg_ms"] = sum(dur[k]) / len(dur[k]) / 1e6
        d["kernel_trace_median_ms"] = statistics.median(dur[k]) / 1e6
        d["hbm_frac_from_kernel_trace_avg"] = d["algorithmic_bytes_per_launch"] / (d["kernel_trace_avg_ms"] * 1e-3) / 8.0e12
    if k in sq and sq[k].get("SQ_WAVE_CYCLES"):
        d["sq_fractions_of_wave_cycles"] = {n: sq[k][n] / sq[k]["SQ_WAVE_CYCLES"] for n in sq[k] if n != "SQ_WAVE_CYCLES"}
    if k in ins:
        d["instructions_per_launch"] = ins[k]
    if k in lds:
        d["lds"] = lds[k]
    if k in clk and k in clk_dur:
        d["effective_clock_ghz"] = clk[k]["GRBM_GUI_ACTIVE"] / 8.0 / statistics.median(clk_dur[k])
    out["kernels"][k] = d
json.dump(out, open("profiles/pmc_c5%s_latest.json" % SUF, "w"), indent=1)
json.dump(out, open("profiles/%s_c5%s_pmc.json" % (R, SUF), "w"), indent=1)
shutil.copy(glob.glob(D + "/prof_c5/**/*kernel_stats.csv", recursive=True)[0], "profiles/%s_c5%s_kernel_stats.csv" % (R, SUF))
# keep the profile file readable: drop torch's one-off initialisation kernels with their kilobyte-long names
rows = [r for r in csv.reader(open("profiles/%s_c5%s_kernel_stats.csv" % (R, SUF)))]
with open("profiles/%s_c5%s_kernel_stats.csv" % (R, SUF), "w", newline="") as fo:
    wtr = csv.writer(fo, quoting=csv.QUOTE_ALL)
    for r in rows:
        wtr.writerow([c if len(c) < 200 else c[:160] + "...(name truncated)" for c in r])
open("profiles/%s_c5%s_bench_under_rocprof.json" % (R, SUF), "w").write([l for l in open(D + "/prof_c5.log") if l.startswith("{")][-1])
for p in (P,):
    src = "%s/c5_%s.json" % (D, p)
    if os.path.exists(src) and os.path.getsize(src) > 10:
        shutil.copy(src, "profiles/bench_%s_c5_%s.json" % (R, p))
        x = json.loads(open(src).read().strip().splitlines()[-1])
        print(p, "%.4g %s" % (x["value"], x["unit"]), "%.4f ms" % x["roofline"]["avg_launch_ms"], "frac %.3f" % x["roofline"]["frac"], x.get("cpu_baseline", {}).get("value"))
for k, d in out["kernels"].items():
    print(k[:40], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items() if not isinstance(b, (dict, str))})
