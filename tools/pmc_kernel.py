#!/usr/bin/env python3
"""Medians of every counter of the `tools/prof.sh pmcset <tag> ...` passes for the kernels whose name contains <substring>, the launch
duration from the same passes and the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration).
usage: python tools/pmc_kernel.py <tag> <kernel name substring>"""
import collections, csv, glob, statistics, sys
tag, sub = sys.argv[1], sys.argv[2]
out, dur = {}, []
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % tag)):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if sub in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: statistics.median(v) for k, v in a.items()})
    t = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if t:
        dur += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if sub in r["Kernel_Name"]]
ns = statistics.median(dur) if dur else float("nan")
print("kernel ~ %s: median duration %.4f ms over %d launches" % (sub, ns / 1e6, len(dur)))
for k, v in sorted(out.items()):
    print("  %-28s %.5g" % (k, v))
if "GRBM_GUI_ACTIVE" in out:
    cyc = out["GRBM_GUI_ACTIVE"] / 8.0
    print("  effective clock %.3f GHz; SIMD cycles per launch %.4g" % (cyc / ns, cyc * 1024))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out:
        print("  matrix pipe busy %.3f; (4 x SQ_ACTIVE_INST_VALU - MFMA busy) / SIMD cycles %.3f" % (
            out["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), (4 * out.get("SQ_ACTIVE_INST_VALU", 0) - out["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * 1024)))
    if "SQ_LDS_IDX_ACTIVE" in out:
        print("  LDS busy %.3f (bank conflicts %.3f of it)" % (out["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), out.get("SQ_LDS_BANK_CONFLICT", 0) / out["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_WAVE_CYCLES" in out:
        print("  of wave cycles: issuing %.3f, waiting on an instruction %.3f, waiting on anything %.3f" % tuple(
            out.get(k, 0) / out["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")))
