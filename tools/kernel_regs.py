#!/usr/bin/env python3
"""Registers, spills, LDS and occupancy of the kernels of one .hip source (the compiler's kernel-resource-usage remarks).
usage: python tools/kernel_regs.py rp_dtw_mfma.hip [name substring ...] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0] if os.path.isabs(args[0]) else os.path.join(ROOT, "rustpotter_amd", "csrc", args[0])
def file_flags(src):
    """the per-file flags of rustpotter_amd/csrc/Makefile (FILE_FLAGS_<source>: the pragma-unroll budget of the matrix DTW kernels)"""
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rustpotter_amd", "csrc"), "print-file-flags", "F=" + os.path.basename(src)],
                         stdout=subprocess.PIPE, text=True).stdout
    return out.split()


cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
       "-Wno-pass-failed", "--cuda-device-only", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", src] + file_flags(src) + extra
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\S+ )?Function Name: (\S+)", line)
    if m:
        cur = subprocess.check_output(["c++filt", m.group(1)], text=True).strip(); rows[cur] = {}; continue
    m = re.search(r"remark: (?:\S+ )?\s*([A-Za-z ]+?)(?: \[bytes/\w+\]| \[waves/SIMD\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
print("%-5s %-5s %-6s %-5s %-6s %-7s %-4s name" % ("vgpr", "agpr", "spill", "sgpr", "scrtch", "lds", "occ"))
for name, r in sorted(rows.items()):
    if args[1:] and not any(p in name for p in args[1:]): continue
    print("%-5s %-5s %-6s %-5s %-6s %-7s %-4s %s" % (r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs", "?"),
          r.get("ScratchSize", "?"), r.get("LDS Size", "?"), r.get("Occupancy", "?"), re.sub(r"\(.*", "", name)[:120]))
