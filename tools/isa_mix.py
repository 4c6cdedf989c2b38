#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop, priced with the committed rate table.

    python tools/isa_mix.py rp_dtw_mfma.hip 'dtw_mfma_kernel<5, 12, false, 8>' [--min-mfma 36] [--flags "-DX ..."] [--out profiles/x.json]

Compiles the source to gfx950 assembly with the product's flags, takes the named kernel, finds its loops (a label that a later
branch jumps back to) and reports the hot one -- the smallest loop with at least --min-mfma matrix instructions (the 12-column block
of dtw_mfma_kernel has 36), else the largest loop that contains no other loop: counts per opcode and per class, and the SIMD issue
cycles of one trip priced with profiles/valu_rate_table.json (cycles per wave-instruction, measured by
tools/scratch/valu_rate_probe.hip on an MI355X).  bench.py turns that into the VALU-issue bound of the kernel."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-pass-failed"]


def file_flags(src):
    """the per-file flags of rustpotter_amd/csrc/Makefile (FILE_FLAGS_<source>: the pragma-unroll budget of the matrix DTW kernels)"""
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rustpotter_amd", "csrc"), "print-file-flags", "F=" + os.path.basename(src)],
                         stdout=subprocess.PIPE, text=True).stdout
    return out.split()


def rate_of(op, table):
    for pat, cyc in table["rates"]:
        if re.fullmatch(pat, op):
            return cyc
    return table["default_valu"]


def main():
    args = sys.argv[1:]
    out = None
    min_mfma = 0
    if "--out" in args:
        i = args.index("--out"); out = args[i + 1]; del args[i:i + 2]
    whole = "--whole" in args   # straight-line kernels (tools/scratch probes): price the whole body
    if whole:
        args.remove("--whole")
    flags = []
    if "--flags" in args:
        i = args.index("--flags"); flags = args[i + 1].split(); del args[i:i + 2]
    if "--min-mfma" in args:
        i = args.index("--min-mfma"); min_mfma = int(args[i + 1]); del args[i:i + 2]
    src, want = args[0], args[1]
    table = json.load(open(os.path.join(ROOT, "profiles", "valu_rate_table.json")))
    path = src if os.path.isabs(src) or os.path.exists(src) else os.path.join(ROOT, "rustpotter_amd", "csrc", src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + FLAGS + file_flags(src) + flags + ["--cuda-device-only", "-S"]
    # one compilation per (sources, command): the assembly of a file is kept under the temp directory (tests price several kernels of one file)
    import hashlib
    h = hashlib.sha256(" ".join(cmd).encode())
    csrc = os.path.join(ROOT, "rustpotter_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(path, "rb").read())
    cache = os.path.join(tempfile.gettempdir(), "rp_isa_mix_cache")
    os.makedirs(cache, exist_ok=True)
    asm = os.path.join(cache, h.hexdigest()[:32] + ".s")
    if not os.path.exists(asm):
        tmp = asm + ".%d.tmp" % os.getpid()
        subprocess.check_call(cmd + ["-o", tmp, path], stderr=subprocess.DEVNULL)
        os.replace(tmp, asm)
    lines = open(asm).read().splitlines()
    # kernel bodies: "<mangled>:" ... "s_endpgm" / ".Lfunc_end"
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
            if re.sub(r"\(.*", "", name).endswith(want) or want in name.split("(")[0]:
                start = i
                break
    assert start is not None, "kernel not found: " + want
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    assert loops or whole, "no loop"
    def insts(a, b):
        return [l.split()[0] for l in body[a:b + 1] if re.match(r"\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
    if whole:
        a, b = 0, len(body) - 1
    elif min_mfma:
        cand = [ab for ab in loops if sum(o.startswith("v_mfma") for o in insts(*ab)) >= min_mfma]
        assert cand, "no loop with that many matrix instructions"
        a, b = min(cand, key=lambda ab: len(insts(*ab)))
    else:
        inner = [ab for ab in loops if not any(o != ab and ab[0] <= o[0] and o[1] <= ab[1] for o in loops)]
        a, b = max(inner, key=lambda ab: len(insts(*ab)))
    ops = insts(a, b)
    counts = {}
    for o in ops:
        counts[o] = counts.get(o, 0) + 1
    classes = {"valu": 0, "mfma": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
    # LDS-array cycles per wave-instruction, conflict-free (MI355X_MICROARCH.md, LDS table)
    lds_cyc = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read_b96": 8, "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_write_b32": 4,
               "ds_write_b64": 6, "ds_write_b96": 10, "ds_write_b128": 13, "ds_write2_b32": 8, "ds_write2_b64": 12}
    valu_cycles = 0.0
    arch_cycles = 0   # the same mix at the architectural issue rates: 2 cycles for a full-rate wave64 instruction, 4 half rate, 8 quarter rate
    lds_cycles = 0
    for o, n in counts.items():
        if o.startswith("v_mfma"): classes["mfma"] += n
        elif o.startswith("v_"):
            classes["valu"] += n
            valu_cycles += n * rate_of(o, table)
            arch_cycles += n * (2 if rate_of(o, table) < 3.5 else 4 if rate_of(o, table) < 6 else 8)
        elif o.startswith("s_"): classes["salu"] += n
        elif o.startswith("ds_"):
            classes["lds"] += n
            lds_cycles += n * lds_cyc.get(o, 4)
        elif o.startswith(("global_", "buffer_", "flat_", "scratch_")): classes["vmem"] += n
        else: classes["other"] += n
    res = {"source": src, "kernel": want, "extra_flags": " ".join(flags), "loop_lines": [a, b], "instructions": len(ops), "classes": classes,
           "valu_issue_cycles_per_trip": round(valu_cycles, 1), "valu_issue_cycles_per_trip_architectural": arch_cycles, "lds_cycles_per_trip": lds_cycles, "rate_table": "profiles/valu_rate_table.json",
           "opcodes": dict(sorted(counts.items(), key=lambda kv: -kv[1]))}
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(os.path.join(ROOT, out), "w"), indent=1)


if __name__ == "__main__":
    main()
