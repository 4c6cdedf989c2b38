"""The matrix-core DTW kernels are built around compile-time row-slot -> register mappings inside fully unrolled 12 / 16-column blocks.  If
the compiler does not unroll a block (its default budget for `#pragma unroll` is too small for them: csrc/Makefile FILE_FLAGS_*, DESIGN.md
4.2 BUILD) the kernel still compiles and still passes parity -- with its accumulators in scratch memory, several times slower.  This test
reads the compiler's own resource remarks (tools/kernel_regs.py compiles with the Makefile's flags; CPU only, hipcc cross-compiles) and fails
on that cliff."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (source, kernel, largest number of spilled registers accepted, largest register count)
CASES = [
    ("rp_dtw_mfma_wide3.hip", "dtw_mfma_wide3_kernel<16, 5, 8>", 0, 256),
    ("rp_dtw_mfma_wide3.hip", "dtw_mfma_wide3_kernel<13, 5, 8>", 0, 256),
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 8, false, 8, true>", 0, 256),      # the headline kernel: two waves per SIMD, nothing spilled
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 12, false, 8, true>", 96, 168),    # its twelve-wave build (RP_MFMA3_WAVES=12)
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 12, false, 4, true>", 64, 168),    # chunks of 3..4 templates
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 12, true, 4, true>", 64, 168),
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 12, false, 8, false>", 64, 168),   # RP_ARITH_FAST_SPLIT
]


@pytest.fixture(scope="module")
def remarks():
    out = {}
    for src in sorted({c[0] for c in CASES}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), src], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        for line in r.stdout.splitlines()[1:]:
            f = line.split(None, 7)
            if len(f) == 8:
                out[(src, re.sub(r"^void rp::", "", f[7]).strip())] = (int(f[0]), int(f[2]))
    return out


def test_the_makefile_hands_the_unroll_budget_to_the_tools():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rustpotter_amd", "csrc"), "print-file-flags", "F=rp_dtw_mfma_wide3.hip"],
                       capture_output=True, text=True, timeout=60)
    assert "-pragma-unroll-threshold" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("src,kernel,max_spill,max_vgpr", CASES)
def test_column_blocks_are_unrolled(remarks, src, kernel, max_spill, max_vgpr):
    assert (src, kernel) in remarks, sorted(k for k in remarks if k[0] == src)
    vgpr, spill = remarks[(src, kernel)]
    assert spill <= max_spill and vgpr <= max_vgpr, "%s: %d registers, %d spilled -- a rolled column block? (csrc/Makefile FILE_FLAGS_%s)" % (kernel, vgpr, spill, src)
