"""bench.py's VALU-issue roofline is priced from the committed instruction mixes of the hot loops (profiles/*_isa_mix.json,
made by tools/isa_mix.py from the compiler's gfx950 assembly x profiles/valu_rate_table.json).  A kernel edit that changes a hot
loop must regenerate them: this test recompiles the kernels (CPU only, hipcc cross-compiles) and compares."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    # the headline kernel: three bf16 parts per operand (RP_ARITH_F32_MATRIX), two matrix instructions per tile and column
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 8, false, 8, true>", ["--min-mfma", "72", "--flags", "-DRP_MFMA_PRICE_NO_ABANDON"], "profiles/dtw_mfma_isa_mix.json"),
    ("rp_dtw_mfma.hip", "dtw_mfma_kernel<5, 12, false, 8, false>", ["--min-mfma", "36", "--flags", "-DRP_MFMA_PRICE_NO_ABANDON"], "profiles/dtw_mfma_f16x2_isa_mix.json"),
    ("rp_mfcc.hip", "mfcc_kernel<true, 6, float, false>", [], "profiles/mfcc_isa_mix.json"),
    ("rp_dtw_ragged.hip", "dtw_ragged_kernel<5>", ["--min-mfma", "32"], "profiles/dtw_ragged_isa_mix.json"),
    ("rp_dtw_mfma_wide.hip", "dtw_mfma_wide_kernel<16, 5, 8>", ["--min-mfma", "100"], "profiles/dtw_mfma_wide_isa_mix.json"),
    ("rp_dtw_mfma_wide3.hip", "dtw_mfma_wide3_kernel<16, 5, 8>", ["--min-mfma", "192"], "profiles/dtw_mfma_wide3_isa_mix.json"),
    ("rp_dtw_mfma_group.hip", "dtw_mfma_group_kernel<5, 4>", ["--min-mfma", "36"], "profiles/dtw_mfma_group_isa_mix.json"),
]


@pytest.mark.parametrize("src,kernel,extra,committed", CASES)
def test_committed_isa_mix_matches_the_source(src, kernel, extra, committed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), src, kernel] + extra, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    now = json.loads(out.stdout)
    was = json.load(open(os.path.join(ROOT, committed)))
    for key in ("instructions", "classes", "valu_issue_cycles_per_trip", "lds_cycles_per_trip", "opcodes"):
        assert now[key] == was[key], "%s changed: run `python tools/isa_mix.py %s '%s' %s --out %s`" % (key, src, kernel, " ".join("'%s'" % e if " " in e or e.startswith("-D") else e for e in extra), committed)


def test_rate_table_prices_every_valu_opcode_of_the_mixes():
    import re
    table = json.load(open(os.path.join(ROOT, "profiles", "valu_rate_table.json")))
    for _, _, _, committed in CASES:
        mix = json.load(open(os.path.join(ROOT, committed)))
        total = 0.0
        for op, n in mix["opcodes"].items():
            if op.startswith("v_") and not op.startswith("v_mfma"):
                rate = next((c for pat, c in table["rates"] if re.fullmatch(pat, op)), table["default_valu"])
                assert 2.0 <= rate <= 9.0
                total += n * rate
        assert abs(total - mix["valu_issue_cycles_per_trip"]) < 0.5
