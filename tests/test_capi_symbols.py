"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU,
exports every function include/rustpotter_hip.h declares, and fails loudly (no CPU
fallback) when no HIP device exists."""
import os
import re

import pytest


def _header_functions():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "include", "rustpotter_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import rustpotter_amd
    from rustpotter_amd.api import SYMBOLS
    L = rustpotter_amd.load_library()
    declared = _header_functions()
    assert declared == sorted(SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.rp_version()


def test_default_config_matches_reference_constants():
    import ctypes as C
    import rustpotter_amd
    from rustpotter_amd.api import _Config
    L = rustpotter_amd.load_library()
    c = _Config()
    L.rp_config_default(C.byref(c))
    # src/constants.rs:1-9, src/config.rs:20-29,192-207
    assert c.fmt.sample_rate == 16000 and c.fmt.sample_format == 3 and c.fmt.channels == 1 and c.fmt.endianness == 1
    assert abs(c.detector.avg_threshold - 0.2) < 1e-7 and c.detector.threshold == 0.5 and c.detector.min_scores == 5
    assert abs(c.detector.score_ref - 0.22) < 1e-7 and c.detector.band_size == 5 and c.detector.score_mode == 1
    assert not c.detector.eager and c.detector.vad_mode == 0
    assert not c.filters.gain_normalizer.enabled and not c.filters.band_pass.enabled
    assert c.filters.band_pass.low_cutoff == 80.0 and c.filters.band_pass.high_cutoff == 400.0
    assert L.rp_mfcc_num_frames(18150) == 108 and L.rp_mfcc_num_frames(479) == 0 and L.rp_mfcc_num_frames(64000) == 396


def test_arithmetic_is_part_of_the_abi():
    """The arithmetic of the DTW cost's products and of the model forward is chosen through the ABI (the reference's config-is-a-struct
    convention, src/config.rs:172-219), not through the environment: the header declares the context flags, the RP_ARITH_* values, the setter /
    getter and the reporting bits; the Python harness and the Rust binding carry the same numbers; no product source reads the retired
    RP_DTW_MFMA / RP_DTW_RAGGED switches; a NULL context is refused by the setter like by every other entry point."""
    import ctypes as C
    import rustpotter_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "rustpotter_hip.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"\b(RP_[A-Z0-9_]+)\s*=\s*(\d+)", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))}
    assert (consts["RP_CTX_HOST_POINTERS"], consts["RP_CTX_FULL_SCORES"], consts["RP_CTX_ARITH_STRICT_F32"], consts["RP_CTX_ARITH_FAST_SPLIT"],
            consts["RP_CTX_RAGGED_MATRIX"]) == (1, 2, 4, 8, 16)
    assert (consts["RP_ARITH_F32_MATRIX"], consts["RP_ARITH_STRICT_F32"], consts["RP_ARITH_FAST_SPLIT"]) == (0, 1, 2)   # 0 = the default: f32-grade
    assert (consts["RP_DTW_PRODUCTS_BF16X3"], consts["RP_DTW_PRODUCTS_F16X2"]) == (256, 512)
    assert (consts["RP_MLP_F32"], consts["RP_MLP_BF16"], consts["RP_MLP_F32_STRICT"], consts["RP_MLP_F32_FAST"]) == (0, 1, 2, 3)
    assert rustpotter_amd.BatchContext.ARITH == {"f32_matrix": 0, "strict_f32": 1, "fast_split": 2}
    from rustpotter_amd.api import MLP_PRECISION
    assert MLP_PRECISION == {"f32": 0, "bf16": 1, "f32_strict": 2, "f32_fast": 3}
    rs = open(os.path.join(root, "bindings", "rustpotter_hip.rs")).read()
    for name in ("RP_CTX_ARITH_STRICT_F32", "RP_CTX_ARITH_FAST_SPLIT", "RP_CTX_RAGGED_MATRIX", "RP_ARITH_F32_MATRIX", "RP_ARITH_STRICT_F32",
                 "RP_ARITH_FAST_SPLIT", "RP_DTW_PRODUCTS_BF16X3", "RP_DTW_PRODUCTS_F16X2", "RP_MLP_F32_FAST"):
        m = re.search(r"pub const %s: c_int = (\d+);" % name, rs)
        assert m and int(m.group(1)) == consts[name], name
    L = rustpotter_amd.load_library()
    assert L.rp_ctx_set_arithmetic(None, 0, 0) < 0 and L.rp_ctx_arithmetic(None, None) < 0
    src_dir = os.path.join(root, "rustpotter_amd", "csrc")
    for f in os.listdir(src_dir):
        if f.endswith((".hip", ".cpp", ".h")):
            txt = open(os.path.join(src_dir, f)).read()
            assert 'getenv("RP_DTW_MFMA")' not in txt and 'getenv("RP_DTW_RAGGED")' not in txt, f


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import rustpotter_amd as ra
    with pytest.raises(ra.RustpotterError, match="no usable HIP device"):
        ra.Rustpotter.new(ra.RustpotterConfig.default())
    with pytest.raises(ra.RustpotterError, match="no usable HIP device"):
        ra.BatchContext(device=0)


def test_product_never_references_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "rustpotter_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "rp_oracle" not in txt and "oracle/" not in txt and "import oracle" not in txt, f


def test_null_handles_and_arguments_are_refused():
    """Every entry point called with NULL handles / NULL pointers / zero sizes returns (an error status or a neutral
    value) instead of dereferencing them; run in a child process so that a crash is a test failure, not a dead session."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import rustpotter_amd
from rustpotter_amd.api import SYMBOLS
L = rustpotter_amd.load_library()
bad = []
for name in SYMBOLS:
    fn = getattr(L, name)
    types = fn.argtypes
    if types is None:  # rp_last_error / rp_version take nothing
        assert name in ("rp_last_error", "rp_version"), name
        fn()
        continue
    args = []
    for t in types:
        if t in (C.c_float, C.c_double):
            args.append(0.0)
        elif t in (C.c_int, C.c_uint, C.c_size_t, C.c_int32, C.c_int64, C.c_uint64, C.c_longlong, C.c_uint16, C.c_bool, C.c_long):
            args.append(0)
        else:
            args.append(None)
    r = fn(*args)
    returns_void = name.endswith("_free") or name in ("rp_config_default", "rp_reset")
    if not returns_void and fn.restype is C.c_int and name not in ("rp_templates_max_len", "rp_ctx_dtw_kernels") and any(a is None for a in args) and r != -1:
        bad.append("%%s returned %%r" %% (name, r))
    print(name, r)
assert not bad, bad
print("ALL-OK")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _build_example(name, tmp_path, extra=()):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    lib_dir = os.path.join(root, "rustpotter_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", name + ".c"),
           "-L" + lib_dir, "-lrustpotter_hip", "-Wl,-rpath," + lib_dir, "-o", exe] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_the_c_examples_link(tmp_path):
    """include/rustpotter_hip.h is the boundary document for a host written in any language: it must compile as C99 on its
    own, and the plain-C programs under examples/ (no Python, no torch) must build and link against the shared library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(root, "include", "rustpotter_hip.h")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    _build_example("detect_wav", tmp_path)
    _build_example("sharded_batch", tmp_path, ["-lm"])


@pytest.mark.gpu
def test_c_examples_run(tmp_path):
    """The same programs on the GPU: the reference's golden wav through the single-stream API from C, and the sharded batch
    entry point with three shards (each finds the utterance planted into its first stream)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = os.path.join(root, "tests", "golden")
    exe = _build_example("detect_wav", tmp_path)
    r = subprocess.run([exe, os.path.join(g, "oye_casa_g.rpw"), os.path.join(g, "oye_casa_g_1.wav")], capture_output=True, text=True, timeout=300)
    # the values tests/detector.rs:24-37 asserts for this recording (Max mode): avg_score 0.6495044, score 0.7310586
    assert r.returncode == 0 and 'detection "oye casa" score 0.7310586 avg_score 0.6495044' in r.stdout and "1 detection(s)" in r.stdout, r.stdout + r.stderr
    exe = _build_example("sharded_batch", tmp_path, ["-lm"])
    r = subprocess.run([exe, "3", "100"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "303 streams in 3 shards" in r.stdout, r.stdout + r.stderr
    # round 4: the example also prints how the shards were gathered and runs the same streams through rp_batch_detect_ingest
    assert "gather: gather into host memory" in r.stdout and "rp_batch_detect_ingest: the same" in r.stdout and "(build gfx950)" in r.stdout, r.stdout
    for shard, glob in ((0, 0), (1, 100), (2, 201)):
        assert "shard %d stream 0 (global %d):" % (shard, glob) in r.stdout, r.stdout

