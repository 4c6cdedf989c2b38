"""Robustness of the .rpw reader (rustpotter_amd/csrc/rpw_reader.cpp): wakeword files are untrusted input
(`Rustpotter::add_wakeword_from_buffer`, src/detector.rs:152-176).  CPU: the reader alone under ASan/UBSan over
mutants of the reference's own files.  GPU: mutants through the whole detector."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")
SEEDS = ["alexa.rpw", "ok_casa-tiny.rpw", "oye_casa_g.rpw", "oye_casa_g_v2.rpw", "oye_casa_real.rpw"]


def test_rpw_reader_survives_mutants_under_asan(tmp_path):
    exe = str(tmp_path / "fuzz_rpw")
    csrc = os.path.join(ROOT, "rustpotter_amd", "csrc")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
           os.path.join(HERE, "fuzz", "fuzz_rpw.cpp"), os.path.join(csrc, "rpw_reader.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    r = subprocess.run([exe, "800"] + [os.path.join(G, s) for s in SEEDS], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed, rejected = (int(x) for x in r.stdout.split()[1::2])
    assert parsed > 100 and rejected > 100  # both outcomes are exercised


def _mutants(seed, n, rng):
    for _ in range(n):
        m = bytearray(seed)
        kind = rng.integers(5)
        if kind == 0:
            del m[int(rng.integers(len(m) + 1)):]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 9))):
                m[int(rng.integers(len(m)))] ^= 1 << int(rng.integers(8))
        elif kind == 2:
            for _ in range(int(rng.integers(1, 5))):
                m[int(rng.integers(min(len(m), 512)))] = int(rng.integers(256))
        elif kind == 3:
            at = int(rng.integers(len(m)))
            m[at] = (0x9b, 0x5b, 0x7b, 0xbb)[int(rng.integers(4))]
            m[at + 1:at + 9] = b"\xff" * len(m[at + 1:at + 9])
        else:
            ln, a, b = int(rng.integers(1, 65)), int(rng.integers(len(m))), int(rng.integers(len(m)))
            piece = m[a:a + ln]
            m[b:b + len(piece)] = piece
            del m[len(seed):]
        yield bytes(m)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_rpw_mutants_through_the_detector(seed):
    """Every mutant is either refused with an error or accepted and then scored without faulting: 120 mutants per seed,
    each accepted one sees 60 chunks of a recording."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import rustpotter_amd as ra
    import rpw_py
    pcm, rate = rpw_py.read_wav_i16(os.path.join(G, "oye_casa_g_1.wav"))
    assert rate == 16000
    pcm = np.tile(pcm, 2)  # 75 chunks: longer than any template window
    cfg = ra.RustpotterConfig.default()
    cfg.fmt.sample_format = ra.SampleFormat.I16
    data = open(os.path.join(G, seed), "rb").read()
    rng = np.random.default_rng(SEEDS.index(seed))
    accepted = refused = 0
    for m in _mutants(data, 120, rng):
        rp = ra.Rustpotter.new(cfg)
        try:
            rp.add_wakeword_from_buffer("w", m)
        except Exception as e:  # noqa: BLE001 - the error text is the API's Result::Err
            assert str(e)
            refused += 1
            continue
        accepted += 1
        for c in range(60):
            rp.process_samples(pcm[480 * c:480 * (c + 1)])
    assert accepted + refused == 120 and refused > 0


@pytest.mark.gpu
def test_wav_mutants_through_the_builder():
    """WakewordRef::new_from_sample_buffers on damaged wav files (header fields, chunk sizes, truncations): an error
    text or a wakeword, never a fault.  src/mfcc/wav_file_extractor.rs:18-69,93-112."""
    sys.path.insert(0, ROOT)
    import rustpotter_amd as ra
    ctx = ra.BatchContext(0)
    good = open(os.path.join(G, "oye_casa_g_1.wav"), "rb").read()
    other = open(os.path.join(G, "oye_casa_g_2.wav"), "rb").read()
    rng = np.random.default_rng(11)
    built = refused = 0
    for i in range(160):
        m = bytearray(good)
        kind = i % 4
        if kind == 0:
            del m[int(rng.integers(len(m) + 1)):]
        elif kind == 1:  # header fields: format tag, channels, rate, bits, chunk ids and sizes
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(min(len(m), 48)))] = int(rng.integers(256))
        elif kind == 2:  # chunk sizes
            at = (4, 16, 40)[int(rng.integers(3))]
            m[at:at + 4] = int(rng.integers(1 << 32)).to_bytes(4, "little")
        else:
            for _ in range(int(rng.integers(1, 9))):
                m[int(rng.integers(len(m)))] ^= 1 << int(rng.integers(8))
        try:
            data = ctx.build_wakeword_ref("w", {"a.wav": bytes(m), "b.wav": other}, 5, from_files=False)
        except Exception as e:  # noqa: BLE001
            assert str(e)
            refused += 1
            continue
        assert len(data) > 0
        built += 1
    assert built > 0 and refused > 0
