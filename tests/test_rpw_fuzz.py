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


def _model_with_dims(dims, n_values):
    """A Tiny wakeword model whose ln1.weight claims `dims` but carries n_values f32 values."""
    sys.path.insert(0, HERE)
    import rpw_py
    out = rpw_py._head(5, 6)
    out += rpw_py._text("labels") + rpw_py._head(4, 2) + rpw_py._text("none") + rpw_py._text("w")
    out += rpw_py._text("train_size") + rpw_py._head(0, 4) + rpw_py._text("mfcc_size") + rpw_py._head(0, 2)
    out += rpw_py._text("m_type") + rpw_py._text("Tiny") + rpw_py._text("weights") + rpw_py._head(5, 4)

    def tensor(name, dims_, values):
        raw = np.asarray(values, "<f4").tobytes()
        return (rpw_py._text(name) + rpw_py._head(5, 3) + rpw_py._text("bytes") + rpw_py._head(4, len(raw)) +
                b"".join(rpw_py._head(0, v) for v in raw) + rpw_py._text("dims") + rpw_py._head(4, len(dims_)) +
                b"".join(rpw_py._head(0, d) for d in dims_) + rpw_py._text("d_type") + rpw_py._text("f32"))
    out += tensor("ln1.weight", dims, np.ones(n_values)) + tensor("ln1.bias", [dims[0]], np.zeros(dims[0] if dims[0] < 64 else 1))
    out += tensor("ln2.weight", [2, dims[0] if dims[0] < 64 else 3], np.ones(2 * (dims[0] if dims[0] < 64 else 3)))
    out += tensor("ln2.bias", [2], np.zeros(2))
    out += rpw_py._text("rms_level") + rpw_py._f32(0.0)
    return out


# dims whose product wraps modulo 2^64 to the number of values actually present (3 * 0x5555555555555556 * 4
# bytes == 8 bytes mod 2^64), a dimension beyond int, and a plain mismatch
CRAFTED = [([3, 0x5555555555555556], 2), ([1 << 40, 1], 1), ([0x80000000, 2], 0), ([3, 8], 23)]


def test_rpw_reader_refuses_wrapped_tensor_dims(tmp_path):
    exe = str(tmp_path / "fuzz_rpw")
    csrc = os.path.join(ROOT, "rustpotter_amd", "csrc")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
           os.path.join(HERE, "fuzz", "fuzz_rpw.cpp"), os.path.join(csrc, "rpw_reader.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    good = tmp_path / "good.rpw"
    good.write_bytes(_model_with_dims([3, 8], 24))
    r = subprocess.run([exe, "check", str(good)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "parsed", r.stdout + r.stderr
    for i, (dims, n) in enumerate(CRAFTED):
        f = tmp_path / ("crafted%d.rpw" % i)
        f.write_bytes(_model_with_dims(dims, n))
        r = subprocess.run([exe, "check", str(f)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.startswith("rejected: "), (dims, r.stdout + r.stderr)


@pytest.mark.gpu
def test_crafted_model_dims_through_the_detector():
    """The same files through Rustpotter::add_wakeword_from_buffer: an error text, no 17 GB copy from an 8-byte vector."""
    sys.path.insert(0, ROOT)
    import rustpotter_amd as ra
    cfg = ra.RustpotterConfig.default()
    rp = ra.Rustpotter.new(cfg)
    rp.add_wakeword_from_buffer("ok", _model_with_dims([3, 8], 24))
    for dims, n in CRAFTED:
        with pytest.raises(ra.RustpotterError):
            ra.Rustpotter.new(cfg).add_wakeword_from_buffer("w", _model_with_dims(dims, n))


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
