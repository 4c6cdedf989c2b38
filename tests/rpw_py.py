"""Minimal CBOR reader for rustpotter `.rpw` files -- test-side only.

Independent of the product's C++ reader (rustpotter_amd/csrc/rpw_reader.cpp) so
that each checks the other.  Wire format: SURVEY.md §8(c); structs
src/wakewords/wakeword_ref.rs:12-20, wakeword_v2.rs:8-16, wakeword_model.rs:11-18,68-72.
"""
import struct
from collections import OrderedDict

import numpy as np


def _half(b):
    return float(np.frombuffer(b, dtype=">f2")[0])


def _dec(buf, p):
    ib = buf[p]
    p += 1
    major, info = ib >> 5, ib & 31
    if major == 7:
        if info == 20:
            return False, p
        if info == 21:
            return True, p
        if info in (22, 23):
            return None, p
        if info == 25:
            return _half(buf[p:p + 2]), p + 2
        if info == 26:
            return struct.unpack(">f", buf[p:p + 4])[0], p + 4
        if info == 27:
            return struct.unpack(">d", buf[p:p + 8])[0], p + 8
        raise ValueError("unsupported simple value %d" % info)
    if info < 24:
        val = info
    elif info == 24:
        val = buf[p]
        p += 1
    elif info == 25:
        val = struct.unpack(">H", buf[p:p + 2])[0]
        p += 2
    elif info == 26:
        val = struct.unpack(">I", buf[p:p + 4])[0]
        p += 4
    elif info == 27:
        val = struct.unpack(">Q", buf[p:p + 8])[0]
        p += 8
    else:
        raise ValueError("indefinite lengths are not used by ciborium for these structs")
    if major == 0:
        return val, p
    if major == 1:
        return -1 - val, p
    if major == 2:
        return bytes(buf[p:p + val]), p + val
    if major == 3:
        return bytes(buf[p:p + val]).decode("utf-8"), p + val
    if major == 4:
        out = []
        for _ in range(val):
            v, p = _dec(buf, p)
            out.append(v)
        return out, p
    if major == 5:
        out = OrderedDict()
        for _ in range(val):
            k, p = _dec(buf, p)
            v, p = _dec(buf, p)
            out[k] = v
        return out, p
    raise ValueError("unsupported major type %d" % major)


def load_rpw(path):
    """Returns a dict; 'kind' in {'ref', 'v2', 'model'}; matrices as float32 ndarrays;
    samples_features keeps FILE order (the Rust HashMap order it was written in)."""
    with open(path, "rb") as f:
        buf = f.read()
    obj, p = _dec(buf, 0)
    assert p == len(buf), "trailing bytes"
    if "labels" in obj:
        w = OrderedDict()
        for name, td in obj["weights"].items():
            raw = td["bytes"]
            raw = bytes(raw) if isinstance(raw, (list, bytes)) else raw
            assert td["d_type"] == "f32"
            w[name] = np.frombuffer(raw, dtype="<f4").reshape(td["dims"]).copy()
        return {"kind": "model", "labels": obj["labels"], "train_size": obj["train_size"],
                "mfcc_size": obj["mfcc_size"], "m_type": obj["m_type"], "weights": w,
                "rms_level": obj["rms_level"]}
    sf = OrderedDict((k, np.asarray(v, dtype=np.float32)) for k, v in obj["samples_features"].items())
    avg = obj.get("avg_features")
    out = {"kind": "v2" if "enabled" in obj else "ref", "name": obj["name"],
           "avg_features": None if avg is None else np.asarray(avg, dtype=np.float32),
           "samples_features": sf, "threshold": obj.get("threshold"), "avg_threshold": obj.get("avg_threshold"),
           "rms_level": obj["rms_level"],
           "mfcc_size": obj.get("mfcc_size", next(iter(sf.values())).shape[1])}
    return out


def read_wav(path):
    """Any PCM int (8/16/32 bit) or IEEE f32 wav -> (ndarray in the file's sample type, interleaved; sample_rate,
    channels); walks RIFF chunks, understands WAVE_FORMAT_EXTENSIBLE."""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    p, fmt, data = 12, None, None
    while p + 8 <= len(b):
        cid, sz = b[p:p + 4], struct.unpack("<I", b[p + 4:p + 8])[0]
        if cid == b"fmt ":
            fmt = list(struct.unpack("<HHIIHH", b[p + 8:p + 24]))
            if fmt[0] == 0xFFFE:  # extensible: the sub-format GUID starts with the real format tag
                fmt[0] = struct.unpack("<H", b[p + 8 + 24:p + 8 + 26])[0]
        elif cid == b"data":
            data = b[p + 8:p + 8 + sz]
        p += 8 + sz + (sz & 1)
    assert fmt is not None and data is not None
    tag, ch, sr, bits = fmt[0], fmt[1], fmt[2], fmt[5]
    dt = {(1, 8): "u1", (1, 16): "<i2", (1, 32): "<i4", (3, 32): "<f4"}[(tag, bits)]
    return np.frombuffer(data, dtype=dt).copy(), sr, ch


def read_wav_i16(path):
    """PCM i16 mono wav -> (int16 ndarray, sample_rate); walks RIFF chunks."""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    p, fmt, data = 12, None, None
    while p + 8 <= len(b):
        cid, sz = b[p:p + 4], struct.unpack("<I", b[p + 4:p + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[p + 8:p + 24])
        elif cid == b"data":
            data = b[p + 8:p + 8 + sz]
        p += 8 + sz + (sz & 1)
    assert fmt is not None and data is not None and fmt[0] == 1 and fmt[1] == 1 and fmt[5] == 16
    return np.frombuffer(data, dtype="<i2").copy(), fmt[2]


def _head(major, n):
    if n < 24:
        return bytes([major << 5 | n])
    for info, fmt in ((24, ">B"), (25, ">H"), (26, ">I"), (27, ">Q")):
        if n < 1 << (8 * struct.calcsize(fmt)):
            return bytes([major << 5 | info]) + struct.pack(fmt, n)
    raise ValueError(n)


def _text(t):
    b = t.encode()
    return _head(3, len(b)) + b


def _f32(v):
    return b"\xfa" + struct.pack(">f", float(v))


def _matrix(m):
    m = np.asarray(m, np.float32)
    return _head(4, m.shape[0]) + b"".join(_head(4, m.shape[1]) + b"".join(_f32(v) for v in row) for row in m)


def dump_rpw_ref(name, samples_features, avg_features=None, threshold=None, avg_threshold=None, rms_level=0.0, mfcc_size=None):
    """WakewordRef as ciborium writes it (src/wakewords/wakeword_ref.rs:12-20): map(7) in struct order; floats are
    written as single precision (a reader must accept any float width)."""
    names = list(samples_features.keys())
    K = int(np.asarray(samples_features[names[0]]).shape[1]) if mfcc_size is None else mfcc_size
    out = _head(5, 7)
    out += _text("name") + _text(name)
    out += _text("avg_features") + (b"\xf6" if avg_features is None else _matrix(avg_features))
    out += _text("samples_features") + _head(5, len(names)) + b"".join(_text(n) + _matrix(samples_features[n]) for n in names)
    out += _text("threshold") + (b"\xf6" if threshold is None else _f32(threshold))
    out += _text("avg_threshold") + (b"\xf6" if avg_threshold is None else _f32(avg_threshold))
    out += _text("rms_level") + _f32(rms_level)
    out += _text("mfcc_size") + _head(0, K)
    return out


def dump_rpw_model(labels, train_size, mfcc_size, m_type, weights, rms_level=float("nan")):
    """WakewordModel as ciborium writes it (src/wakewords/wakeword_model.rs:11-18): tensors are
    {bytes: array of u8, dims: array, d_type: "f32"} (little-endian f32), weights in the given order."""
    out = _head(5, 6)
    out += _text("labels") + _head(4, len(labels)) + b"".join(_text(l) for l in labels)
    out += _text("train_size") + _head(0, int(train_size))
    out += _text("mfcc_size") + _head(0, int(mfcc_size))
    out += _text("m_type") + _text(m_type)
    out += _text("weights") + _head(5, len(weights))
    for name, w in weights.items():
        w = np.ascontiguousarray(w, "<f4")
        raw = w.tobytes()
        out += _text(name) + _head(5, 3)
        out += _text("bytes") + _head(4, len(raw)) + b"".join(_head(0, v) for v in raw)
        out += _text("dims") + _head(4, w.ndim) + b"".join(_head(0, d) for d in w.shape)
        out += _text("d_type") + _text("f32")
    out += _text("rms_level") + _f32(rms_level)
    return out
