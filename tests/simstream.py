"""Builds the reference's detector simulation stream (tests/detector.rs:372-426):
5 s zeros + oye_casa_g_1.wav + 5 s zeros + oye_casa_g_2.wav + 5 s zeros as i16,
wav header stripped by a fixed 44 bytes, optional per-sample gain with
f32::round (half away from zero) + clamp."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read_raw(path, gain):
    with open(path, "rb") as f:
        b = f.read()[44:]
    a = np.frombuffer(b[: len(b) // 2 * 2], dtype="<i2").astype(np.float32) * np.float32(gain)
    a = np.sign(a) * np.floor(np.abs(a) + np.float32(0.5))  # f32::round
    return np.clip(a, -32768, 32767).astype(np.int16)


def simulation_stream_i16(gain1=1.0, gain2=1.0):
    z = np.zeros(16000 * 5, np.int16)
    return np.concatenate([z, _read_raw(os.path.join(GOLDEN, "oye_casa_g_1.wav"), gain1), z,
                           _read_raw(os.path.join(GOLDEN, "oye_casa_g_2.wav"), gain2), z])


def i16_to_f32(a):
    """v as f32 / i16::MAX as f32 (src/audio/audio_types.rs:108-117)."""
    return a.astype(np.float32) / np.float32(32767.0)
