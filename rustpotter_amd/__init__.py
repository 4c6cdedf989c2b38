"""rustpotter_amd -- MI355X (gfx950) HIP implementation of rustpotter's MFCC + DTW
wakeword scoring path behind a C ABI (include/rustpotter_hip.h).

This Python package is a thin ctypes binding used by the tests and by bench.py; the
product is librustpotter_hip.so (rustpotter_amd/csrc).  There is no CPU fallback: if
the shared library is missing, or no HIP device is usable, the calls raise.
"""
from .api import (  # noqa: F401
    AudioFmt, BandPassConfig, BatchContext, DetectorConfig, Endianness, FiltersConfig,
    GainNormalizationConfig, Model, Rustpotter, RustpotterConfig, RustpotterDetection, RustpotterError,
    SampleFormat, ScoreMode, StreamBatch, Templates, VADMode, arithmetic_all, batch_detect_sharded, batch_detect_sharded_dev, build_info, sharded_gather_info, lib_path, load_library, mfcc_num_frames, resampler_frame_lengths,
)
