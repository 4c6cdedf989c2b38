"""Why three MFCC gates in this suite are wider than SURVEY 8d's element-wise 1e-5 -- proven here, not in tools/: for mfcc_size >= 16
(tests/test_gpu_parity.py:42-48: 1e-5 of the frame's largest coefficient), steady tones (1e-4) and speech-like signals (3e-5,
tests/sweep_parity.py run_mfcc_sweep) the f32 ORACLE itself is as far from the mathematically exact value of the same formula as the
kernel is.  Each test evaluates the pipeline of src/mfcc/extractor.rs:60-198 in f64 (same f32 tables, same f32-rounded pre-emphasis as the
reference keeps it) and asserts that the kernel is as good an f32 evaluation as the oracle: its rms error against the f64 value is at
most 1.15 x the oracle's over the whole array (measured 1.02-1.06 on every signal family, tools/scratch/diag_f64.py) and 1.4 x per
cepstral coefficient, its largest error at most 2.5 x the oracle's largest (two maxima over ~1 000 frames of equally distributed errors
differ by that much by chance: 0.4-2.2 measured), wherever the two differ by more than the strict gate the kernel is no further from the
f64 value than 2.5 x the oracle's worst error on that coefficient, and an element beyond even the loosened gate is one where the kernel
is the CLOSER of the two (steady tones: the oracle itself is up to 1.5e-4 from the exact value).  A kernel defect (wrong twiddle, lost
bin, order-of-magnitude worse rounding) fails these whatever the loosened gates allow."""
import numpy as np
import pytest

from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


@pytest.fixture(scope="module")
def ctx(ra):
    return ra.BatchContext(device=0, host_pointers=True)


def f64_mfcc(pcm, K):
    """The reference's formula in double precision: pre-emphasis restarted every 160 samples and rounded to f32 as the reference stores it
    (extractor.rs:87-97), Hamming window (f32 table), DFT-480, |X|^2 on bins 0..239, mel bank (f32 table, no normalisation), ln(x +
    f32::MIN_POSITIVE), un-normalised DCT-II x 2 (f32 cosine table), c0 dropped; frame j = shifts j+1..j+3."""
    ham = orc.hamming_window().astype(np.float64)
    fb, _ = orc.mel_filter_bank(K)
    dct = orc.dct_table(K).astype(np.float64)
    nch = len(pcm) // 480
    x = np.asarray(pcm[: nch * 480], np.float32).astype(np.float64).reshape(-1, 160)
    pre = x.copy()
    pre[:, 1:] = x[:, 1:] - np.float64(np.float32(0.97)) * x[:, :-1]
    pre = np.float32(pre).astype(np.float64).reshape(-1)
    out = []
    for j in range(3 * nch - 3):
        fr = pre[(j + 1) * 160:(j + 4) * 160] * ham
        P = np.abs(np.fft.fft(fr)[:240]) ** 2
        lg = np.log(fb.astype(np.float64) @ P + np.finfo(np.float32).tiny)
        out.append(2 * (dct @ lg)[1:])
    return np.array(out)


def _tones(rng, n):
    t = np.arange(n) / 16000.0
    return sum(np.sin(2 * np.pi * rng.uniform(50, 7900) * t + rng.uniform(0, 6.28)) for _ in range(int(rng.integers(1, 4))))


def _utterance(rng, n):
    t = np.arange(n, dtype=np.float64) / 16000.0
    x = np.zeros(n)
    for _ in range(int(rng.integers(2, 6))):
        f0, f1 = rng.uniform(120, 3500, 2)
        ph = 2 * np.pi * np.cumsum(np.linspace(f0, f1, n)) / 16000.0
        env = np.interp(t, np.linspace(0, t[-1], 8), rng.uniform(0.0, 1.0, 8))
        x += rng.uniform(0.05, 0.3) * env * np.sin(ph + rng.uniform(0, 6.28))
    return x + rng.standard_normal(n) * 0.003


def _compare(ctx, signals, K, loose_gate, framescale):
    """kernel / oracle / f64 on every signal; returns the largest kernel-vs-oracle error in units of the strict gate."""
    got, ref, tru = [], [], []
    for x in signals:
        x = np.ascontiguousarray(x, np.float32)
        got.append(ctx.mfcc(x[None, :], K)[0].astype(np.float64))
        ref.append(orc.mfcc_stream(x, K).astype(np.float64))
        tru.append(f64_mfcc(x, K))
    got, ref, tru = np.concatenate(got), np.concatenate(ref), np.concatenate(tru)
    assert got.shape == ref.shape == tru.shape and got.shape[0] > 100
    ek, eo = np.abs(got - tru), np.abs(ref - tru)
    # the kernel is no worse an f32 evaluation than the oracle: rms error overall and per coefficient, largest error overall
    rk, ro = np.sqrt((ek ** 2).mean(axis=0)), np.sqrt((eo ** 2).mean(axis=0))
    assert np.sqrt((ek ** 2).mean()) <= 1.15 * np.sqrt((eo ** 2).mean()), np.sqrt((ek ** 2).mean()) / np.sqrt((eo ** 2).mean())
    assert np.all(rk <= 1.4 * ro), (rk / ro).max()
    assert ek.max() <= 2.5 * eo.max(), ek.max() / eo.max()
    # the loosened gate itself -- an element beyond it must be one where the kernel is the closer of the two -- and where the strict gate
    # is exceeded the kernel stays inside the oracle's own error band
    strict = 1e-5 * np.maximum(np.abs(ref), 1.0)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0) if framescale else np.maximum(np.abs(ref), 1.0)
    viol = np.abs(got - ref) > loose_gate * scale
    assert np.all(ek[viol] <= eo[viol]), float((np.abs(got - ref) / scale).max())
    beyond = np.abs(got - ref) > strict
    assert np.all(ek[beyond] <= 2.5 * np.broadcast_to(eo.max(axis=0), ek.shape)[beyond])
    return float((np.abs(got - ref) / strict).max()), float(eo.max()), float(ek.max())


@pytest.mark.parametrize("K", [16, 23, 40])
def test_large_mfcc_sizes_the_oracle_is_as_far_from_f64_as_the_kernel(ctx, K):
    """tests/test_gpu_parity.py mfcc_close_framescale: the DCT sums reach |60| (one ulp 3.8e-6) while small coefficients are ~1."""
    sig = [orc.synth_pcm(SEED, s, 480 * 60) for s in range(6)]
    over, eo, ek = _compare(ctx, sig, K, 1e-5, framescale=True)
    assert eo > 4e-6   # the premise: the f32 oracle itself misses the exact value by about the strict gate on |c| ~ 1


def test_steady_tones(ctx):
    """run_mfcc_sweep kind 2 (gate 1e-4): the far mel filters sit 60-90 dB under the peak, where the rounding noise of ANY f32 FFT is no
    longer small against the local energy and the logarithm turns it into absolute differences above 1e-5."""
    rng = np.random.default_rng(2)
    sig = [_tones(rng, 480 * 40) * 10.0 ** rng.uniform(-1.5, 0.3) for _ in range(8)]
    over, eo, ek = _compare(ctx, sig, 5, 1e-4, framescale=False)
    assert eo > 1e-5   # the premise: the oracle's own distance from the exact value exceeds the strict gate on these signals


def test_speech_like_signals(ctx):
    """run_mfcc_sweep kind 3 (gate 3e-5)."""
    rng = np.random.default_rng(3)
    sig = [_utterance(rng, 480 * 40) * 10.0 ** rng.uniform(-1.5, 0.3) for _ in range(8)]
    _compare(ctx, sig, 5, 3e-5, framescale=False)


def test_broadband_noise_keeps_the_strict_gate(ctx):
    """The default-size path on broadband input needs no allowance: element-wise 1e-5 * max(|ref|, 1), and the same f64 comparison."""
    sig = [orc.synth_pcm(SEED, 40 + s, 480 * 60) for s in range(4)]
    over, _, _ = _compare(ctx, sig, 5, 1e-5, framescale=False)
    assert over <= 1.0
