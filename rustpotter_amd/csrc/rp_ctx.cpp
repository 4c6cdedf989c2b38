// rp_ctx.cpp -- device context, buffers, template upload, per-kernel event timing.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <mutex>
#include <type_traits>

#include "rp_host.h"

namespace rp {

hipError_t allow_dynamic_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> done;  // (device, kernel) -> bytes granted
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find({dev, kernel});
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[{dev, kernel}] = bytes;
    return e;
}


static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    if (e == hipErrorMemoryAllocation && std::strncmp(what, "dtw", 3) == 0)   // launch_dtw: the band and one template length of frames
        set_last_error(std::string("HIP error in ") + what + ": band_size and template length need more than the 160 KB of LDS a compute "
                       "unit has (lower band_size or shorten the wakeword's templates)");
    else
        set_last_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
    (void)hipGetLastError();  // the runtime keeps the code as its "last error": clear it, or the next kernel launch
                              // (checked with hipGetLastError) would report this failure again
    return false;
}

DevBuf::~DevBuf() { if (p) (void)hipFree(p); }
bool DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    if (!hip_ok(hipMalloc(&p, want), "hipMalloc")) { p = nullptr; return false; }
    cap = want;
    return true;
}

int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
    return n;
}

Ctx *Ctx::create(int device, int flags) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("no usable HIP device: librustpotter_hip has no CPU fallback (" +
                       std::string(e == hipSuccess ? "device count 0" : hipGetErrorString(e)) + ")");
        return nullptr;
    }
    if (device < 0 || device >= n) { set_last_error("HIP device ordinal out of range"); return nullptr; }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    std::unique_ptr<Ctx> c(new Ctx());
    c->device = device;
    c->flags = flags;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cu = cus;
    if (!hip_ok(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    c->stream = c->own_stream;
    // the DTW launchers' per-call words (DtwWork, rp_kernels.h): zero between launches, so zeroed once here
    const size_t wbytes = ((size_t)2 * kDtwSchedChunks + 2 + 2 * (size_t)kDtwFixCap + 2) * sizeof(uint32_t);
    if (!c->ws_dtw.reserve(wbytes) || !hip_ok(hipMemset(c->ws_dtw.p, 0, wbytes), "hipMemset(dtw work)")) return nullptr;
    return c.release();
}

bool Ctx::ingest_ready() {
    if (copy_stream) return true;   // set only when the stream AND its four events exist (a half-made set is torn down below)
    hipStream_t cs = nullptr;
    if (!hip_ok(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreate(copy)")) return false;
    bool ok = true;
    for (int b = 0; b < 2 && ok; ++b)
        ok = hip_ok(hipEventCreateWithFlags(&ingest_landed[b], hipEventDisableTiming), "hipEventCreate") &&
             hip_ok(hipEventCreateWithFlags(&ingest_freed[b], hipEventDisableTiming), "hipEventCreate");
    if (!ok) {
        const std::string why = last_error();
        for (int b = 0; b < 2; ++b) {
            if (ingest_landed[b]) { (void)hipEventDestroy(ingest_landed[b]); ingest_landed[b] = nullptr; }
            if (ingest_freed[b]) { (void)hipEventDestroy(ingest_freed[b]); ingest_freed[b] = nullptr; }
        }
        (void)hipStreamDestroy(cs);
        set_last_error(why);
        return false;
    }
    copy_stream = cs;
    return true;
}

DtwWork Ctx::dtw_work_for(size_t S, size_t rows) {
    DtwWork wk = dtw_work();
    if (!(arith.mode == kArithFastSplit && arith.ragged)) return wk;   // the opt-in kernel's blocks are only reserved when it can be taken
    const size_t prep_bytes = (S * 8 * sizeof(float) + 255) & ~(size_t)255, list_bytes = (rows + 1) * sizeof(uint32_t);
    if (S && rows && rows <= 0xffffffffULL && ws_rag.reserve(prep_bytes + list_bytes + 16)) {
        wk.rag_prep = ws_rag.as<float>();
        wk.rag_list = reinterpret_cast<uint32_t *>(ws_rag.as<unsigned char>() + prep_bytes);
        wk.rag_streams = S;
        wk.rag_rows = rows;
    }
    return wk;
}

uint32_t *Ctx::hot_flags(size_t S) {
    const size_t bytes = S * sizeof(uint32_t) + 16;
    if (bytes > ws_hot.cap) {
        if (!ws_hot.reserve(bytes) || !hip_ok(hipMemsetAsync(ws_hot.p, 0, ws_hot.cap, stream), "hipMemsetAsync(hot)")) return nullptr;
    }
    return ws_hot.as<uint32_t>();
}

uint32_t *Ctx::mlp_redo(size_t B) {
    const size_t bytes = (2 + B) * sizeof(uint32_t);
    if (bytes > ws_mlp_redo.cap) {   // a new block: its counters start at zero (later calls leave them so)
        if (!ws_mlp_redo.reserve(bytes) || !hip_ok(hipMemsetAsync(ws_mlp_redo.p, 0, 2 * sizeof(uint32_t), stream), "hipMemsetAsync(mlp redo)")) return nullptr;
    }
    return ws_mlp_redo.as<uint32_t>();
}

Ctx::~Ctx() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto &t : pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto &kv : tables) {
        MfccTablesDev &t = kv.second;
        (void)hipFree(t.hamming); (void)hipFree(t.tw240); (void)hipFree(t.tw480); (void)hipFree(t.fb); (void)hipFree(t.dct); if (t.melw) (void)hipFree(t.melw);
    }
    if (own_stream) (void)hipStreamDestroy(own_stream);
    for (int b = 0; b < 2; ++b) {
        if (ingest_landed[b]) (void)hipEventDestroy(ingest_landed[b]);
        if (ingest_freed[b]) (void)hipEventDestroy(ingest_freed[b]);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
}

template <class T> static bool upload(T **dst, const std::vector<T> &src) {
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(dst), src.size() * sizeof(T)), "hipMalloc(table)")) return false;
    return hip_ok(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(table)");
}

PinBuf::~PinBuf() { if (p) (void)hipHostFree(p); }
bool PinBuf::reserve(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) { (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    if (!hip_ok(hipHostMalloc(&p, want, hipHostMallocDefault), "hipHostMalloc")) { p = nullptr; return false; }
    if (!hip_ok(hipHostGetDevicePointer(&dev, p, 0), "hipHostGetDevicePointer")) { (void)hipHostFree(p); p = nullptr; return false; }
    cap = want;
    return true;
}

const Resampler *Ctx::resampler_for(size_t fs_in) {
    auto it = resamplers.find(fs_in);
    if (it != resamplers.end()) return it->second.get();
    std::unique_ptr<Resampler> r(Resampler::create(this, fs_in));
    if (!r) return nullptr;
    return (resamplers[fs_in] = std::move(r)).get();
}

const MfccTablesDev *Ctx::tables_for(int K) {
    auto it = tables.find(K);
    if (it != tables.end()) return &it->second;
    if (K < 1 || K > 63) { set_last_error("mfcc_size out of the supported range 1..63"); return nullptr; }
    HostTables h = build_tables(K);
    MfccTablesDev d;
    d.K1 = h.K1;
    // the compile-time sparsity of the mfcc_size 5 / 16 kernels must cover the table
    auto covered = [&](auto touches) {
        for (int f = 0; f < h.K1; ++f)
            for (int k = 0; k < kBins; ++k) {
                if (h.fb[(size_t)f * kBins + k] == 0.f) continue;
                if (!(k <= 120 ? touches(f, k / 16, false) : touches(f, (240 - k) / 16, true))) return false;
            }
        return true;
    };
    if (h.K1 == 6) d.mel_sparse = covered([](int f, int k2, bool m) { return mel_touches<6>(f, k2, m); });
    if (h.K1 == 14) d.mel_sparse = covered([](int f, int k2, bool m) { return mel_touches<14>(f, k2, m); });
    if (h.K1 == 17) d.mel_sparse = covered([](int f, int k2, bool m) { return mel_touches<17>(f, k2, m); });
    if (!upload(&d.hamming, h.hamming) || !upload(&d.tw240, h.tw240) || !upload(&d.tw480, h.tw480) ||
        !upload(&d.fb, h.fb) || !upload(&d.dct, h.dct))
        return nullptr;
    if (d.mel_sparse) {  // compact per-lane rows for the 16-byte LDS reads of the sparse kernels
        auto build = [&](auto k1c) {
            constexpr int K1C = decltype(k1c)::value;
            std::vector<float> w((size_t)16 * kMelRowPitch<K1C>, 0.f);
            for (int l = 0; l < 16; ++l)
                for (int f = 0; f < K1C; ++f)
                    for (int k2 = 0; k2 < 8; ++k2) {
                        const int k = l + 16 * k2;
                        if (mel_touches<K1C>(f, k2, false)) w[(size_t)l * kMelRowPitch<K1C> + mel_index<K1C>(f, k2, false)] = h.fb[(size_t)f * kBins + k];
                        // bin 240 - k; for k == 0 it does not exist (the kernel forces that power to 0)
                        if (mel_touches<K1C>(f, k2, true)) w[(size_t)l * kMelRowPitch<K1C> + mel_index<K1C>(f, k2, true)] = k > 0 ? h.fb[(size_t)f * kBins + (240 - k)] : 0.f;
                    }
            return w;
        };
        const std::vector<float> w = h.K1 == 6 ? build(std::integral_constant<int, 6>{}) : h.K1 == 14 ? build(std::integral_constant<int, 14>{})
                                                 : build(std::integral_constant<int, 17>{});
        if (!upload(&d.melw, w)) return nullptr;
    }
    return &(tables[K] = d);
}

void Ctx::time_begin(int kernel) {
    if (!timing) return;
    Timed t; t.kernel = kernel;
    (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, stream);
    pending.push_back(t);
}
void Ctx::time_end() {
    if (!timing || pending.empty()) return;
    (void)hipEventRecord(pending.back().b, stream);
}
void Ctx::time_collect() {
    for (auto &t : pending) {
        float ms = 0.f;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            sum_ms[t.kernel] += ms; count[t.kernel] += 1;
        }
        (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
    }
    pending.clear();
}

// The window side of the matrix-core DTW kernels truncates both parts of its split (rp_device.h pk_f16_second): x0 + x1 falls short of x
// by kDtwSplitShort of itself on average, so every cosine came out that much too small and a warping path added the shortfall up cell
// after cell -- the systematic part of those kernels' distance to the f32 kernels (round 3: rms 3e-7 of a score at score_ref 0.22).
// The template image is multiplied by 1 + kDtwSplitShort and both of its parts are rounded to nearest (a1 symmetric about zero makes the
// dropped x1 a1 term zero-mean too).  Calibrated on the GPU (tools/probe_score_ref.py, mean signed error of 11 520 scores against the
// oracle at score_ref 0.22: -7.7e-8 without the gain, +6.8e-9 at 1.2e-7, +6.0e-8 at 1.8e-7; rms 3.0e-7 in round 3, 1.5e-7 with the
// rounding alone, 1.2e-7 with the gain -- the f32 register kernels are at 0.9e-7).
#ifndef RP_DTW_SPLIT_SHORT
#define RP_DTW_SPLIT_SHORT 1.1e-7
#endif
static const double kDtwSplitShort = RP_DTW_SPLIT_SHORT;
// f32 -> f16 bits, round to nearest even
static uint16_t f16_rtn_bits(float v) {
    const _Float16 h = (_Float16)v;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static float f16_bits_to_f32(uint16_t hb) {
    const int e = (hb >> 10) & 0x1f;
    const uint32_t m = hb & 0x3ffu;
    const float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(0x400u | m), e - 25);
    return (hb & 0x8000u) ? -v : v;
}

// dtw_mfma_kernel's A operand of one chunk (rp_dtw_mfma.hip): per template row r [k half 2][template slot 8] x 8 f16.  With
// a = -(unit row) (1 + kDtwSplitShort) = a0 + a1 (a0 = rtn_f16(a), a1 = rtn_f16(a - a0)) and the half's components (ca, cb) = (0, 1) / (3, 4), the eight
// slots pair with the window side's (xa0, xb0 | xa1, xb1 | xa0, xb0 | x2_0, x2_1 or 1.0):
//   [ca.0, cb.0 | ca.0, cb.0 | ca.1, cb.1 | half 0: c2.0, c2.0 / half 1: c2.1, 1.0]
// i.e. x0 a0 + x1 a0 + x0 a1 for every component, and 1.0 x 1.0: the instruction accumulates 1 - a.x.  Slots past the chunk's
// count and the 16 rows after the last one are zero.
static void append_mfma_image(std::vector<uint16_t> &img, const DtwChunk &c, const float *unit, int Lpad) {
    const int K = 5;
    const size_t base = img.size();
    img.resize(base + (size_t)(c.len + 16) * kDtwMfmaRowBytes / 2, 0);
    for (int r = 0; r < c.len; ++r)
        for (int t = 0; t < c.count; ++t) {
            uint16_t p[5][2];
            for (int k = 0; k < K; ++k) {
                const float a = (float)(-(double)unit[((size_t)c.tid[t] * Lpad + r) * K + k] * (1.0 + kDtwSplitShort));
                p[k][0] = f16_rtn_bits(a);
                p[k][1] = f16_rtn_bits(a - f16_bits_to_f32(p[k][0]));
            }
            for (int kh = 0; kh < 2; ++kh) {
                const int ca = kh ? 3 : 0, cb = ca + 1;
                uint16_t sl[8] = {p[ca][0], p[cb][0], p[ca][0], p[cb][0], p[ca][1], p[cb][1], 0, 0};
                if (kh == 0) { sl[6] = p[2][0]; sl[7] = p[2][0]; }
                else { sl[6] = p[2][1]; sl[7] = 0x3c00; }
                std::memcpy(&img[base + ((size_t)r * kDtwMfmaRowBytes + kh * 128 + t * 16) / 2], sl, 16);
            }
        }
}

// f32 -> bf16 bits, round to nearest even (the values split here are finite)
static uint16_t bf16_rtn_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_bits_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}
// a = p0 + p1 + p2 EXACTLY, three bf16 values (8 significant bits each: an f32 has 24).  Rounded to nearest so that p1, p2 are symmetric about
// zero (the partial products the kernel drops -- x1 p2, x2 p1, x2 p2 -- are zero-mean); truncation, always exact, if a rounding carried a part
// out of reach (it cannot for normal values; subnormal leftovers are cut, far below anything the kernel resolves).
static void bf16_split3(float a, uint16_t p[3]) {
    p[0] = bf16_rtn_bits(a);
    const float r1 = a - bf16_bits_to_f32(p[0]);
    p[1] = bf16_rtn_bits(r1);
    const float r2 = r1 - bf16_bits_to_f32(p[1]);
    p[2] = bf16_rtn_bits(r2);
    if (bf16_bits_to_f32(p[0]) + bf16_bits_to_f32(p[1]) + bf16_bits_to_f32(p[2]) != a) {   // (each partial sum is exact in f32 when the split is)
        uint32_t u;
        std::memcpy(&u, &a, 4);
        p[0] = (uint16_t)(u >> 16);
        const float t1 = a - bf16_bits_to_f32(p[0]);
        std::memcpy(&u, &t1, 4);
        p[1] = (uint16_t)(u >> 16);
        const float t2 = t1 - bf16_bits_to_f32(p[1]);
        std::memcpy(&u, &t2, 4);
        p[2] = (uint16_t)(u >> 16);
    }
}

// dtw_mfma_kernel's A operand of one chunk in the f32-grade form (kArithF32Matrix; rp_dtw_mfma.hip, P3): per template row r
// [k-step 2][k half 2][template slot 8] x 8 bf16.  a = -(unit row) = a0 + a1 + a2 exactly; the window side splits its unit frame the same way,
// x = x0 + x1 + x2, and the two k-steps accumulate the six partial products x_i a_j with i + j <= 2 of every component, and 1.0 x 1.0:
//   k-step 0, half kh (components (ca, cb) = (0, 1) / (3, 4)):  [ca.0 cb.0 | half 0: c2.0 c2.0 / half 1: c2.0 c2.2 | ca.0 cb.0 | ca.0 cb.0]
//     against the window's registers 0..3                        [xa2 xb2  | half 0: x2_0 x2_1 / half 1: x2_2 x2_0 | xa0 xb0  | xa1 xb1 ]
//   k-step 1:                                                   [ca.1 cb.1 | ca.1 cb.1 | ca.2 cb.2 | half 0: c2.1 c2.1 / half 1: 1.0  0  ]
//     against the window's registers 2..5                        [xa0 xb0  | xa1 xb1  | xa0 xb0  | half 0: x2_0 x2_1 / half 1: 1.0  0  ]
// (x2_i: part i of component 2, which both lane halves split.)
// Slots past the chunk's count and the 16 rows after the last one are zero.
static void append_mfma_image3(std::vector<uint16_t> &img, const DtwChunk &c, const float *unit, int Lpad) {
    const int K = 5;
    const size_t base = img.size();
    img.resize(base + (size_t)(c.len + 16) * kDtwMfma3RowBytes / 2, 0);
    for (int r = 0; r < c.len; ++r)
        for (int t = 0; t < c.count; ++t) {
            uint16_t p[5][3];
            for (int k = 0; k < K; ++k) bf16_split3(-unit[((size_t)c.tid[t] * Lpad + r) * K + k], p[k]);
            for (int kh = 0; kh < 2; ++kh) {
                const int ca = kh ? 3 : 0, cb = ca + 1;
                // the kernel keeps the window side's two operands as ONE run of six registers, [x2a x2b | c2' | x0a x0b | x1a x1b | x0a x0b | c2''], the
                // first k-step reading registers 0..3 and the second 2..5 (the middle two are shared): the slots follow that order
                uint16_t s0[8] = {p[ca][0], p[cb][0], p[2][0], kh == 0 ? p[2][0] : p[2][2], p[ca][0], p[cb][0], p[ca][0], p[cb][0]};
                uint16_t s1[8] = {p[ca][1], p[cb][1], p[ca][1], p[cb][1], p[ca][2], p[cb][2], kh == 0 ? p[2][1] : (uint16_t)0x3f80, kh == 0 ? p[2][1] : (uint16_t)0};

                std::memcpy(&img[base + ((size_t)r * kDtwMfma3RowBytes + kh * 128 + t * 16) / 2], s0, 16);
                std::memcpy(&img[base + ((size_t)r * kDtwMfma3RowBytes + 256 + kh * 128 + t * 16) / 2], s1, 16);
            }
        }
}

// dtw_ragged_kernel's A operand of ONE template (rp_dtw_ragged.hip): per row [k half 2] x 8 f16, the same split as above without the
// constant slot (the cell adds its 1 itself): k half 0 = a0_0 a0_1 | a0_0 a0_1 | a1_0 a1_1 | a0_2 a0_2, k half 1 = a0_3 a0_4 | a0_3 a0_4 |
// a1_3 a1_4 | a1_2 0; 16 zero rows behind the last one.
static void append_ragged_image(std::vector<uint16_t> &img, int t, int len, const float *unit, int Lpad) {
    const int K = 5;
    const size_t base = img.size();
    img.resize(base + (size_t)(len + 16) * 16, 0);
    for (int r = 0; r < len; ++r) {
        uint16_t p[5][2];
        for (int k = 0; k < K; ++k) {
            const float a = (float)(-(double)unit[((size_t)t * Lpad + r) * K + k] * (1.0 + kDtwSplitShort));
            p[k][0] = f16_rtn_bits(a);
            p[k][1] = f16_rtn_bits(a - f16_bits_to_f32(p[k][0]));
        }
        for (int kh = 0; kh < 2; ++kh) {
            const int ca = kh ? 3 : 0, cb = ca + 1;
            uint16_t sl[8] = {p[ca][0], p[cb][0], p[ca][0], p[cb][0], p[ca][1], p[cb][1], 0, 0};
            if (kh == 0) { sl[6] = p[2][0]; sl[7] = p[2][0]; }
            else { sl[6] = p[2][1]; sl[7] = 0; }
            std::memcpy(&img[base + (size_t)r * 16 + kh * 8], sl, 16);
        }
    }
}

// dtw_mfma_wide_kernel's A operand of one chunk (rp_dtw_mfma_wide.hip): per template row [k-step][k half 2][template slot 8] x 8 f16.
// Lane half kh owns components kh * CHM .. kh * CHM + CHM - 1 (zero beyond K).  With a = -(unit row) = a0 + a1 the registers of a half
// are, per component pair (p, q): (a0p, a0q), (a0p, a0q), (a1p, a1q) against the window side's (x0p, x0q), (x1p, x1q), (x0p, x0q); an odd
// last component s: (a0s, a0s), (a1s, c) against (x0s, x1s), (x0s, c), c = 1.0 in half 1 only: the 1 of 1 - a.x.  An even count has no slot
// left for it: the kernel starts the sum at 1 instead (through round 4 a register (c, 0) of its own cost mfcc_size 16 a fourth k-step).
static void append_mfma_wide_image(std::vector<uint16_t> &img, const DtwChunk &c, const float *unit, int Lpad, int K) {
    const int CHM = dtw_mfma_wide_chm(K), NPAIR = CHM / 2, KS = dtw_mfma_wide_ksteps(K), row_bytes = dtw_mfma_wide_row_bytes(K);
    const size_t base = img.size();
    img.resize(base + (size_t)(c.len + 16) * row_bytes / 2, 0);
    for (int r = 0; r < c.len; ++r)
        for (int t = 0; t < c.count; ++t)
            for (int kh = 0; kh < 2; ++kh) {
                std::vector<uint16_t> v(8 * KS, 0);  // 4 KS registers of two f16
                auto part = [&](int j, int which) -> uint16_t {
                    const int comp = kh * CHM + j;
                    if (comp >= K) return 0;
                    const float a = (float)(-(double)unit[((size_t)c.tid[t] * Lpad + r) * K + comp] * (1.0 + kDtwSplitShort));
                    const uint16_t a0 = f16_rtn_bits(a);
                    return which == 0 ? a0 : f16_rtn_bits(a - f16_bits_to_f32(a0));
                };
                for (int j = 0; j < NPAIR; ++j) {
                    v[2 * (3 * j) + 0] = part(2 * j, 0); v[2 * (3 * j) + 1] = part(2 * j + 1, 0);
                    v[2 * (3 * j + 1) + 0] = part(2 * j, 0); v[2 * (3 * j + 1) + 1] = part(2 * j + 1, 0);
                    v[2 * (3 * j + 2) + 0] = part(2 * j, 1); v[2 * (3 * j + 2) + 1] = part(2 * j + 1, 1);
                }
                if (CHM % 2) {
                    v[2 * (3 * NPAIR) + 0] = part(CHM - 1, 0); v[2 * (3 * NPAIR) + 1] = part(CHM - 1, 0);
                    v[2 * (3 * NPAIR + 1) + 0] = part(CHM - 1, 1); v[2 * (3 * NPAIR + 1) + 1] = kh ? 0x3c00 : 0;
                }
                for (int ks = 0; ks < KS; ++ks)
                    std::memcpy(&img[base + ((size_t)r * row_bytes + ks * 256 + kh * 128 + t * 16) / 2], &v[8 * ks], 16);
            }
}

// dtw_mfma_wide3_kernel's A operand of one chunk of up to four templates (rp_dtw_mfma_wide3.hip): per template row [k-step 6][k half 2][template
// slot 4] x 8 bf16.  Lane half kh owns components kh * CHM .. kh * CHM + CHM - 1 (zero beyond K); a = -(unit row) = a0 + a1 + a2 exactly.  The
// 24 registers of a half pair the window side's run of twelve registers with the template parts that complete the six products x_i a_j
// (i + j <= 2) of every component: slot i of k-step ks holds what dtw_mfma_wide3_slot(K, ks, i) says (rp_kernels.h).
static void append_mfma_wide3_image(std::vector<uint16_t> &img, const DtwChunk &c, const float *unit, int Lpad, int K) {
    const int CHM = dtw_mfma_wide_chm(K), KS = kDtwWide3KSteps, row_bytes = kDtwWide3RowBytes;
    const size_t base = img.size();
    img.resize(base + (size_t)(c.len + 16) * row_bytes / 2, 0);
    for (int r = 0; r < c.len; ++r)
        for (int t = 0; t < c.count; ++t)
            for (int kh = 0; kh < 2; ++kh) {
                std::vector<uint16_t> v(8 * KS, 0);  // 4 KS registers of two bf16
                auto part = [&](int j, int which) -> uint16_t {   // part `which` of component j of this half; -1: zero; -2: the constant (half 1)
                    if (j == -2) return kh ? 0x3f80 : 0;
                    const int comp = kh * CHM + j;
                    if (j < 0 || j >= CHM || comp >= K) return 0;
                    uint16_t p[3];
                    bf16_split3(-unit[((size_t)c.tid[t] * Lpad + r) * K + comp], p);
                    return p[which];
                };
                for (int ks = 0; ks < KS; ++ks)
                    for (int i = 0; i < 4; ++i) {
                        const W3Slot sl = dtw_mfma_wide3_slot(K, ks, i);
                        v[2 * (4 * ks + i)] = part(sl.c0, sl.p0);
                        v[2 * (4 * ks + i) + 1] = part(sl.c1, sl.p1);
                    }
                for (int ks = 0; ks < KS; ++ks)
                    std::memcpy(&img[base + ((size_t)r * row_bytes + ks * 128 + kh * 64 + t * 16) / 2], &v[8 * ks], 16);
            }
}

// Template rows are scaled to unit L2 norm in f64 and rounded once to f32; an all-zero
// row stays zero so that its cosine similarity is 0 (src/mfcc/comparator.rs:43-47).
Templates *Templates::create(Ctx *ctx, int T, int K, const int *lens, const float *feats, int avg_len,
                             const float *avg) {
    if (T < 1 || K < 1) { set_last_error("Can not create an empty wakeword"); return nullptr; }  // wakeword_ref.rs:53
    int max_len = 0, longest = 0;
    for (int t = 0; t < T; ++t) {
        if (lens[t] < 1) { set_last_error("wakeword template without frames"); return nullptr; }
        if (lens[t] > max_len) max_len = lens[t];
    }
    longest = max_len;
    const int has_avg = (avg && avg_len > 0) ? 1 : 0;
    if (has_avg && avg_len > longest) longest = avg_len;
    {   // the DTW kernels keep a tile of windows plus one template length of frames in the CU's 160 KB of LDS; the band is at
        // least |m - n| wide (dtw.rs:64-67: an averaged template longer than the window) and 5 by default -- a larger
        // band_size is only known at scoring time, where launch_dtw refuses what does not fit
        int window_len = 0;
        for (int t = 0; t < T; ++t) window_len = std::max(window_len, lens[t]);
        const int diff = longest - window_len, wmin = std::max(5, diff);
        const size_t band_floats = (size_t)(2 * wmin + 1) * 64;
        const size_t need = ((size_t)(64 + longest - 1) * (size_t)(K | 1) + (size_t)K * 64 + band_floats) * sizeof(float);
        if (need > 160 * 1024) {
            const size_t fixed = (size_t)K * 64 + band_floats;
            const size_t lim = 160 * 1024 / sizeof(float) > fixed ? (160 * 1024 / sizeof(float) - fixed) / (size_t)(K | 1) : 0;
            set_last_error("wakeword template of " + std::to_string(longest) + " frames is too long for the device kernels (limit " +
                           std::to_string(lim > 63 ? lim - 63 : 0) + " frames at mfcc_size " + std::to_string(K) + ")");
            return nullptr;
        }
    }
    const int Ttot = T + has_avg, Lpad = longest;
    // (unit: 32 rows of slack behind the last template -- dtw_ragged_kernel requests the rows of a band a few columns past a template's end)
    std::vector<float> unit((size_t)(Ttot * Lpad + 32) * K, 0.f), raw((size_t)Ttot * Lpad * K, 0.f);
    std::vector<int> hl(Ttot);
    bool ref_only = false;
    size_t off = 0;
    for (int t = 0; t < Ttot; ++t) {
        const float *src = t < T ? feats + off : avg;
        const int L = t < T ? lens[t] : avg_len;
        hl[t] = L;
        for (int r = 0; r < L; ++r) {
            double nn = 0.0;
            for (int k = 0; k < K; ++k) nn += (double)src[(size_t)r * K + k] * (double)src[(size_t)r * K + k];
            // the reference tests sqrt(dot_a*dot_b) == 0 in f32; a row whose f32 squared norm is 0 is a zero row
            float nf = 0.f;
            for (int k = 0; k < K; ++k) nf += src[(size_t)r * K + k] * src[(size_t)r * K + k];
            double inv = (nf > 0.f && nn > 0.0) ? 1.0 / std::sqrt(nn) : 0.0;
            for (int k = 0; k < K; ++k) unit[((size_t)t * Lpad + r) * K + k] = (float)((double)src[(size_t)r * K + k] * inv);
            for (int k = 0; k < K; ++k) raw[((size_t)t * Lpad + r) * K + k] = src[(size_t)r * K + k];
            // outside this range the reference's sqrt(dot_a * dot_b) is not what a unit-length row gives (TemplatesDev::ref_only)
            if (!(nf == 0.f || (nf >= kDtwNormLo && nf <= kDtwNormHiRow))) ref_only = true;
        }
        if (t < T) off += (size_t)L * K;
    }
    std::unique_ptr<Templates> tp(new Templates());
    tp->ctx = ctx;
    TemplatesDev &d = tp->dev;
    d.arith = &ctx->arith;
    d.T = T; d.K = K; d.Lpad = Lpad; d.has_avg = has_avg; d.max_len = max_len; d.max_diff = longest - max_len;
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.lens), sizeof(int) * Ttot), "hipMalloc(lens)")) return nullptr;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.unit), sizeof(float) * unit.size()), "hipMalloc(templates)")) return nullptr;
    if (!hip_ok(hipMemcpy(d.lens, hl.data(), sizeof(int) * Ttot, hipMemcpyHostToDevice), "hipMemcpy(lens)")) return nullptr;
    if (!hip_ok(hipMemcpy(d.unit, unit.data(), sizeof(float) * unit.size(), hipMemcpyHostToDevice), "hipMemcpy(templates)")) return nullptr;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.raw), sizeof(float) * raw.size()), "hipMalloc(templates)")) return nullptr;
    if (!hip_ok(hipMemcpy(d.raw, raw.data(), sizeof(float) * raw.size(), hipMemcpyHostToDevice), "hipMemcpy(templates)")) return nullptr;
    d.ref_only = ref_only ? 1 : 0;

    // chunks for the register kernels: sample templates grouped by length, up to 8 per chunk, classed by chunk size
    // (TemplatesDev::class_first); the averaged template is its own chunk, last of the single-template class.
    std::vector<int> order(T);
    for (int t = 0; t < T; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hl[a] < hl[b]; });
    const int reg_tile = dtw_register_tile(K, 5);  // chunk size is fixed at upload time, before the band is known
    const int chunk_cap = reg_tile > 0 ? reg_tile : kChunkMax;
    std::vector<DtwChunk> byclass[4];
    for (int i = 0; i < T;) {
        int j = i;
        while (j < T && hl[order[j]] == hl[order[i]] && j - i < chunk_cap) ++j;
        DtwChunk c{};
        c.len = hl[order[i]]; c.count = j - i; c.tc = c.count <= 2 ? 2 : c.count <= 4 ? 4 : 8;
        for (int q = 0; q < kChunkMax; ++q) c.tid[q] = order[i + (q < c.count ? q : 0)];
        byclass[c.count == 1 ? 3 : c.tc == 2 ? 0 : c.tc == 4 ? 1 : 2].push_back(c);
        i = j;
    }
    if (has_avg) {
        DtwChunk c{};
        c.len = avg_len; c.count = 1; c.tc = 2;
        for (int q = 0; q < kChunkMax; ++q) c.tid[q] = T;
        byclass[3].push_back(c);
    }
    std::vector<DtwChunk> chunks;
    std::vector<float> dup;
    std::vector<uint16_t> aimg;  // dtw_mfma_kernel's A images (mfcc_size 5, chunks of 3..8 templates)
    std::vector<uint16_t> aimg3; // ... in the three-part bf16 form (the default arithmetic)
    // class 4 (not a launch class of its own): the tc-4 halves of the class-2 chunks, see TemplatesDev::split_first
    std::vector<DtwChunk> halves;
    bool can_split = !byclass[2].empty();
    for (const DtwChunk &c : byclass[2]) can_split = can_split && c.count >= 7;
    if (can_split)
        for (const DtwChunk &c : byclass[2])
            for (int h = 0; h < 2; ++h) {
                DtwChunk x{};
                x.len = c.len; x.count = h == 0 ? 4 : c.count - 4; x.tc = 4;
                for (int q = 0; q < kChunkMax; ++q) x.tid[q] = c.tid[4 * h + (q < x.count ? q : 0)];
                halves.push_back(x);
            }
    for (int cls = 0; cls < 5; ++cls) {
        if (cls < 4) { d.class_first[cls] = (int)chunks.size(); d.class_count[cls] = (int)byclass[cls].size(); }
        else { d.split_first = (int)chunks.size(); d.split_count = (int)halves.size(); }
        for (DtwChunk c : (cls < 4 ? byclass[cls] : halves)) {
            c.rows_off = (int)dup.size();
            c.aimg_off = 0;
            c.aimg3_off = 0;
            if (K == 5 && (cls == 1 || cls == 2)) {
                c.aimg_off = (int)(aimg.size() * sizeof(uint16_t) / 16);
                append_mfma_image(aimg, c, unit.data(), Lpad);
                c.aimg3_off = (int)(aimg3.size() * sizeof(uint16_t) / 16);
                append_mfma_image3(aimg3, c, unit.data(), Lpad);
                int &ml = cls == 2 ? d.mfma_min_len : d.mfma_min_len4;
                ml = ml == 0 ? c.len : std::min(ml, c.len);
            }
            for (int r = 0; r < c.len; ++r)
                for (int pr = 0; pr < c.tc / 2; ++pr)
                    for (int k = 0; k < K; ++k)
                        for (int h = 0; h < 2; ++h) {  // (t0, t1) interleaved: one scalar pair feeds a packed FMA
                            const int tt = 2 * pr + h;
                            dup.push_back(unit[((size_t)c.tid[tt < c.count ? tt : 0] * Lpad + r) * K + k]);
                        }
            chunks.push_back(c);
        }
    }
    if (K == 13 || K == 16) {  // dtw_mfma_wide_kernel: every length at least three times, else the wide register kernels keep the set
        std::vector<DtwChunk> wide8;
        bool all3 = T >= 3;
        for (int i = 0; i < T && all3;) {
            int j = i;
            while (j < T && hl[order[j]] == hl[order[i]]) ++j;
            if (j - i < 3) { all3 = false; break; }
            for (int b = i; b < j; b += kChunkMax) {
                DtwChunk c{};
                c.len = hl[order[i]]; c.count = std::min(kChunkMax, j - b); c.tc = 8;
                for (int q = 0; q < kChunkMax; ++q) c.tid[q] = order[b + (q < c.count ? q : 0)];
                wide8.push_back(c);
            }
            i = j;
        }
        if (all3) {
            d.wide8_first = (int)chunks.size(); d.wide8_count = (int)wide8.size();
            for (DtwChunk c : wide8) {
                c.rows_off = 0;
                c.aimg_off = (int)(aimg.size() * sizeof(uint16_t) / 16);
                append_mfma_wide_image(aimg, c, unit.data(), Lpad, K);
                d.mfma_min_len = d.mfma_min_len == 0 ? c.len : std::min(d.mfma_min_len, c.len);
                chunks.push_back(c);
            }
            // the same templates as chunks of up to FOUR for the three-part kernel (dtw_mfma_wide3_kernel): a length's templates spread evenly
            // over its chunks (five: 3 + 2, not 4 + 1)
            d.wide4_first = (int)chunks.size();
            for (int i = 0; i < T;) {
                int j = i;
                while (j < T && hl[order[j]] == hl[order[i]]) ++j;
                const int nch = (j - i + 3) / 4;
                for (int b = i, ci = 0; ci < nch; ++ci) {
                    const int cnt = (j - b + (nch - ci) - 1) / (nch - ci);
                    DtwChunk c{};
                    c.len = hl[order[i]]; c.count = cnt; c.tc = 4; c.rows_off = 0; c.aimg_off = 0;
                    for (int q = 0; q < kChunkMax; ++q) c.tid[q] = order[b + (q < cnt ? q : 0)];
                    c.aimg3_off = (int)(aimg3.size() * sizeof(uint16_t) / 16);
                    append_mfma_wide3_image(aimg3, c, unit.data(), Lpad, K);
                    d.wide4_min_len = d.wide4_min_len == 0 ? c.len : std::min(d.wide4_min_len, c.len);
                    chunks.push_back(c);
                    ++d.wide4_count;
                    b += cnt;
                }
                i = j;
            }
        }
    }
    if (K == 5) {   // dtw_ragged_kernel: what the equal-length matrix kernel leaves (lengths that occur once or twice), any lengths per chunk
        std::vector<int> rag;   // `order` is sorted by length: shortest first
        for (int cls : {0, 3})
            for (const DtwChunk &c : byclass[cls])
                for (int q = 0; q < c.count; ++q) if (c.tid[q] < T) rag.push_back(c.tid[q]);
        std::stable_sort(rag.begin(), rag.end(), [&](int a, int b) { return hl[a] < hl[b]; });
        if (!rag.empty()) {
            std::vector<uint16_t> rimg;
            std::vector<int> roff(T, 0);
            d.rag_first = (int)chunks.size();
            d.rag_min_len = hl[rag[0]];
            // chunks of up to 8, filled evenly (9 templates: 5 + 4, not 8 + 1)
            const int n_rc = ((int)rag.size() + kChunkMax - 1) / kChunkMax;
            for (int b = 0, ci = 0; ci < n_rc; ++ci) {
                const int cnt = ((int)rag.size() - b + (n_rc - ci) - 1) / (n_rc - ci);
                DtwChunk c{};
                c.count = cnt; c.tc = 8; c.len = hl[rag[b + cnt - 1]];
                int a_bytes = 0, rows = 0;
                for (int q = 0; q < kChunkMax; ++q) c.tid[q] = rag[b + (q < cnt ? q : 0)];
                for (int q = 0; q < cnt; ++q) {
                    const int t = rag[b + q];
                    roff[t] = (int)(rimg.size() * sizeof(uint16_t) / 16);
                    append_ragged_image(rimg, t, hl[t], unit.data(), Lpad);
                    a_bytes += (hl[t] + 16) * 32;
                    rows += hl[t] * K;
                }
                d.rag_a_cap = std::max(d.rag_a_cap, a_bytes);
                d.rag_rows_cap = std::max(d.rag_rows_cap, (rows + 3) & ~3);
                chunks.push_back(c);
                b += cnt;
            }
            d.rag_count = n_rc;
            if (!hip_ok(hipMalloc(&d.rimg, sizeof(uint16_t) * rimg.size()), "hipMalloc(rimg)")) return nullptr;
            if (!hip_ok(hipMemcpy(d.rimg, rimg.data(), sizeof(uint16_t) * rimg.size(), hipMemcpyHostToDevice), "hipMemcpy(rimg)")) return nullptr;
            if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.rag_off), sizeof(int) * T), "hipMalloc(rag_off)")) return nullptr;
            if (!hip_ok(hipMemcpy(d.rag_off, roff.data(), sizeof(int) * T, hipMemcpyHostToDevice), "hipMemcpy(rag_off)")) return nullptr;
        }
    }
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.chunks), sizeof(DtwChunk) * chunks.size()), "hipMalloc(chunks)")) return nullptr;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.dup), sizeof(float) * dup.size()), "hipMalloc(dup)")) return nullptr;
    if (!hip_ok(hipMemcpy(d.chunks, chunks.data(), sizeof(DtwChunk) * chunks.size(), hipMemcpyHostToDevice), "hipMemcpy(chunks)")) return nullptr;
    if (!hip_ok(hipMemcpy(d.dup, dup.data(), sizeof(float) * dup.size(), hipMemcpyHostToDevice), "hipMemcpy(dup)")) return nullptr;
    if (!aimg.empty()) {
        if (!hip_ok(hipMalloc(&d.aimg, sizeof(uint16_t) * aimg.size()), "hipMalloc(aimg)")) return nullptr;
        if (!hip_ok(hipMemcpy(d.aimg, aimg.data(), sizeof(uint16_t) * aimg.size(), hipMemcpyHostToDevice), "hipMemcpy(aimg)")) return nullptr;
    }
    if (!aimg3.empty()) {
        if (!hip_ok(hipMalloc(&d.aimg3, sizeof(uint16_t) * aimg3.size()), "hipMalloc(aimg3)")) return nullptr;
        if (!hip_ok(hipMemcpy(d.aimg3, aimg3.data(), sizeof(uint16_t) * aimg3.size(), hipMemcpyHostToDevice), "hipMemcpy(aimg3)")) return nullptr;
    }
    if (K == 5 && d.class_count[2] >= 4 && d.aimg) {   // dtw_mfma_group_kernel: class-2 chunks of one length, four to a workgroup
        std::vector<int> quads;   // (the kernel's two-chunk shape measured SLOWER than dtw_mfma_kernel -- half the sharing does not pay the ring -- and is not built)
        std::vector<char> grouped(d.class_count[2], 0);
        for (int i = 0; i < d.class_count[2];) {
            int j = i;
            while (j < d.class_count[2] && chunks[d.class_first[2] + j].len == chunks[d.class_first[2] + i].len) ++j;
            const int L = chunks[d.class_first[2] + i].len;
            int b = i;
            if (dtw_mfma_group_lds_bytes(L, 4) <= 160 * 1024)
                for (; b + 4 <= j; b += 4) { quads.push_back(d.class_first[2] + b); d.grp4_max_len = std::max(d.grp4_max_len, L); std::fill(grouped.begin() + b, grouped.begin() + b + 4, 1); }
            i = j;
        }
        bool ok = !quads.empty();
        for (int i = 0; i < d.class_count[2] && ok;) {   // what is left, as runs
            if (grouped[i]) { ++i; continue; }
            int j = i;
            while (j < d.class_count[2] && !grouped[j]) ++j;
            if (d.rest_runs == 8) { ok = false; break; }
            d.rest_first[d.rest_runs] = d.class_first[2] + i; d.rest_count[d.rest_runs] = j - i; ++d.rest_runs;
            i = j;
        }
        if (ok) {
            if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.grp_first), sizeof(int) * quads.size()), "hipMalloc(grp_first)")) return nullptr;
            if (!hip_ok(hipMemcpy(d.grp_first, quads.data(), sizeof(int) * quads.size(), hipMemcpyHostToDevice), "hipMemcpy(grp_first)")) return nullptr;
            d.grp_count = (int)quads.size();
        } else {
            d.rest_runs = 0;
        }
    }
    // chunks that index the call's tile counters (DtwWork::sched): every chunk but the ragged ones at the end of the array -- dtw_ragged_kernel
    // takes its tiles by index and uses no counter (round-5 advice: its chunks must not count against the limit of a kernel that never runs
    // in the default arithmetic)
    d.n_chunks_total = d.rag_count > 0 ? d.rag_first : (int)chunks.size();
    if (d.n_chunks_total > kDtwSchedChunks) { set_last_error("wakeword reference with too many template lengths for the device kernels"); return nullptr; }
    return tp.release();
}

Templates::~Templates() {
    if (dev.lens) (void)hipFree(dev.lens);
    if (dev.unit) (void)hipFree(dev.unit);
    if (dev.chunks) (void)hipFree(dev.chunks);
    if (dev.dup) (void)hipFree(dev.dup);
    if (dev.aimg) (void)hipFree(dev.aimg);
    if (dev.aimg3) (void)hipFree(dev.aimg3);
    if (dev.raw) (void)hipFree(dev.raw);
    if (dev.rimg) (void)hipFree(dev.rimg);
    if (dev.grp_first) (void)hipFree(dev.grp_first);
    if (dev.rag_off) (void)hipFree(dev.rag_off);
}

// f32 -> bf16, round to nearest even (matches the kernel's in-register conversion)
static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

Model *Model::create(Ctx *ctx, int n_layers, const int *dims, const float *const *weights, const float *const *biases) {
    if (n_layers < 1 || n_layers > 3) { set_last_error("Incorrect model layers"); return nullptr; }
    for (int l = 0; l <= n_layers; ++l) if (dims[l] < 1) { set_last_error("Incorrect model layers"); return nullptr; }
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    std::unique_ptr<Model> m(new Model());
    m->ctx = ctx;
    m->dims.assign(dims, dims + n_layers + 1);
    auto up = [&](const void *src, size_t bytes, void **dst) {
        return hip_ok(hipMalloc(dst, bytes), "hipMalloc(model)") && hip_ok(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(model)");
    };
    for (int l = 0; l < n_layers; ++l) {
        float *w = nullptr, *b = nullptr;
        if (!up(weights[l], sizeof(float) * (size_t)dims[l] * dims[l + 1], reinterpret_cast<void **>(&w))) return nullptr;
        m->W.push_back(w);
        if (!up(biases[l], sizeof(float) * (size_t)dims[l + 1], reinterpret_cast<void **>(&b))) return nullptr;
        m->B.push_back(b);
    }
    m->w1_host.assign(weights[0], weights[0] + (size_t)dims[0] * dims[1]);
    MlpDev &d = m->dev;
    d.n_layers = n_layers;
    for (int l = 0; l <= n_layers; ++l) d.dims[l] = dims[l];
    const int n1 = dims[1];
    d.nt = n1 <= 16 ? 1 : n1 <= 32 ? 2 : n1 <= 80 ? 5 : n1 <= 144 ? 9 : 0;
    bool tail_ok = true;
    for (int l = 2; l < n_layers; ++l) tail_ok = tail_ok && dims[l] <= 255;
    m->mfma_ok = d.nt > 0 && dims[0] % 4 == 0 && tail_ok;
    if (m->mfma_ok) {
        d.kpad = (dims[0] + 127) / 128 * 128;  // whole unrolled k-groups of both MFMA kernels
        const int rows = 16 * d.nt;
        std::vector<float> wf((size_t)rows * d.kpad, 0.f), b1(rows, 0.f);
        std::vector<uint16_t> wh((size_t)rows * d.kpad, 0), wsp((size_t)2 * rows * d.kpad, 0), wtp((size_t)3 * rows * d.kpad, 0);
        for (int o = 0; o < n1; ++o) {
            b1[o] = biases[0][o];
            for (int i = 0; i < dims[0]; ++i) {
                float v = weights[0][(size_t)o * dims[0] + i];
                wf[(size_t)o * d.kpad + i] = v;
                wh[(size_t)o * d.kpad + i] = f32_to_bf16(v);
                const _Float16 w0 = (_Float16)v, w1 = (_Float16)(v - (float)w0);   // kMlpF16x2: w = w0 + w1 to 22 bits
                __builtin_memcpy(&wsp[(size_t)o * d.kpad + i], &w0, 2);
                __builtin_memcpy(&wsp[(size_t)rows * d.kpad + (size_t)o * d.kpad + i], &w1, 2);
                uint16_t p3[3];   // kMlpBf16x3: w = p0 + p1 + p2 exactly
                bf16_split3(v, p3);
                for (int part = 0; part < 3; ++part) wtp[((size_t)part * rows + o) * d.kpad + i] = p3[part];
            }
        }
        std::vector<float> tail;
        for (int l = 1; l < n_layers; ++l) {
            tail.insert(tail.end(), weights[l], weights[l] + (size_t)dims[l] * dims[l + 1]);
            tail.insert(tail.end(), biases[l], biases[l] + dims[l + 1]);
        }
        if (tail.empty()) tail.push_back(0.f);
        d.tail_floats = (int)tail.size();
        if (!mlp_mfma_fits(d)) { m->mfma_ok = false; d.nt = 0; return m.release(); }   // e.g. a 255-wide hidden layer: per-layer kernel
        if (dims[0] % 16 == 0 && n1 <= 160) {   // mlp_windows_kernel / mlp_windows_wide_kernel (mfcc_size 16): [frame][32-output tile q][part][k-half][output][8 k]
            const int Lw = dims[0] / 16, NQ = (n1 + 31) / 32;
            std::vector<uint16_t> img((size_t)Lw * NQ * 2 * 2 * 32 * 8, 0);
            for (int f = 0; f < Lw; ++f)
                for (int part = 0; part < 2; ++part)
                    for (int h = 0; h < 2; ++h)
                        for (int o = 0; o < n1; ++o)
                            for (int e = 0; e < 8; ++e)
                                img[((((((size_t)f * NQ + o / 32) * 2 + part) * 2 + h) * 32) + o % 32) * 8 + e] =
                                    wsp[(size_t)part * rows * d.kpad + (size_t)o * d.kpad + 16 * f + 8 * h + e];
            if (!up(img.data(), img.size() * 2, &d.wwin)) return nullptr;
            std::vector<uint16_t> img3((size_t)Lw * NQ * 3 * 2 * 32 * 8, 0);   // the three-part bf16 image of the same kernels (RP_MLP_F32)
            for (int f = 0; f < Lw; ++f)
                for (int part = 0; part < 3; ++part)
                    for (int h = 0; h < 2; ++h)
                        for (int o = 0; o < n1; ++o)
                            for (int e = 0; e < 8; ++e)
                                img3[((((((size_t)f * NQ + o / 32) * 3 + part) * 2 + h) * 32) + o % 32) * 8 + e] =
                                    wtp[(size_t)part * rows * d.kpad + (size_t)o * d.kpad + 16 * f + 8 * h + e];
            if (!up(img3.data(), img3.size() * 2, &d.wwin3)) return nullptr;
        }
        if (!up(wf.data(), wf.size() * 4, reinterpret_cast<void **>(&d.w1f)) || !up(wh.data(), wh.size() * 2, &d.w1h) || !up(wsp.data(), wsp.size() * 2, &d.w1s) || !up(wtp.data(), wtp.size() * 2, &d.w1t) ||
            !up(b1.data(), b1.size() * 4, reinterpret_cast<void **>(&d.b1)) || !up(tail.data(), tail.size() * 4, reinterpret_cast<void **>(&d.tail)))
            return nullptr;
    }
    return m.release();
}

const float *Model::wsum_for(int K) {
    auto it = wsums.find(K);
    if (it != wsums.end()) return it->second->as<float>();
    if (K < 1 || dims[0] % K != 0 || dev.nt <= 0) return nullptr;
    const int L = dims[0] / K, n1 = dims[1], rows = 16 * dev.nt;
    std::vector<float> ws((size_t)rows * K, 0.f);
    for (int o = 0; o < n1; ++o)
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
            for (int i = 0; i < L; ++i) acc += (double)w1_host[(size_t)o * dims[0] + (size_t)i * K + k];
            ws[(size_t)o * K + k] = (float)acc;
        }
    std::unique_ptr<DevBuf> b(new DevBuf());
    if (!b->reserve(ws.size() * sizeof(float))) return nullptr;
    if (!hip_ok(hipMemcpy(b->p, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(wsum)")) return nullptr;
    const float *p = b->as<float>();
    wsums[K] = std::move(b);
    return p;
}

// Plan of mlp_stream_kernel for rows starting at x.  The layer-1 weights are laid out in the order its MFMA lanes read
// them: k-step m covers k in [32m, 32m + 32); lane (li = l & 15, lk = l >> 4) holds, for output 16 n + li, the eight k
// its A fragment carries -- k = 32m + 4lk + e and 32m + 16 + 4lk + e, e < 4 -- as one 16-byte piece per (k-step, n) in
// bf16, or two (one per half) in f32.  Zero past dims[0] and past dims[1].  The image does not depend on where the rows
// start (the kernel shifts its reads instead), so it is built once per precision.
bool Model::stream_plan(const float *x, size_t B, int precision, MlpStreamPlan *plan) {
    if (!mfma_ok || (precision != kMlpF32 && precision != kMlpBf16 && precision != kMlpF16x2 && precision != kMlpBf16x3) || !mlp_stream_supported(dev, x, precision)) return false;
    const int in = dims[0], n1 = dims[1], nt = dev.nt;
    const int q0 = (int)((reinterpret_cast<uintptr_t>(x) >> 4) & 7);
    const int par = ((in / 4) % 8) != 0;  // in % 16 == 0: the row pitch is 0 or 4 chunks mod 8
    const int q1 = par ? (q0 + 4) & 7 : q0;
    const int lines = (in - 1 + 4 * std::max(q0, q1)) / 32 + 1;   // lines that hold bytes of a row
    plan->units = (lines + 1) / 2;
    plan->par = par;
    plan->q[0] = q0; plan->q[1] = q1;
    plan->nbt = par ? 2 * (int)((B + 255) / 256) : (int)((B + 127) / 128);
    const int lines_max = (in - 1 + 4 * 7) / 32 + 1;             // at the largest phase
    const int ksteps = 2 * ((lines_max + 1) / 2);                // two per unit, for every phase
    std::unique_ptr<DevBuf> &buf = stream_img[precision];
    if (!buf) {
        const bool f32 = precision == kMlpF32, f16x2 = precision == kMlpF16x2, b3 = precision == kMlpBf16x3;
        const size_t wk = (size_t)(precision == kMlpBf16 ? 1024 : b3 ? 3072 : 2048) * nt;
        std::vector<uint8_t> img(wk * ksteps, 0);
        auto wat = [&](int o, long k) -> float { return (o < n1 && k < in) ? w1_host[(size_t)o * in + k] : 0.f; };
        for (int m = 0; m < ksteps; ++m)
            for (int n = 0; n < nt; ++n)
                for (int l = 0; l < 64; ++l) {
                    const int li = l & 15, lk = l >> 4, o = 16 * n + li;
                    const long k0 = 32L * m + 4L * lk;
                    if (f32) {
                        for (int h = 0; h < 2; ++h) {
                            float *dst = reinterpret_cast<float *>(img.data() + wk * m + ((size_t)h * nt + n) * 1024 + l * 16);
                            for (int e = 0; e < 4; ++e) dst[e] = wat(o, k0 + 16 * h + e);
                        }
                    } else if (b3) {
                        // pieces [part][n], part = 0, 1, 2: the exact three-part bf16 split of the weight; the eight k of a piece as in the bf16 image
                        for (int e = 0; e < 8; ++e) {
                            uint16_t p3[3];
                            bf16_split3(wat(o, k0 + (e < 4 ? e : 12 + e)), p3);
                            for (int part = 0; part < 3; ++part)
                                reinterpret_cast<uint16_t *>(img.data() + wk * m + ((size_t)part * nt + n) * 1024 + l * 16)[e] = p3[part];
                        }
                    } else if (f16x2) {
                        // pieces [part][n]: part 0 = w0 = f16(w), part 1 = w1 = f16(w - w0); the eight k of a piece as in the bf16 image
                        uint16_t *d0 = reinterpret_cast<uint16_t *>(img.data() + wk * m + (size_t)n * 1024 + l * 16);
                        uint16_t *d1 = reinterpret_cast<uint16_t *>(img.data() + wk * m + ((size_t)nt + n) * 1024 + l * 16);
                        for (int e = 0; e < 8; ++e) {
                            const float w = wat(o, k0 + (e < 4 ? e : 12 + e));
                            const _Float16 w0 = (_Float16)w;
                            const _Float16 w1 = (_Float16)(w - (float)w0);
                            __builtin_memcpy(d0 + e, &w0, 2);
                            __builtin_memcpy(d1 + e, &w1, 2);
                        }
                    } else {
                        uint16_t *dst = reinterpret_cast<uint16_t *>(img.data() + wk * m + (size_t)n * 1024 + l * 16);
                        for (int e = 0; e < 8; ++e) dst[e] = f32_to_bf16(wat(o, k0 + (e < 4 ? e : 12 + e)));
                    }
                }
        std::unique_ptr<DevBuf> b(new DevBuf());
        if (!b->reserve(img.size())) return false;
        if (!hip_ok(hipMemcpy(b->p, img.data(), img.size(), hipMemcpyHostToDevice), "hipMemcpy(stream weights)")) return false;
        buf = std::move(b);
        stream_ksteps = ksteps;
    }
    if (2 * plan->units > stream_ksteps) return false;
    plan->wimg = buf->p;
    return true;
}

Model::~Model() {
    for (float *p : W) (void)hipFree(p);
    for (float *p : B) (void)hipFree(p);
    if (dev.w1f) (void)hipFree(dev.w1f);
    if (dev.w1h) (void)hipFree(dev.w1h);
    if (dev.w1s) (void)hipFree(dev.w1s);
    if (dev.w1t) (void)hipFree(dev.w1t);
    if (dev.wwin3) (void)hipFree(dev.wwin3);
    if (dev.wwin) (void)hipFree(dev.wwin);
    if (dev.b1) (void)hipFree(dev.b1);
    if (dev.tail) (void)hipFree(dev.tail);
}

}  // namespace rp
