// rp_dtw_ragged.hip -- dtw_ragged_kernel: the banded DTW of mfcc_size 5 on the matrix cores for templates of UNEQUAL length -- the shape
// of every wakeword file the reference ships (tests/wakeword.rs:27-71: 90..108, 99..126, 144..168 frames) and of BASELINE config C1
// (DESIGN.md §4.2b, round 5).  Built, parity-tested and measured; NOT the default path (dtw_ragged_supported below says why).  Same scoring as every other DTW kernel (src/mfcc/dtw.rs:56-105 + comparator.rs:15-48 +
// normalizer.rs:17-29 + wakeword_comp.rs:22-37).
//
// Why dtw_mfma_kernel cannot take such a set: the reference cuts the window to EACH template's length before it takes the column
// means (wakeword_comp.rs:22-27,99-104), so the mean-normalised, unit-length frame -- that kernel's B operand, prepared once per
// (window, column) and shared by the eight templates of a chunk -- is a different vector for every template.  Preparing it per
// (window, template, column) costs more vector instructions (centre, norm, scale, two-way f16 split: ~40) than the products it moves
// to the matrix pipe.  This kernel moves the products WITHOUT a per-window operand:
//
//   cost(r, c) = 1 - a_r . (x_f - mu) / |x_f - mu|            (a_r: unit template row, x_f: frame f = w + c - 1, mu: the window's mean)
//              = 1 + kappa(c) * ( a_r . (mu - o) s  -  a_r . (x_f - o) s ),      kappa(c) = 1 / (s |x_f - mu|)
//
//   * B operand = the two-way f16 split of (x_f - o) s -- o an offset and s a power of two per STREAM (ragged_prep_kernel: the mean of the
//     stream's first frames, and the scale that puts its largest |x - o| component below 2^14; functions of the stream alone, so a stream's
//     bits do not depend on the batch it is scored in) -- which does not depend on the window: every frame of a 512-window tile is split
//     ONCE while it is staged in LDS, and a band column costs each lane one 16-byte LDS read instead of ~27 vector instructions.
//   * C operand = G(r) = a_r . (mu - o) s, one f32 value per template row in the band: a ring of 16 values in the accumulator's own
//     layout (5 multiply-adds when a row enters the band); the instruction leaves G - a_r . (x_f - o) s = -s a_r . (x_f - mu), exact mean
//     handling in the f32 accumulator.
//   * a band cell is v_fma(acc, kappa, 1) + v_min3 + v_add; kappa(c) per (window, template, column) from the f32 frame itself (5
//     subtractions, 5 multiply-adds, v_rsq), as the reference rounds it.
//   * A wave owns 64 windows (lane = window) and walks the chunk's templates one after the other, shortest first: the column sums of
//     MfccNormalizer::normalize are sequential, so the sum over the first L_t frames is a prefix of the sum over the first L_t+1 frames --
//     one running sum serves every template, bit for bit the reference's.  Templates of any lengths share a wave without idle slots.
//   * v_mfma_f32_32x32x16_f16 hands lane (n, h) the rows (reg & 3) + 8 (reg >> 2) + 4 h of column n: 16 rows per lane half.  A tile = the
//     16 circular row slots of ONE template (row r in slot r mod 16) twice, in the rows of half 0 and in the rows of half 1; instruction 1
//     (A with the half-1 rows zero, B = frames of windows 0..31) and instruction 2 (half-0 rows zero, B = frames of windows 32..63,
//     accumulating) leave every lane the 16 slots of ITS window.  Half of the matrix work is on zeros: the pipe has the room (0.3 busy).
//   * Precision: 22-bit operands as in dtw_mfma_kernel, but relative to |x_f - o| instead of |x_f - mu|: a cell's error is 2^-22 x
//     ratio, ratio = |x_f - o| / |x_f - mu| (1..3 for speech and noise).  A window that meets a frame with ratio > 32, a frame too small
//     for the f16 parts after scaling (digital silence behind speech: its windows centre to rounding residue), or one outside the
//     reference's own norm range (rp_kernels.h kDtwNormLo) is LISTED and scored again by the scale-invariant register kernels in their
//     list mode (launch_dtw_k5; a whole batch of such windows costs one register pass, there is no cliff) -- or, for callers whose frame
//     array has no slack behind its end, by dtw_ref_kernel like every fast kernel's out-of-range pairs.
#include "rp_device.h"

#include <cstdlib>

namespace rp {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRK = 5;                          // MFCC coefficients per frame
constexpr int kRSlots = 16;                     // circular row slots of a template (band + neighbours: 2 W + 2 <= 16)
constexpr int kRTile = 64 * kDtwRaggedWaves;    // windows per workgroup tile
constexpr float kRRatio = 32.f;                 // |x_f - o| / |x_f - mu| above which a (window, template) is rescored by dtw_ref_kernel
constexpr float kRKappaMax = 7.5f;              // 1 / (s |x_f - mu|) above which the second f16 parts are subnormal (abs error 2^-24 each)

__device__ __forceinline__ unsigned pkrtz(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned x = (unsigned)__shfl_xor((int)v, o, 64); v = x > v ? x : v; }
    return v;
}

}  // namespace

// Per stream: the offset o (mean of its first <= 64 scored frames) and the power-of-two scale s of dtw_ragged_kernel's B images, from the
// stream's own frames [first_win, first_win + n_frames): prep[s] = {o0..o4, s, bad, -}.  bad: a component of |x - o| is >= 2^15, inf or NaN
// (|frame|^2 may leave the reference's range, kDtwFixLimit): every window of the stream is listed.  One wave per stream.
__global__ __launch_bounds__(256) void ragged_prep_kernel(const float *__restrict__ mfcc, size_t frame_pitch, size_t n_streams, size_t first_win,
                                                          size_t n_frames, float *__restrict__ prep) {
    constexpr int K = kRK;
    const int lane = threadIdx.x & 63;
    const size_t st = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (st >= n_streams) return;
    const float *x = mfcc + (st * frame_pitch + first_win) * K;
    const size_t n0 = n_frames < 64 ? n_frames : 64;
    float o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) o[j] = wave_sum((size_t)lane < n0 ? x[(size_t)lane * K + j] : 0.f) / (float)n0;
    unsigned mxu = 0;
    for (size_t i = lane; i < n_frames; i += 64)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const unsigned a = __float_as_uint(fabsf(x[i * K + j] - o[j]));   // as bits: NaN > inf > finite
            mxu = a > mxu ? a : mxu;
        }
    mxu = wave_max_u(mxu);
    // |x - o| < 2^(eb - 126) with eb the biased exponent of the largest; s = 2^(140 - eb) puts it below 2^14 (f16: both split parts stay
    // normal down to 2^-16 of it); s <= 2^60
    const int eb = (int)(mxu >> 23);
    const bool bad = eb > 127 + 14;
    int sb = 267 - eb;
    sb = sb > 187 ? 187 : sb;
    const float sc = mxu == 0 || bad ? 1.f : __uint_as_float((unsigned)sb << 23);
    if (lane == 0) {
        float *p = prep + st * 8;
#pragma unroll
        for (int j = 0; j < K; ++j) p[j] = o[j];
        p[5] = sc; p[6] = bad ? 1.f : 0.f; p[7] = 0.f;
    }
}

// One workgroup = kDtwRaggedWaves waves on one ragged chunk (its A images are staged once) walking tiles of 512 consecutive entries of the
// flattened (stream, window) space; the tile's frames -- every stream segment it touches plus max_len + 2 frames -- are staged and split
// once for all eight waves.  rows_list != null: imprecise windows are appended there ([0] = count, rows from [1]) for the register kernels'
// list mode; else to `fix` (dtw_ref_kernel).  abandon_nc: as dtw_mfma_kernel.
template <int W>
__global__ __launch_bounds__(64 * kDtwRaggedWaves, 4) void dtw_ragged_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_streams, size_t first_win, size_t n_win, size_t out_win_pitch,
    const DtwChunk *__restrict__ chunks, int chunk_base, unsigned n_chunks, const int *__restrict__ lens, const int *__restrict__ rag_off,
    const uint4 *__restrict__ rimg, const float *__restrict__ unit, int Lpad, int T, float score_ref, float *__restrict__ scores, int max_len,
    int frames_cap, int a_cap_bytes, float abandon_nc, const float *__restrict__ prep, uint32_t *__restrict__ rows_list, uint32_t *__restrict__ fix) {
    constexpr int K = kRK, B = 2 * W, NS = kRSlots;
    static_assert(B + 2 <= NS, "the band and its two neighbours must fit the circular row slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned ci = blockIdx.x % n_chunks, n_groups = gridDim.x / n_chunks;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int count = ch->count;
    // ---- LDS: A images | frames: components 0..3 [F] x 16 B, component 4 [F] | B images: k half 0 [F] x 16 B, k half 1 [F] x 16 B (lane n of
    // a half reads piece f + n: sixteen lanes cover 256 consecutive bytes, the conflict-free shape of ds_read_b128) | tables
    const int F = frames_cap;
    unsigned char *a_lds = smem;
    f32x4 *xa = reinterpret_cast<f32x4 *>(smem + a_cap_bytes);
    float *x4 = reinterpret_cast<float *>(xa + F);
    u32x4 *bimg = reinterpret_cast<u32x4 *>(x4 + F);
    int *a_off = reinterpret_cast<int *>(bimg + 2 * (size_t)F), *t_len = a_off + kChunkMax, *t_id = t_len + kChunkMax;
    if (tid == 0) {
        int off = 0;
        for (int i = 0; i < count; ++i) {
            const int t = ch->tid[i];
            t_id[i] = t; t_len[i] = lens[t]; a_off[i] = off;
            off += (lens[t] + NS) * 32;
        }
    }
    __syncthreads();
    for (int i = 0; i < count; ++i) {   // the chunk's A images (32 bytes per row, 16 zero rows behind each)
        const int t = t_id[i], L = t_len[i];
        const u32x4 *src = reinterpret_cast<const u32x4 *>(rimg) + rag_off[t];
        u32x4 *dst = reinterpret_cast<u32x4 *>(a_lds + a_off[i]);
        for (int j = tid; j < (L + NS) * 2; j += 64 * kDtwRaggedWaves) dst[j] = src[j];
    }
    // A operand: this lane supplies row m = lane & 31 of the tile, k half = lane >> 5; row m = slot (m & 3) + 4 (m >> 3) of lane half
    // (m >> 2) & 1 (the C/D layout read backwards)
    const int mrow = lane & 31, kh = lane >> 5;
    const int jA = (mrow & 3) + 4 * (mrow >> 3), hsA = (mrow >> 2) & 1;
    const int jA0 = hsA == 0 ? jA : 99, jA1 = hsA == 1 ? jA : 99;   // the slot this lane feeds in instruction 1 / 2 (99: zeros)
    const size_t total_entries = n_streams * n_win;
    const size_t n_tiles = (total_entries + kRTile - 1) / kRTile;
    const unsigned P = (unsigned)n_win + (unsigned)max_len + 2u;   // staged frames of a whole stream segment

    for (size_t tile = blockIdx.x / n_chunks; tile < n_tiles; tile += n_groups) {
        __syncthreads();   // the previous tile's frames are done with (and the images are in place)
        const size_t e0 = tile * kRTile;
        const unsigned remaining = total_entries - e0 < (size_t)kRTile ? (unsigned)(total_entries - e0) : (unsigned)kRTile;
        const size_t s0 = e0 / n_win;
        const unsigned w0 = (unsigned)(e0 - s0 * n_win);
        const unsigned nseg = (remaining + w0 - 1) / (unsigned)n_win + 1;
        const unsigned Ftot = remaining + nseg * ((unsigned)max_len + 2u);
        // ---- stage: the frames of every segment and their B images: (x - o) s = x0 + x1, x0 = rtz_f16, x1 = rtz_f16(rest) (the template
        // side carries the gain that undoes the truncation, rp_ctx.cpp kDtwSplitShort); k half 0: x0_0 x0_1 | x1_0 x1_1 | x0_0 x0_1 | x0_2 x1_2,
        // k half 1: x0_3 x0_4 | x1_3 x1_4 | x0_3 x0_4 | x0_2 0 -- against a0 a0 | a0 a0 | a1 a1 | a0_2 a0_2 and a0 a0 | a0 a0 | a1 a1 | a1_2 0 of the row image
        for (unsigned i = tid; i < Ftot; i += 64 * kDtwRaggedWaves) {
            const unsigned k = (i + w0) / P;
            const size_t fr = first_win + (size_t)(i + w0 - k * P), st = s0 + k;
            const bool ok = st < n_streams && fr < frame_pitch;
            float xv[K] = {0.f, 0.f, 0.f, 0.f, 0.f};
            float y[K], r[K];
            if (ok) {
                const float *src = mfcc + (st * frame_pitch + fr) * K;
#pragma unroll
                for (int j = 0; j < K; ++j) xv[j] = src[j];
            }
            const float *pp = prep + (st < n_streams ? st : n_streams - 1) * 8;
            const float sc = pp[5];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                y[j] = ok ? (xv[j] - pp[j]) * sc : 0.f;
                r[j] = y[j] - __uint_as_float(__float_as_uint(y[j]) & 0xffffe000u);
            }
            xa[i] = (f32x4){xv[0], xv[1], xv[2], xv[3]};
            x4[i] = xv[4];
            u32x4 lo, hi;
            lo.x = pkrtz(y[0], y[1]); lo.y = pk_f16_second(r[0], r[1]); lo.z = lo.x;
            hi.x = pkrtz(y[3], y[4]); hi.y = pk_f16_second(r[3], r[4]); hi.z = hi.x;
            const unsigned x2 = pkrtz(y[2], 0.f) & 0xffffu, r2 = pk_f16_second(r[2], 0.f) & 0xffffu;
            lo.w = x2 | (r2 << 16);
            hi.w = x2;
            bimg[i] = lo;
            bimg[(size_t)F + i] = hi;
        }
        __syncthreads();

        // ---- lanes -> (stream, window)
        const unsigned el = (unsigned)tid < remaining ? (unsigned)tid : remaining - 1;
        const bool valid = (unsigned)tid < remaining;
        const unsigned kseg = (el + w0) / (unsigned)n_win;
        const unsigned w = el + w0 - kseg * (unsigned)n_win;
        const size_t strm = s0 + kseg;
        const float *pw = prep + strm * 8;
        const float s = pw[5];
        const bool bad_tile = pw[6] != 0.f;   // (the stream's: every window of it is listed)
        const unsigned fb = kseg * P + w - w0;   // the window's first frame in the planes
        // instruction 1 takes the frames of the windows of lanes 0..31 (k half from this lane's half), instruction 2 those of lanes 32..63
        const auto fsw = __builtin_amdgcn_permlane32_swap(fb, fb, false, false);
        const unsigned fb1 = fsw[0], fb2 = fsw[1];
        const f32x4 *xpa = xa + fb;                               // own frames
        const float *xp4 = x4 + fb;
        const u32x4 *bp1 = bimg + (size_t)kh * F + fb1, *bp2 = bimg + (size_t)kh * F + fb2;

        float sum[K] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int prevL = 0;
        bool listed = bad_tile;
        for (int ti = 0; ti < count; ++ti) {
            const int L = t_len[ti], tcol = t_id[ti];
            // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums -- continued from the shorter template's
#pragma unroll 4
            for (int i = prevL; i < L; ++i) {
                const f32x4 v = xpa[i];
                sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w; sum[4] += xp4[i];
            }
            prevL = L;
            float nmus[K], dl[K];   // -mu s;  (mu - o) s
            float dd = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float mu = sum[j] / (float)L;
                nmus[j] = -(mu * s);
                dl[j] = mu * s - pw[j] * s;
                dd = fmaf(dl[j], dl[j], dd);
            }
            // largest kappa = 1 / (s |x_f - mu|) this pass may meet: ratio <= 1 + |mu - o| s kappa, the f16 parts, the reference's own range
            float klim = dd > 0.f ? (kRRatio - 1.f) * __builtin_amdgcn_rsqf(dd) : kRKappaMax;
            klim = fminf(fminf(klim, kRKappaMax), kDtwFixLimit / s);   // (kappa of the unscaled frame = kappa s)
            const unsigned char *aimg_t = a_lds + a_off[ti] + kh * 16;
            const float *rows_t = unit + (size_t)tcol * Lpad * K;   // wave-uniform: scalar loads (the buffer ends with 32 rows of slack, rp_ctx.cpp)
            const float abandon_cost = abandon_nc * (float)(L + L);

            float Q[B + 1];
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[q] = RP_INF;
            Q[W - 1] = 0.f;
            // rows 1..16 in their slots
            u32x4 A0 = {0u, 0u, 0u, 0u}, A1 = {0u, 0u, 0u, 0u};
            {
                const int r0 = jA == 0 ? NS : jA;
                const u32x4 av = *reinterpret_cast<const u32x4 *>(aimg_t + (r0 - 1) * 32);
                if (hsA == 0) A0 = av; else A1 = av;
            }
            v16f G;   // rows 1 .. W + 2 are all the first two columns' bands hold; the others enter through RG_ROW before they are read
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                float g = 0.f;
                if (j >= 1 && j <= W + 2) {
                    const float *ar = rows_t + (j - 1) * K;
                    g = ar[0] * dl[0];
#pragma unroll
                    for (int k2 = 1; k2 < K; ++k2) g = fmaf(ar[k2], dl[k2], g);
                }
                G[j] = g;
            }
            v16f acc;   // one tile: the instruction pair of column c + 1 is issued behind the last cell of column c and lands under the norm and row work
            u32x4 Bc1, Bc2;
            float xr[K], kap[2], kmax = 0.f;
            float nd_[K], nbb_ = 0.f, nk_ = 0.f, ng_ = 0.f;

// kappa of the frame in xr[] (the window's column cc): 1 / (s |x - mu|), 0 for the zero vector (similarity 0, comparator.rs:43-47)
#define RG_NORM(dst)                                                                                                          \
    {                                                                                                                         \
        float d_[K];                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < K; ++j) d_[j] = fmaf(xr[j], s, nmus[j]);                                        \
        float bb_ = d_[0] * d_[0];                                                                                            \
        _Pragma("unroll") for (int j = 1; j < K; ++j) bb_ = fmaf(d_[j], d_[j], bb_);                                          \
        dst = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;                                                                   \
        kmax = fmaxf(kmax, dst);                                                                                              \
    }
#define RG_LOAD(cc)                                                                                                           \
    {                                                                                                                         \
        Bc1 = bp1[(cc) - 1]; Bc2 = bp2[(cc) - 1];                                                                             \
        const f32x4 v_ = xpa[(cc) - 1];                                                                                       \
        xr[0] = v_.x; xr[1] = v_.y; xr[2] = v_.z; xr[3] = v_.w; xr[4] = xp4[(cc) - 1];                                         \
    }
#define RG_MFMA()                                                                                                             \
    {                                                                                                                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), __builtin_bit_cast(f16x8, Bc1), G, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A1), __builtin_bit_cast(f16x8, Bc2), acc, 0, 0, 0);           \
    }
// template row rn (1-based) enters slot rn mod 16: its image into the lanes that feed the slot, its G into the ring
#define RG_ROW(rn, slot)                                                                                                      \
    {                                                                                                                         \
        const int ri_ = (rn) - 1;                                                                                             \
        if (jA0 == (slot)) A0 = *reinterpret_cast<const u32x4 *>(aimg_t + ri_ * 32);                                          \
        if (jA1 == (slot)) A1 = *reinterpret_cast<const u32x4 *>(aimg_t + ri_ * 32);                                          \
        RG_G_OF_ROW(slot)                                                                                                     \
    }
#define RG_G_OF_ROW(slot) G[slot] = ng_;
// The norm of column c + 1 and the G of the row that enters the ring stand in pieces BETWEEN the cells of column c (none of them depends on
// the cells; the slot the new row takes is outside the band of the matrix instructions that follow): in one piece behind the cells they
// were a chain of ~20 dependent instructions the wave waited on by itself -- 3 % of the kernel, measured interleaved on one box
#define RG_PIECES_AT(q)                                                                                                       \
    if ((q) == (1 * B) / 10) { _Pragma("unroll") for (int j = 0; j < K; ++j) nd_[j] = fmaf(xr[j], s, nmus[j]); }               \
    if ((q) == (4 * B) / 10) { nbb_ = nd_[0] * nd_[0]; _Pragma("unroll") for (int j = 1; j < K; ++j) nbb_ = fmaf(nd_[j], nd_[j], nbb_); } \
    if ((q) == (7 * B) / 10) { nk_ = nbb_ > 0.f ? __builtin_amdgcn_rsqf(nbb_) : 0.f; kmax = fmaxf(kmax, nk_); }                \
    if ((q) == (8 * B) / 10) { ng_ = arn[0] * dl[0]; _Pragma("unroll") for (int k2 = 1; k2 < K; ++k2) ng_ = fmaf(arn[k2], dl[k2], ng_); } \
    __builtin_amdgcn_sched_barrier(0);
#define RG_NORM_TAIL(dst) { dst = nk_; }
// column c = c0 + u: rows r_q = c - W + 1 + q, q = 0..2W-1, sit in slot (u + q + 2 - W) mod 16
#define RG_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        RG_LOAD(c + 1)                                                                                                        \
        float arn[K];   /* the unit row that enters the band at the end of this step: requested now */                        \
        _Pragma("unroll") for (int k2 = 0; k2 < K; ++k2) arn[k2] = rows_t[(c + 1 + W) * K + k2];                              \
        __builtin_amdgcn_sched_barrier(0);   /* the requests go out HERE, a column's cells ahead of their use: left to the scheduler they sink \
                                                to just before the matrix instructions and every column waits out an LDS + scalar-load round trip */ \
        float up = RP_INF;                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            const float cost = fmaf(acc[sl], kap[u & 1], 1.f);                                                                \
            const float m = fminf(fminf(up, Q[q + 1]), Q[q]);                                                                 \
            float v = cost + m;                                                                                               \
            if (GUARD) v = (c - W + 1 + q >= 1) ? v : RP_INF;                                                                 \
            Q[q] = v;                                                                                                         \
            up = v;                                                                                                           \
            RG_PIECES_AT(q)                                                                                                   \
        }                                                                                                                     \
        asm volatile("" ::"v"(acc));   /* every row slot of the tile stays allocated until here: no other value moves into the registers \
                                          of the slots outside the band while the instruction pair that fills them is in flight */         \
        RG_MFMA()                                                                                                             \
        RG_NORM_TAIL(kap[(u + 1) & 1])                                                                                        \
        /* the row update below is a pair of exec-masked loads = basic-block boundaries: without these anchors the compiler sinks the \
           cells of a whole 16-column block behind them and keeps sixteen accumulator tiles alive */                                 \
        _Pragma("unroll") for (int q = 0; q < B; ++q) asm volatile("" : "+v"(Q[q]));                                          \
        asm volatile("" : "+v"(kap[(u + 1) & 1]));                                                                            \
        RG_ROW(c + 2 + W, (u + 3 + W) % NS)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    } while (0)

            RG_LOAD(1)
            RG_NORM(kap[0])
            RG_MFMA()
            int c0 = 1;
            bool dead = false;
            {   // first block: cells of rows < 1 stay +inf (L >= 16)
#pragma unroll
                for (int u = 0; u < NS; ++u) { const int c = c0 + u; RG_STEP(true); }
            }
#define RG_ABANDON_CHECK()                                                                                                    \
    if (abandon_nc < RP_INF) {                                                                                                \
        float m_ = Q[0];                                                                                                      \
        _Pragma("unroll") for (int q = 1; q < B; ++q) m_ = fminf(m_, Q[q]);                                                   \
        if (!__any(valid && m_ <= abandon_cost)) dead = true;                                                                 \
    }
            for (c0 = 1 + NS; c0 + NS - 1 <= L; c0 += NS) {
                RG_ABANDON_CHECK()
                if (dead) break;
#pragma unroll
                for (int u = 0; u < NS; ++u) { const int c = c0 + u; RG_STEP(false); }
            }
            if (!dead && c0 <= L) {
                RG_ABANDON_CHECK()
                if (!dead) {
#pragma unroll
                    for (int u = 0; u < NS - 1; ++u) {   // the last L mod 16 columns
                        const int c = c0 + u;
                        if (c <= L) RG_STEP(false);
                    }
                }
            }
#undef RG_ABANDON_CHECK
#undef RG_STEP
#undef RG_PIECES_AT
#undef RG_G_OF_ROW
#undef RG_NORM_TAIL
#undef RG_ROW
#undef RG_MFMA
#undef RG_LOAD
#undef RG_NORM
            // D[m - 1][n] with m == n == L (dtw.rs:101): band position q = W - 2
            const float nc = Q[W - 2] / (float)(L + L);
            const float sc = dead ? 0.f : 1.f / (1.f + expf((nc - score_ref) / score_ref));
            if (valid) scores[(strm * out_win_pitch + (size_t)w) * T + tcol] = sc;
            listed = listed || kmax > klim;
        }
        if (valid && listed) {
            if (rows_list) rows_list[1 + atomicAdd(rows_list, 1u)] = (uint32_t)(strm * n_win + (size_t)w);   // row ids s * n_win + w, as the gate's list
            else dtw_fix_append(fix, strm * out_win_pitch + (size_t)w, (uint32_t)(chunk_base + (int)ci));
        }
    }
}

size_t dtw_ragged_lds_bytes(const TemplatesDev &t, size_t n_win, int *frames_cap) {
    const int nseg = (int)((kRTile - 1) / n_win) + 2;
    const int F = ((kRTile + nseg * (t.max_len + 2)) + 3) & ~3;
    if (frames_cap) *frames_cap = F;
    return (size_t)t.rag_a_cap + (size_t)F * (kRK * sizeof(float) + 32) + 3 * kChunkMax * sizeof(int);
}

bool dtw_ragged_supported(const TemplatesDev &t, int band, size_t n_win, float score_ref) {
    // OPT-IN (rp_ctx_set_arithmetic(RP_ARITH_FAST_SPLIT, ragged_matrix = 1) / RP_CTX_RAGGED_MATRIX, read per call).  Measured on MI355X at the reference's own shape (65 536 streams x templates of 108 / 96 /
    // 90 / 93 / 102 frames): 16.5-17.7 ms against 18.2-18.6 ms for the register kernels -- 0-8 % of the step -- while its scores differ
    // from theirs in the 7th digit, so taking it by default would end the bit-equality of offline batches with live-stream batches, the
    // gated / detect-only forms and the single-stream handle for exactly the references users ship.  Not worth it; DESIGN.md §4.2b has
    // the cost model (per (window, template, column): 3 vector ops per band cell + the frame's norm + the mean term = 155 issue cycles
    // and two matrix instructions that take VALU issue slots with them, against 257 for the register kernels and 98 for equal lengths).
    if (!t.arith_ragged() || t.K != kRK || !t.rimg || t.rag_count == 0 || t.max_diff != 0) return false;
    if (!(score_ref >= kDtwRaggedMinScoreRef)) return false;
    if (band < 3 || band > 5) return false;
    if (n_win < 64) return false;                  // tiles of 512 consecutive windows: a handful of stream segments each
    if (t.rag_min_len < kRSlots) return false;    // the first 16 columns are one guarded block
    return dtw_ragged_lds_bytes(t, n_win, nullptr) <= 160 * 1024;   // (two workgroups per CU up to 80 KB: BASELINE-sized calls)
}

hipError_t launch_dtw_ragged(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch,
                             size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float abandon_nc, bool list_rows) {
    if (t.rag_count <= 0 || S == 0 || n_win == 0) return hipSuccess;
    if (!wk.fix || !wk.rag_prep || wk.rag_streams < S) return hipErrorInvalidValue;
    if (list_rows && (!wk.rag_list || wk.rag_rows < S * n_win * (size_t)t.rag_count || S * n_win > 0xffffffffULL)) return hipErrorInvalidValue;
    dtw_mark(wk, kDtwRanRagged | kDtwRanF16x2);
    int F = 0;
    const size_t lds = dtw_ragged_lds_bytes(t, n_win, &F);
    const size_t n_tiles = (S * n_win + kRTile - 1) / kRTile;
    size_t groups = 2 * (size_t)device_cu_count() / (size_t)t.rag_count;
    if (groups < 1) groups = 1;
    if (groups > n_tiles) groups = n_tiles;
    const size_t blocks = groups * (size_t)t.rag_count;
    uint32_t *rows_list = list_rows ? wk.rag_list : nullptr;
    if (rows_list)
        if (hipError_t e = hipMemsetAsync(rows_list, 0, sizeof(uint32_t), st); e != hipSuccess) return e;
    const size_t n_frames = n_win + (size_t)t.max_len - 1;   // frames of a stream the call scores
    hipLaunchKernelGGL(ragged_prep_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, mfcc, frame_pitch, S, first_win,
                       first_win + n_frames <= frame_pitch ? n_frames : frame_pitch - first_win, wk.rag_prep);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
#define RP_LAUNCH_RAGGED(WW)                                                                                                        \
    do {                                                                                                                            \
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_ragged_kernel<WW>), 160 * 1024); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dtw_ragged_kernel<WW>), dim3((unsigned)blocks), dim3(64 * kDtwRaggedWaves), lds, st, mfcc, frame_pitch, S, first_win, \
                           n_win, out_win_pitch, t.chunks, t.rag_first, (unsigned)t.rag_count, t.lens, t.rag_off, reinterpret_cast<const uint4 *>(t.rimg), \
                           t.unit, t.Lpad, t.T, score_ref, scores, t.max_len, F, t.rag_a_cap, abandon_nc, wk.rag_prep, rows_list, wk.fix); \
    } while (0)
    switch (band) {
    case 3: RP_LAUNCH_RAGGED(3); break;
    case 4: RP_LAUNCH_RAGGED(4); break;
    case 5: RP_LAUNCH_RAGGED(5); break;
    default: return hipErrorNotSupported;
    }
#undef RP_LAUNCH_RAGGED
    return hipGetLastError();
}

}  // namespace rp
