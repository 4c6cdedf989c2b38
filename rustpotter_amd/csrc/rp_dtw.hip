// rp_dtw.hip -- window scoring: dtw_band_kernel / dtw_band_wide_kernel / dtw_generic_kernel
// (src/mfcc/dtw.rs:56-105 + comparator.rs + normalizer.rs + wakeword_comp.rs:22-27; one lane per window, band and ring
// in registers) and aggregate_kernel (src/wakewords/comp/wakeword_comp.rs:38-49,108-139).  DESIGN.md §4.2.
#include "rp_device.h"

#include <cmath>
#include <cstdlib>

namespace rp {

// -------------------------------------------------------------------------- DTW
// Scoring of one (window, template) pair, reference semantics:
//   window = frames [w, w+L) of the stream (cut to the template length L keeping the
//   OLDEST frames, wakeword_comp.rs:22-27), column-mean normalised (normalizer.rs);
//   D[r][c] = (1 - cos(a[r-1], b[c-1])) + min(D[r-1][c], D[r][c-1], D[r-1][c-1]) on the band
//   c in [r-W, r+W-1]; result D[m-1][n] (dtw.rs:101); score = 1/(1+exp((cost/(m+n)-ref)/ref)).
// Template rows are pre-scaled to unit length on the host and window frames are scaled
// to unit length here, so a cell costs K fused multiply-adds instead of three dot
// products, a sqrt and a divide (comparator.rs:28-48); zero vectors stay zero, which
// reproduces the reference's "magnitude == 0 -> similarity 0".
constexpr int kDtwWin = 64;   // windows per wave

// The averaged-template gate's hand-over to the template kernels (launch_dtw_gated).  count == nullptr: no gate, every
// window is scored.  list != nullptr: LIST mode -- the lanes take the listed rows (windows that passed) and read their frames
// from global memory; the launch does nothing when the list is dense (*count >= dense_min).  list == nullptr with a count:
// DENSE mode -- the ordinary LDS-staged launch over every window, which does nothing unless the list is dense.  (Nearly
// everything passing is the common case at the reference's default threshold; scoring all rows through the staged kernel is
// then ~6 % cheaper than gathering them one by one.  Both launches are always issued; one of them exits on a scalar compare.)
struct GateList {
    const uint32_t *list = nullptr, *count = nullptr;
    uint32_t dense_min = 0;
    // entries the list may hold per row of the call (1: the gate lists a row once; dtw_ragged_kernel's list holds a window once per ragged
    // chunk that could not resolve it): the list-mode grids cover S x n_win x list_mult entries
    uint32_t list_mult = 1;
    // Early abandon (detect-only calls in ScoreMode::Max): a DTW whose cheapest band cell already costs more than
    // abandon_nc * (m + n) cannot end with a score above the detection threshold (cell costs are >= 0 and every warping
    // path crosses every row), so a wave whose 64 windows x templates are ALL past that bound stops and reports score 0 for
    // them.  Windows that can still fire are never touched: their wave runs to the end, with every template exact.
    // +inf: off (the per-window score arrays are part of the call's result).  See dtw_abandon_nc().
    float abandon_nc = __builtin_inff();
    const DtwFusedAgg *fuse = nullptr;  // ScoreMode::Max folded into the matrix-core kernel (rp_kernels.h); set by launch_dtw only
    // DtwWork::fix: windows with a frame whose squared norm leaves kDtwNormLo..kDtwFixLimit are listed for dtw_ref_kernel (the
    // reference's sqrt(dot_a * dot_b) is not scale invariant there, comparator.rs:42-47); every launcher sets it
    uint32_t *fix = nullptr;
    uint32_t *sched = nullptr;   // DtwWork::sched for the matrix-core launches
    uint32_t *ran = nullptr;     // DtwWork::ran
    DtwWork wk_all;              // the call's whole DtwWork (dtw_ragged_kernel's blocks)
    bool padded = false;         // the frame array ends with slack: the register kernels' list mode may follow dtw_ragged_kernel
    DtwWork work() const { DtwWork w = wk_all; w.sched = sched; w.fix = fix; w.ran = ran; return w; }
};

// One wave = 64 consecutive windows of one stream x one chunk of TC same-length templates.
// Per lane: the window's column means, a ring of the 2W unit-length window frames inside the
// band (shared by all TC templates), and TC bands of 2W+1 running costs held as register pairs of
// two templates: one v_pk_fma_f32 forms the cosine costs of a band cell for both templates (the
// coefficient pair comes from scalar registers, the window component is broadcast by op_sel),
// one v_pk_add_f32 adds the two v_min3_f32 results.  Rows are unrolled 2W at a time so
// every ring slot and band index is a compile-time register.  Only the first 2W rows can touch
// columns c < 1 and need the +inf guard; columns c > n are never read back by an in-range cell
// (they only feed cells further right / below-right), so they are left unguarded.
// GX: lanes read their window's frames straight from global memory instead of an LDS stage: used when a
// stream contributes only a few windows per launch (streaming batches), so that the 64 lanes of a wave
// can belong to many different streams.
template <int K, int W, int TC, bool GX>
__global__ __launch_bounds__(kDtwWin) void dtw_band_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, unsigned n_chunks,
    int chunk_base, size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks,
    const float *__restrict__ dup, int T, float score_ref, float *__restrict__ scores, float *__restrict__ avg,
    int flat, size_t n_streams, GateList gl = GateList{}) {
    constexpr int B = 2 * W;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;  // odd pitch: conflict-free lane-strided LDS reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [64 + 2 (L + W)][KP]: up to two stream segments

    const unsigned tile = blockIdx.x % tiles;
    const unsigned ci = (blockIdx.x / tiles) % n_chunks;
    const int lane = threadIdx.x;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    // Lane -> (stream, window).  flat != 0: the 64 lanes are consecutive entries of the flattened
    // (stream, window) space, so a wave may straddle two streams (needs n_win >= 64) and no lane is
    // wasted on a ragged last tile; flat == 0: tiles never cross a stream.
    size_t s;
    int w;
    bool valid;
    const float *xl;
    if (GX) {
        static_assert(!GX || KP == K, "global-memory frames have pitch K");
        size_t f = (size_t)tile * kDtwWin + lane;
        if (gl.list) {
            // the windows that passed the averaged-template gate (gate_compact_kernel): row ids s * n_win + w.  The grid is
            // sized for every window; tiles past the list's end have nothing to do.
            const uint32_t n_listed = *gl.count;
            if ((size_t)tile * kDtwWin >= n_listed || (gl.dense_min && n_listed >= gl.dense_min)) return;
            valid = f < n_listed;
            f = gl.list[valid ? f : n_listed - 1];
        } else {
            if (gl.count && *gl.count < gl.dense_min) return;
            valid = f < n_streams * n_win;
        }
        s = valid ? f / n_win : 0;
        w = valid ? (int)(f - s * n_win) : 0;
        xl = mfcc + (s * frame_pitch + first_win + (size_t)w) * K;
    } else {
        if (gl.count && *gl.count < gl.dense_min) return;  // DENSE mode of the gate: only when (nearly) every window passed
        size_t sA, sB = 0;
        int wA, nA, nB = 0;
        if (flat) {
            const size_t f0 = (size_t)tile * kDtwWin;  // here `tiles` counts flattened tiles and there is no stream index
            sA = f0 / n_win;
            wA = (int)(f0 - sA * n_win);
            nA = (int)n_win - wA < kDtwWin ? (int)n_win - wA : kDtwWin;
            if (nA < kDtwWin && sA + 1 < n_streams) { sB = sA + 1; nB = kDtwWin - nA; }
        } else {
            sA = blockIdx.x / ((size_t)tiles * n_chunks);
            wA = (int)tile * kDtwWin;
            nA = (int)n_win - wA < kDtwWin ? (int)n_win - wA : kDtwWin;
        }
        const int segA = nA + L + W;  // frames staged for the first stream segment
        {
            const float *src = mfcc + sA * frame_pitch * K;
            const size_t g0 = first_win + wA;
            for (int i = lane; i < segA * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = g0 + f;
                xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        if (nB > 0) {
            const float *src = mfcc + sB * frame_pitch * K;
            const int segB = nB + L + W;
            for (int i = lane; i < segB * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = first_win + f;
                xs[(segA + f) * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        __syncthreads();
        const bool inA = lane < nA;
        valid = inA || (lane - nA < nB);
        s = inA ? sA : sB;
        w = inA ? wA + lane : lane - nA;
        xl = xs + (inA ? lane : (valid ? segA + lane - nA : 0)) * KP;
    }
    // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums
    float mu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = 0.f;
#ifndef RP_DTW_MEAN_UNROLL  // frames in flight per wait: left rolled, every frame of a window paid a whole LDS / memory round trip
#define RP_DTW_MEAN_UNROLL 10
#endif
#pragma unroll RP_DTW_MEAN_UNROLL
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] += xl[i * KP + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;

    v2f ring[B / 2][K];  // ring[j][k] = { y_slot(2j)[k], y_slot(2j+1)[k] }
#pragma unroll
    for (int j = 0; j < B / 2; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = (v2f){0.f, 0.f};

    float chk = 0.f;  // max over the frames seen of (squared norm, its reciprocal square root): > kDtwFixLimit -> listed for dtw_ref_kernel
#define RP_LOAD_COL(c, slot)                                                              \
    do {                                                                                  \
        float y_[K], bb_ = 0.f;                                                           \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            y_[k] = xl[((c)-1) * KP + k] - mu[k];                                         \
            bb_ = fmaf(y_[k], y_[k], bb_);                                                \
        }                                                                                 \
        const float inv_ = bb_ > 0.f ? rsqrtf(bb_) : 0.f;                                 \
        chk = fmaxf(fmaxf(chk, inv_), bb_); /* one v_max3_f32: the norm-range test */     \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            if (((slot)&1) == 0) ring[(slot) / 2][k].x = y_[k] * inv_;                    \
            else ring[(slot) / 2][k].y = y_[k] * inv_;                                    \
        }                                                                                 \
    } while (0)

#pragma unroll
    for (int c = 1; c < W; ++c) RP_LOAD_COL(c, c % B);

    // P[tp][q] = D[r-1][(r-1-W)+q] of templates (2tp, 2tp+1); row 0 has D[0][0] = 0 at q = W
    v2f P[TC / 2][B + 1];
#pragma unroll
    for (int t = 0; t < TC / 2; ++t) {
#pragma unroll
        for (int q = 0; q <= B; ++q) P[t][q] = (v2f){RP_INF, RP_INF};
        P[t][W] = (v2f){0.f, 0.f};
    }

    const float *rows = dup + ch->rows_off;
#define RP_ROWS(GUARD)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < B; ++u) {                                                    \
        const int r = r0 + u;                                                                          \
        if (r < L) { /* rows 1..m-1 only: row m is never read (dtw.rs:101) */                          \
            RP_LOAD_COL(r + W - 1, (u + W) % B);                                                       \
            _Pragma("unroll") for (int t = 0; t < TC / 2; ++t) {                                       \
                /* coefficients of the template pair, interleaved (t0,t1) per k: one scalar pair */    \
                const v2f *arow = reinterpret_cast<const v2f *>(rows + ((size_t)(r - 1) * (TC / 2) + t) * K * 2); \
                v2f a2[K];                                                                             \
                _Pragma("unroll") for (int k = 0; k < K; ++k) a2[k] = arow[k];                         \
                /* costs of the 2W band cells first (independent FMA chains), then the serial min chain */ \
                v2f d[B];                                                                              \
                _Pragma("unroll") for (int q = 0; q < B; ++q) d[q] = (v2f){1.f, 1.f};                  \
                _Pragma("unroll") for (int k = 0; k < K; ++k) {                                        \
                    _Pragma("unroll") for (int q = 0; q < B; ++q) {                                    \
                        const int slot = (1 + u + q + B - W) % B;                                      \
                        const v2f yy = (slot & 1) ? ring[slot / 2][k].yy : ring[slot / 2][k].xx;       \
                        d[q] = __builtin_elementwise_fma(-a2[k], yy, d[q]);                            \
                    }                                                                                  \
                }                                                                                      \
                v2f left = (v2f){RP_INF, RP_INF};                                                      \
                _Pragma("unroll") for (int q = 0; q < B; ++q) {                                        \
                    v2f m;                                                                             \
                    m.x = fminf(fminf(P[t][q + 1].x, left.x), P[t][q].x);                              \
                    m.y = fminf(fminf(P[t][q + 1].y, left.y), P[t][q].y);                              \
                    v2f v = d[q] + m;                                                                  \
                    if (GUARD) v = (r - W + q >= 1) ? v : (v2f){RP_INF, RP_INF};                       \
                    P[t][q] = v;                                                                       \
                    left = v;                                                                          \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }

    {
        const int r0 = 1;
        RP_ROWS(true)
    }
    const float abandon_cost = gl.abandon_nc * (float)(L + L);
    for (int r0 = 1 + B; r0 < L; r0 += B) {
        if (gl.abandon_nc < RP_INF) {  // wave-uniform; once per 2W rows
            bool alive = false;
#pragma unroll
            for (int t = 0; t < TC / 2; ++t) {
                v2f m = P[t][0];
#pragma unroll
                for (int q = 1; q < B; ++q) m = (v2f){fminf(m.x, P[t][q].x), fminf(m.y, P[t][q].y)};
                // the averaged template (tid == T) is never abandoned: its score is reported and gates (as in the band2 / wide
                // kernels); Templates::create keeps it out of this kernel's chunks today, this guard keeps that an optimisation
                alive = alive || (2 * t < ch->count && (m.x <= abandon_cost || ch->tid[2 * t] >= T)) ||
                        (2 * t + 1 < ch->count && (m.y <= abandon_cost || ch->tid[2 * t + 1] >= T));
            }
            if (!__any(alive && valid)) {
                if (valid) {
                    const size_t row = s * out_win_pitch + (size_t)w;
                    for (int t = 0; t < ch->count; ++t)
                        if (ch->tid[t] < T) scores[row * T + ch->tid[t]] = 0.f;
                    if (chk > kDtwFixLimit) dtw_fix_append(gl.fix, row, (uint32_t)(chunk_base + (int)ci));
                }
                return;
            }
        }
        RP_ROWS(false)
    }
#undef RP_ROWS
#undef RP_LOAD_COL

    if (valid) {
        const size_t row = s * out_win_pitch + (size_t)w;
        const float denom = (float)(L + L);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (t < ch->count) {
                const float cost = (t & 1) ? P[t / 2][W + 1].y : P[t / 2][W + 1].x;  // D[m-1][n] for m == n
                const float nc = cost / denom;
                const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
                const int tid = ch->tid[t];
                if (tid < T) scores[row * T + tid] = sc;
                else avg[row] = sc;
            }
        }
        if (chk > kDtwFixLimit) dtw_fix_append(gl.fix, row, (uint32_t)(chunk_base + (int)ci));
    }
}

// One template, TWO windows per lane.  A chunk that holds a single template (every template of a ragged reference such as
// the reference's own oye_casa_g.rpw, 108/96/90/93/102 frames; the averaged template) would leave one half of every packed
// instruction of dtw_band_kernel<.., 2> idle.  Here the two halves are two windows: a wave owns 128 consecutive entries
// of the flattened (stream, window) space, lane l entries l and l + 64; means, ring and band are all register pairs,
// the template coefficient is a scalar broadcast to both halves.  Same operations per cell as dtw_band_kernel, same
// bits.  GX / list as there (entries read their frames from global memory).
template <int K, int W, bool GX>
__global__ __launch_bounds__(kDtwWin) void dtw_band2_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, unsigned n_chunks,
    int chunk_base, size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks,
    const float *__restrict__ dup, int T, float score_ref, float *__restrict__ scores, float *__restrict__ avg,
    int flat, size_t n_streams, GateList gl = GateList{}) {
    constexpr int B = 2 * W, NW = 2 * kDtwWin;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [128 + 2 (L + W)][KP]: up to two stream segments

    const unsigned tile = blockIdx.x % tiles;
    const unsigned ci = (blockIdx.x / tiles) % n_chunks;
    const int lane = threadIdx.x;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    size_t s[2];
    int w[2];
    bool valid[2];
    const float *xl[2];
    if (GX) {
        static_assert(!GX || KP == K, "global-memory frames have pitch K");
        unsigned n_listed = 0;
        if (gl.list) {
            n_listed = *gl.count;
            if ((size_t)tile * NW >= n_listed || (gl.dense_min && n_listed >= gl.dense_min)) return;
        } else if (gl.count && *gl.count < gl.dense_min) return;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            size_t f = (size_t)tile * NW + (size_t)e * kDtwWin + lane;
            if (gl.list) {
                valid[e] = f < n_listed;
                f = gl.list[valid[e] ? f : n_listed - 1];
            } else {
                valid[e] = f < n_streams * n_win;
            }
            s[e] = valid[e] ? f / n_win : 0;
            w[e] = valid[e] ? (int)(f - s[e] * n_win) : 0;
            xl[e] = mfcc + (s[e] * frame_pitch + first_win + (size_t)w[e]) * K;
        }
    } else {
        if (gl.count && *gl.count < gl.dense_min) return;
        size_t sA, sB = 0;
        int wA, nA, nB = 0;
        if (flat) {
            const size_t f0 = (size_t)tile * NW;
            sA = f0 / n_win;
            wA = (int)(f0 - sA * n_win);
            nA = (int)n_win - wA < NW ? (int)n_win - wA : NW;
            if (nA < NW && sA + 1 < n_streams) { sB = sA + 1; nB = NW - nA; }
        } else {
            sA = blockIdx.x / ((size_t)tiles * n_chunks);
            wA = (int)tile * NW;
            nA = (int)n_win - wA < NW ? (int)n_win - wA : NW;
        }
        const int segA = nA + L + W;
        {
            const float *src = mfcc + sA * frame_pitch * K;
            const size_t g0 = first_win + wA;
            for (int i = lane; i < segA * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = g0 + f;
                xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        if (nB > 0) {
            const float *src = mfcc + sB * frame_pitch * K;
            const int segB = nB + L + W;
            for (int i = lane; i < segB * K; i += kDtwWin) {
                int f = i / K, k = i - f * K;
                size_t g = first_win + f;
                xs[(segA + f) * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ent = e * kDtwWin + lane;
            const bool inA = ent < nA;
            valid[e] = inA || (ent - nA < nB);
            s[e] = inA ? sA : sB;
            w[e] = inA ? wA + ent : ent - nA;
            xl[e] = xs + (inA ? ent : (valid[e] ? segA + ent - nA : 0)) * KP;
        }
    }
    const float *x0 = xl[0], *x1 = xl[1];
    // MfccNormalizer::normalize: sequential column sums, both windows at once
    v2f mu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = (v2f){0.f, 0.f};
#pragma unroll RP_DTW_MEAN_UNROLL
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] += (v2f){x0[i * KP + k], x1[i * KP + k]};
    }
    const float fl = (float)L;
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = (v2f){mu[k].x / fl, mu[k].y / fl};

    v2f ring[B][K];  // ring[slot][k] = unit-length frame of the slot, (window 0, window 1)
#pragma unroll
    for (int j = 0; j < B; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = (v2f){0.f, 0.f};

    v2f chk = (v2f){0.f, 0.f};  // the norm-range test of the two windows (see dtw_band_kernel)
#define RP_LOAD_COL2(c, slot)                                                                       \
    do {                                                                                            \
        v2f y_[K], bb_ = (v2f){0.f, 0.f};                                                           \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                             \
            y_[k] = (v2f){x0[((c)-1) * KP + k], x1[((c)-1) * KP + k]} - mu[k];                      \
            bb_ = __builtin_elementwise_fma(y_[k], y_[k], bb_);                                     \
        }                                                                                           \
        const v2f inv_ = (v2f){bb_.x > 0.f ? rsqrtf(bb_.x) : 0.f, bb_.y > 0.f ? rsqrtf(bb_.y) : 0.f}; \
        chk.x = fmaxf(fmaxf(chk.x, inv_.x), bb_.x); chk.y = fmaxf(fmaxf(chk.y, inv_.y), bb_.y);     \
        _Pragma("unroll") for (int k = 0; k < K; ++k) ring[slot][k] = y_[k] * inv_;                 \
    } while (0)

#pragma unroll
    for (int c = 1; c < W; ++c) RP_LOAD_COL2(c, c % B);

    v2f P[B + 1];
#pragma unroll
    for (int q = 0; q <= B; ++q) P[q] = (v2f){RP_INF, RP_INF};
    P[W] = (v2f){0.f, 0.f};

    const float *rows = dup + ch->rows_off;  // [len][1][K][2]: the template's coefficients (stored twice)
#define RP_ROWS2(GUARD)                                                                                \
    _Pragma("unroll") for (int u = 0; u < B; ++u) {                                                    \
        const int r = r0 + u;                                                                          \
        if (r < L) {                                                                                   \
            RP_LOAD_COL2(r + W - 1, (u + W) % B);                                                      \
            const float *arow = rows + (size_t)(r - 1) * K * 2;                                        \
            v2f d[B];                                                                                  \
            _Pragma("unroll") for (int q = 0; q < B; ++q) d[q] = (v2f){1.f, 1.f};                      \
            _Pragma("unroll") for (int k = 0; k < K; ++k) {                                            \
                const float a = arow[2 * k];                                                           \
                const v2f na = (v2f){-a, -a};                                                          \
                _Pragma("unroll") for (int q = 0; q < B; ++q)                                          \
                    d[q] = __builtin_elementwise_fma(na, ring[(1 + u + q + B - W) % B][k], d[q]);      \
            }                                                                                          \
            v2f left = (v2f){RP_INF, RP_INF};                                                          \
            _Pragma("unroll") for (int q = 0; q < B; ++q) {                                            \
                v2f m;                                                                                 \
                m.x = fminf(fminf(P[q + 1].x, left.x), P[q].x);                                        \
                m.y = fminf(fminf(P[q + 1].y, left.y), P[q].y);                                        \
                v2f v = d[q] + m;                                                                      \
                if (GUARD) v = (r - W + q >= 1) ? v : (v2f){RP_INF, RP_INF};                           \
                P[q] = v;                                                                              \
                left = v;                                                                              \
            }                                                                                          \
        }                                                                                              \
    }
    {
        const int r0 = 1;
        RP_ROWS2(true)
    }
    const int tid = ch->tid[0];
    const float abandon_cost = gl.abandon_nc * (float)(L + L);
    for (int r0 = 1 + B; r0 < L; r0 += B) {
        if (gl.abandon_nc < RP_INF && tid < T) {  // never for the averaged template: its score is reported and gates
            v2f m = P[0];
#pragma unroll
            for (int q = 1; q < B; ++q) m = (v2f){fminf(m.x, P[q].x), fminf(m.y, P[q].y)};
            const bool alive = (valid[0] && m.x <= abandon_cost) || (valid[1] && m.y <= abandon_cost);
            if (!__any(alive)) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (valid[e]) {
                        scores[(s[e] * out_win_pitch + (size_t)w[e]) * T + tid] = 0.f;
                        if ((e ? chk.y : chk.x) > kDtwFixLimit) dtw_fix_append(gl.fix, s[e] * out_win_pitch + (size_t)w[e], (uint32_t)(chunk_base + (int)ci));
                    }
                return;
            }
        }
        RP_ROWS2(false)
    }
#undef RP_ROWS2
#undef RP_LOAD_COL2

    const float denom = (float)(L + L);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (valid[e]) {
            const size_t row = s[e] * out_win_pitch + (size_t)w[e];
            const float cost = e ? P[W + 1].y : P[W + 1].x;  // D[m-1][n] for m == n
            const float nc = cost / denom;
            const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
            if (tid < T) scores[row * T + tid] = sc;
            else avg[row] = sc;
            if ((e ? chk.y : chk.x) > kDtwFixLimit) dtw_fix_append(gl.fix, row, (uint32_t)(chunk_base + (int)ci));
        }
    }
}

// Variant for wide frames (K = 16): the ring alone is 160 registers, so the band costs are formed one
// template at a time with two band cells per v_pk_fma_f32 (coefficient duplicated into a scalar pair)
// instead of holding the costs of a template pair for all 2W cells.
template <int K, int W, int TC, bool GX = false>
__global__ __launch_bounds__(kDtwWin) void dtw_band_wide_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, unsigned n_chunks,
    int chunk_base, size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks,
    const float *__restrict__ dup, int T, float score_ref, float *__restrict__ scores, float *__restrict__ avg,
    size_t n_streams = 0, GateList gl = GateList{}) {
    constexpr int B = 2 * W;
    constexpr int TP = TC >= 2 ? TC / 2 : 1;      // template pairs per row of `dup` (a one-template chunk is stored as a pair)
    // LDS rows get an odd pitch (conflict-free lane-strided reads); GX lanes read rows of pitch K from global memory
    constexpr int KP = GX ? K : ((K % 2 == 0) ? K + 1 : K);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [64 + L + W][KP]

    const unsigned tile = blockIdx.x % tiles;
    const unsigned ci = (blockIdx.x / tiles) % n_chunks;
    const int lane = threadIdx.x;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    size_t s;
    size_t wl;      // window index of this lane inside its stream
    bool valid;
    const float *xl;
    if (GX) {
        // lanes are consecutive entries of the flattened (stream, window) space, or of the gate's list (see dtw_band_kernel)
        size_t f = (size_t)tile * kDtwWin + lane;
        if (gl.list) {
            const uint32_t n_listed = *gl.count;
            if ((size_t)tile * kDtwWin >= n_listed || (gl.dense_min && n_listed >= gl.dense_min)) return;
            valid = f < n_listed;
            f = gl.list[valid ? f : n_listed - 1];
        } else {
            if (gl.count && *gl.count < gl.dense_min) return;
            valid = f < n_streams * n_win;
        }
        s = valid ? f / n_win : 0;
        wl = valid ? f - s * n_win : 0;
        xl = mfcc + (s * frame_pitch + first_win + wl) * K;
    } else {
        if (gl.count && *gl.count < gl.dense_min) return;
        s = blockIdx.x / ((size_t)tiles * n_chunks);
        const size_t w0 = first_win + (size_t)tile * kDtwWin;
        const int n_stage = kDtwWin + L + W;
        const float *src = mfcc + s * frame_pitch * K;
        for (int i = lane; i < n_stage * K; i += kDtwWin) {
            int f = i / K, k = i - f * K;
            size_t g = w0 + f;
            xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
        }
        __syncthreads();
        wl = (size_t)tile * kDtwWin + lane;
        valid = wl < n_win;
        xl = xs + lane * KP;
    }
    // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums
    float mu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = 0.f;
#ifndef RP_DTW_MEAN_UNROLL_WIDE  // 13 / 16 components per frame: four frames in flight keep the register count where it was
#define RP_DTW_MEAN_UNROLL_WIDE 4
#endif
#pragma unroll RP_DTW_MEAN_UNROLL_WIDE
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] += xl[i * KP + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;

    v2f ring[B / 2][K];  // ring[j][k] = { y_slot(2j)[k], y_slot(2j+1)[k] }
#pragma unroll
    for (int j = 0; j < B / 2; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = (v2f){0.f, 0.f};

    float chk = 0.f;  // max over the frames seen of (squared norm, its reciprocal square root): > kDtwFixLimit -> listed for dtw_ref_kernel
#define RP_LOAD_COL(c, slot)                                                              \
    do {                                                                                  \
        float y_[K], bb_ = 0.f;                                                           \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            y_[k] = xl[((c)-1) * KP + k] - mu[k];                                         \
            bb_ = fmaf(y_[k], y_[k], bb_);                                                \
        }                                                                                 \
        const float inv_ = bb_ > 0.f ? rsqrtf(bb_) : 0.f;                                 \
        chk = fmaxf(fmaxf(chk, inv_), bb_); /* one v_max3_f32: the norm-range test */     \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                   \
            if (((slot)&1) == 0) ring[(slot) / 2][k].x = y_[k] * inv_;                    \
            else ring[(slot) / 2][k].y = y_[k] * inv_;                                    \
        }                                                                                 \
    } while (0)

#pragma unroll
    for (int c = 1; c < W; ++c) RP_LOAD_COL(c, c % B);

    // P[t][q] = D[r-1][(r-1-W)+q] of template t; row 0 has D[0][0] = 0 at q = W
    float P[TC][B + 1];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
#pragma unroll
        for (int q = 0; q <= B; ++q) P[t][q] = RP_INF;
        P[t][W] = 0.f;
    }

    const float *rows = dup + ch->rows_off;
#define RP_ROWS(GUARD)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < B; ++u) {                                                    \
        const int r = r0 + u;                                                                          \
        if (r < L) { /* rows 1..m-1 only: row m is never read (dtw.rs:101) */                          \
            RP_LOAD_COL(r + W - 1, (u + W) % B);                                                       \
            _Pragma("unroll") for (int t = 0; t < TC; ++t) {                                           \
                const float *arow = rows + ((size_t)(r - 1) * TP + t / 2) * K * 2 + (t & 1);           \
                v2f dd[B / 2];                                                                         \
                _Pragma("unroll") for (int j = 0; j < B / 2; ++j) dd[j] = (v2f){1.f, 1.f};             \
                _Pragma("unroll") for (int k = 0; k < K; ++k) {                                        \
                    const v2f a2 = (v2f){arow[2 * k], arow[2 * k]};                                   \
                    _Pragma("unroll") for (int j = 0; j < B / 2; ++j)                                  \
                        dd[j] = __builtin_elementwise_fma(-a2, ring[j][k], dd[j]);                     \
                }                                                                                      \
                float left = RP_INF;                                                                   \
                _Pragma("unroll") for (int q = 0; q < B; ++q) {                                        \
                    const int slot = (1 + u + q + B - W) % B;                                          \
                    const float d = (slot & 1) ? dd[slot / 2].y : dd[slot / 2].x;                      \
                    float v = d + fminf(fminf(P[t][q + 1], left), P[t][q]);                            \
                    if (GUARD) v = (r - W + q >= 1) ? v : RP_INF;                                      \
                    P[t][q] = v;                                                                       \
                    left = v;                                                                          \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }

    {
        const int r0 = 1;
        RP_ROWS(true)
    }
    const float abandon_cost = gl.abandon_nc * (float)(L + L);
    for (int r0 = 1 + B; r0 < L; r0 += B) {
        if (gl.abandon_nc < RP_INF && ch->tid[0] < T) {
            bool alive = false;
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                float m = P[t][0];
#pragma unroll
                for (int q = 1; q < B; ++q) m = fminf(m, P[t][q]);
                alive = alive || (t < ch->count && m <= abandon_cost);
            }
            if (!__any(alive && valid)) {
                if (valid) {
                    for (int t = 0; t < ch->count; ++t) scores[(s * out_win_pitch + wl) * T + ch->tid[t]] = 0.f;
                    if (chk > kDtwFixLimit) dtw_fix_append(gl.fix, s * out_win_pitch + wl, (uint32_t)(chunk_base + (int)ci));
                }
                return;
            }
        }
        RP_ROWS(false)
    }
#undef RP_ROWS
#undef RP_LOAD_COL

    if (valid) {
        const size_t row = s * out_win_pitch + wl;
        const float denom = (float)(L + L);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            if (t < ch->count) {
                const float cost = P[t][W + 1];  // D[m-1][n] for m == n
                const float nc = cost / denom;
                const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
                const int tid = ch->tid[t];
                if (tid < T) scores[row * T + tid] = sc;
                else avg[row] = sc;
            }
        }
        if (chk > kDtwFixLimit) dtw_fix_append(gl.fix, row, (uint32_t)(chunk_base + (int)ci));
    }
}

// Fallback for any K / band size / m != n: one wave = 64 windows x one template, band
// and column means in LDS (lane-minor, conflict-free), costs evaluated per cell.
__global__ __launch_bounds__(64) void dtw_generic_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, unsigned tiles, size_t first_win,
    size_t n_win, size_t out_win_pitch, const int *__restrict__ lens, const float *__restrict__ unit, int Lpad, int K,
    int T, int t_first, int t_count, int max_len, int band, float score_ref, float *__restrict__ scores, float *__restrict__ avg,
    const float *gate_avg, float gate_threshold, uint32_t *__restrict__ fix) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KP = K | 1;
    const unsigned tile = blockIdx.x % tiles;
    const int t = t_first + (int)((blockIdx.x / tiles) % t_count);   // templates t_first .. t_first + t_count - 1 (T = the averaged one)
    const size_t s = blockIdx.x / ((size_t)tiles * t_count);
    const int lane = threadIdx.x;
    // the averaged-template gate (wakeword_comp.rs:85-93) at wave granularity: when none of the wave's 64 windows passed it,
    // the sample templates are not compared at all; a wave with a passing window scores all of its windows and the
    // aggregate pass writes 0 for the rejected ones, so both cases leave the same aggregates
    if (gate_avg) {
        const size_t wl = (size_t)tile * 64 + lane;
        const bool pass = wl < n_win && !(gate_avg[s * out_win_pitch + wl] < gate_threshold);
        if (!__any(pass)) return;
    }
    const size_t w0 = first_win + (size_t)tile * 64;
    const int m = lens[t];
    const int n = m < max_len ? m : max_len;  // window cut to the template length
    const int diff = m > n ? m - n : n - m;
    const int W = band > diff ? band : diff;
    const int B = 2 * W;

    const int n_stage = 64 + max_len - 1;
    float *xs = reinterpret_cast<float *>(smem);      // [n_stage][KP]
    float *mus = xs + (size_t)n_stage * KP;           // [K][64]
    float *Pb = mus + (size_t)K * 64;                 // [B+1][64]
    const float *src = mfcc + s * frame_pitch * K;
    for (int i = lane; i < n_stage * K; i += 64) {
        int f = i / K, k = i - f * K;
        size_t g = w0 + f;
        xs[f * KP + k] = g < n_frames_total ? src[g * K + k] : 0.f;
    }
    __syncthreads();
    const float *xl = xs + lane * KP;
    for (int k = 0; k < K; ++k) {
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += xl[i * KP + k];
        mus[k * 64 + lane] = sum / (float)n;
    }
    for (int q = 0; q <= B; ++q) Pb[q * 64 + lane] = RP_INF;
    Pb[W * 64 + lane] = 0.f;
    const float *trow = unit + (size_t)t * Lpad * K;
    float chk = 0.f;  // the norm-range test (see dtw_band_kernel)
    for (int r = 1; r < m; ++r) {
        float left = RP_INF;
        for (int q = 0; q < B; ++q) {
            const int c = r - W + q;
            float v = RP_INF;
            if (c >= 1 && c <= n) {
                // the arithmetic of the register kernels, operation for operation (unit-length frame first, then
                // d = 1 - a.y as one fma chain from 1): every DTW kernel gives the same bits for the same window
                float bb = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float y = xl[(c - 1) * KP + k] - mus[k * 64 + lane];
                    bb = fmaf(y, y, bb);
                }
                const float inv = bb > 0.f ? rsqrtf(bb) : 0.f;
                chk = fmaxf(fmaxf(chk, inv), bb);
                float d = 1.f;
                for (int k = 0; k < K; ++k) {
                    const float y = (xl[(c - 1) * KP + k] - mus[k * 64 + lane]) * inv;
                    d = fmaf(-trow[(r - 1) * K + k], y, d);
                }
                v = d + fminf(fminf(Pb[(q + 1) * 64 + lane], left), Pb[q * 64 + lane]);
            }
            Pb[q * 64 + lane] = v;
            left = v;
        }
    }
    if (tile * (size_t)64 + lane < n_win) {
        const int qs = n - (m - 1 - W);  // column n of row m-1
        float cost = (qs >= 0 && qs < B) ? Pb[qs * 64 + lane] : RP_INF;
        float nc = cost / (float)(m + n);
        float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
        size_t row = s * out_win_pitch + (size_t)tile * 64 + lane;
        if (t < T) scores[row * T + t] = sc;
        else avg[row] = sc;
        if (chk > kDtwFixLimit) dtw_fix_append(fix, row, kFixSpecTemplate | (uint32_t)t);
    }
}

// ---- the reference-shaped cell (comparator.rs:28-48) for what the scale-invariant kernels cannot reproduce ------------------
// dot_ab / sqrt(dot_a * dot_b) in f32 with `== 0 -> 0`: when the PRODUCT of the two squared norms underflows the reference's
// similarity becomes 0 (distance 1) although neither vector is zero, when it is subnormal the quotient loses bits, when it
// overflows the similarity is 0 again.  Every fast kernel lists the (window, chunk or template) pairs that met a frame outside
// kDtwNormLo..kDtwFixLimit (DtwWork::fix); this kernel rescores them cell by cell as the reference does -- three sequential dot
// products of the template row AS GIVEN and the mean-normalised frame, one sqrt, one divide -- and overwrites their scores.
// One lane per listed pair (a chunk's templates one after the other); frames and rows come from global memory, band and means
// sit in LDS lane-minor.  ALL mode (force_all, or more pairs than the list holds): every window x templates t_first ..
// t_first + t_count - 1 (template sets with a row outside kDtwNormLo..kDtwNormHiRow, TemplatesDev::ref_only).  agg_out: the pair's
// chunk holds every sample template (DtwFusedAgg) -- ScoreMode::Max and the stream's hot flag are rewritten too.
// Launched after the fast kernels of every call; without listed pairs each workgroup reads one word and leaves.  Otherwise the
// last workgroup to leave puts fix[0] and fix[1] back to zero.
__global__ __launch_bounds__(64) void dtw_ref_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, size_t first_win, size_t n_win, size_t out_win_pitch,
    size_t n_streams, const int *__restrict__ lens, const float *__restrict__ raw, int Lpad, int K, int T, int max_len, int band, int Wmax,
    float score_ref, const DtwChunk *__restrict__ chunks, float *__restrict__ scores, float *__restrict__ avg, uint32_t *fix,
    int force_all, int t_first, int t_count, float *__restrict__ agg_out, uint32_t *__restrict__ agg_hot, float agg_threshold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *mus = reinterpret_cast<float *>(smem);  // [K][64]
    float *Pb = mus + (size_t)K * 64;              // [2 Wmax + 1][64]
    const int lane = threadIdx.x;
    const uint32_t n_listed = fix[0];
    // nothing listed (every call on ordinary audio): leave without touching the counters -- 512 workgroups adding to one word
    // took 11 us, as long as a twentieth of a live-stream call's DTW launch.  With a list every workgroup sees the same non-zero
    // count (nobody resets it before all have checked in), so either all run the protocol below or none does.
    if (!force_all && n_listed == 0) return;
    const bool all = force_all || n_listed > kDtwFixCap;
    const size_t per_row = agg_out ? 1 : (size_t)t_count;
    const size_t n_entries = all ? n_streams * n_win * per_row : (size_t)n_listed;
    const unsigned long long *list = reinterpret_cast<const unsigned long long *>(fix + 2);
    if (blockIdx.x == 0 && lane == 0 && n_entries) atomicAdd(dtw_fix_stats(fix), (unsigned long long)n_entries);  // rp_ctx_dtw_ref_pairs
    for (size_t e = (size_t)blockIdx.x * 64 + lane; e < n_entries; e += (size_t)gridDim.x * 64) {
        size_t row;
        int tb, te, from_chunk = -1;   // templates tb .. te - 1, or the tid[] of chunk from_chunk
        if (all) {
            const size_t rw = e / per_row;
            const size_t s_ = rw / n_win;
            row = s_ * out_win_pitch + (rw - s_ * n_win);
            if (agg_out) { tb = t_first; te = t_first + t_count; }
            else { tb = t_first + (int)(e - rw * per_row); te = tb + 1; }
        } else {
            const unsigned long long v = list[e];
            row = (size_t)(v >> 24);
            const uint32_t spec = (uint32_t)v & kFixSpecMask;
            if (spec & kFixSpecTemplate) { tb = (int)(spec & (kFixSpecTemplate - 1)); te = tb + 1; }
            else { from_chunk = (int)spec; tb = 0; te = chunks[from_chunk].count; }
        }
        const size_t s = row / out_win_pitch, w = row - s * out_win_pitch;
        const size_t f0 = s * frame_pitch + first_win + w;       // the window's first frame
        const float *xw = mfcc + f0 * K;   // only columns 1..n are read: the window's own frames
        float best = 0.f;
        int mean_n = -1;
        for (int ti = tb; ti < te; ++ti) {
            const int t = from_chunk >= 0 ? chunks[from_chunk].tid[ti] : ti;
            const int m = lens[t];
            const int n = m < max_len ? m : max_len;  // window cut to the template length (wakeword_comp.rs:22-27)
            const int diff = m > n ? m - n : n - m;
            const int W = band > diff ? band : diff;  // dtw.rs:64-67
            const int B = 2 * W;
            if (n != mean_n) {  // MfccNormalizer::normalize (normalizer.rs:17-29): sequential column sums over the n frames
                for (int k = 0; k < K; ++k) {
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += xw[(size_t)i * K + k];
                    mus[k * 64 + lane] = sum / (float)n;
                }
                mean_n = n;
            }
            for (int q = 0; q <= B; ++q) Pb[q * 64 + lane] = RP_INF;
            Pb[W * 64 + lane] = 0.f;
            const float *trow = raw + (size_t)t * Lpad * K;
            for (int r = 1; r < m; ++r) {   // rows 1..m-1: row m is never read (dtw.rs:101)
                float left = RP_INF;
                for (int q = 0; q < B; ++q) {
                    const int c = r - W + q;
                    float v = RP_INF;
                    if (c >= 1 && c <= n) {
                        float dot_ab = 0.f, dot_a = 0.f, dot_b = 0.f;
                        for (int k = 0; k < K; ++k) {
                            const float ca = trow[(r - 1) * K + k];
                            const float cb = xw[(size_t)(c - 1) * K + k] - mus[k * 64 + lane];
                            dot_ab += ca * cb;   // -ffp-contract=off: a multiply and an add, as the reference
                            dot_a += ca * ca;
                            dot_b += cb * cb;
                        }
                        const float magnitude = sqrtf(dot_a * dot_b);
                        const float sim = magnitude == 0.f ? 0.f : dot_ab / magnitude;
                        v = (1.f - sim) + fminf(fminf(Pb[(q + 1) * 64 + lane], left), Pb[q * 64 + lane]);
                    }
                    Pb[q * 64 + lane] = v;
                    left = v;
                }
            }
            const int qs = n - (m - 1 - W);  // column n of row m-1
            const float cost = (qs >= 0 && qs < B) ? Pb[qs * 64 + lane] : RP_INF;
            const float nc = cost / (float)(m + n);
            const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
            if (t < T) { scores[row * T + t] = sc; best = fmaxf(best, sc); }
            else avg[row] = sc;
        }
        if (agg_out) {
            agg_out[row] = best;
            if (agg_hot && best > agg_threshold) agg_hot[s] = 1u;
        }
    }
    __syncthreads();
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(fix + 1, 1u) == gridDim.x - 1) { fix[0] = 0; fix[1] = 0; }
    }
}

// A launcher that fails between its fast kernels and the list pass must not leave pairs of THIS call (rows of these arrays) listed for
// the next one: the call's words go back to zero behind whatever was already queued.  Returns the error it was given.
static hipError_t dtw_abort(hipStream_t st, const DtwWork &wk, hipError_t e) {
    if (wk.fix) (void)hipMemsetAsync(wk.fix, 0, 2 * sizeof(uint32_t), st);
    if (wk.sched) (void)hipMemsetAsync(wk.sched, 0, 2 * (size_t)kDtwSchedChunks * sizeof(uint32_t), st);
    (void)hipGetLastError();
    return e;
}

// The pass behind every fast launch: rescoring of the listed pairs (see dtw_ref_kernel).  force_all: every window x templates
// t_first .. t_first + t_count - 1 (index T = the averaged template).
static hipError_t launch_dtw_ref(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                                 size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, float *scores, float *avg,
                                 bool force_all, int t_first, int t_count, const DtwFusedAgg *fuse = nullptr) {
    if (!wk.fix || !t.raw) return hipErrorInvalidValue;
    const int Wmax = band > t.max_diff ? band : t.max_diff;
    const size_t lds = ((size_t)t.K * 64 + (size_t)(2 * Wmax + 1) * 64) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorMemoryAllocation;
    if (lds > 64 * 1024)
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_ref_kernel), 160 * 1024); e != hipSuccess) return e;
    // list mode: four workgroups per CU -- with nothing listed each reads one word and leaves; with more pairs than the list holds
    // (kDtwFixCap: pathological input) the kernel falls back to every window of the call, and a grid of one workgroup per CU would
    // walk them sixteen thousand lanes wide
    size_t blocks = 4 * (size_t)device_cu_count();
    if (force_all) dtw_mark(wk, kDtwRanRefAll);
    if (force_all) {
        const size_t need = (S * n_win * (size_t)(fuse ? 1 : t_count) + 63) / 64;
        blocks = need < 8 * (size_t)device_cu_count() ? (need ? need : 1) : 8 * (size_t)device_cu_count();
    }
    const bool fused = fuse && fuse->agg;
    hipLaunchKernelGGL(dtw_ref_kernel, dim3((unsigned)blocks), dim3(64), lds, st, mfcc, frame_pitch, frame_pitch, first_win, n_win, out_win_pitch, S,
                       t.lens, t.raw, t.Lpad, t.K, t.T, t.max_len, band, Wmax, score_ref, t.chunks, scores, avg, wk.fix, force_all ? 1 : 0, t_first,
                       t_count, fused ? fuse->agg : nullptr, fused ? fuse->hot : nullptr, fused ? fuse->threshold : 0.f);
    return hipGetLastError();
}

template <int K, int W, int TC>
static hipError_t launch_dtw_class(hipStream_t st, const TemplatesDev &t, int chunk_base, int n_chunks, const float *mfcc, size_t S,
                                   size_t frame_pitch, size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch,
                                   float score_ref, float *scores, float *avg, bool few_windows, GateList gl = GateList{}) {
    if (n_chunks <= 0) return hipSuccess;
    dtw_mark(gl.work(), kDtwRanRegister);
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    if ((few_windows || gl.list) && KP == K) {
        // streams contribute fewer than 64 windows each (or the windows come from a list): lanes of a wave span many
        // streams and read their frames from global memory (the caller guarantees W*K floats of slack after the last
        // stream's frames)
        const size_t ft = (S * n_win * (gl.list ? (size_t)gl.list_mult : 1) + kDtwWin - 1) / kDtwWin;
        const size_t blocks = ft * (size_t)n_chunks;
        if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
        hipLaunchKernelGGL((dtw_band_kernel<K, W, TC, (KP == K)>), dim3((unsigned)blocks), dim3(kDtwWin), 0, st, mfcc, frame_pitch,
                           frame_pitch, (unsigned)ft, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                           t.chunks, t.dup, t.T, score_ref, scores, avg, 1, S, gl);
        return hipGetLastError();
    }
    if (gl.list) return hipErrorNotSupported;
    // flattened (stream, window) tiling when every stream has at least one full tile of windows
    const int flat = (n_win >= (size_t)kDtwWin && S > 1) ? 1 : 0;
    const size_t ft = flat ? (S * n_win + kDtwWin - 1) / kDtwWin : tiles;
    const size_t blocks = flat ? ft * (size_t)n_chunks : tiles * (size_t)n_chunks * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)(kDtwWin + 2 * (t.max_len + W)) * KP * sizeof(float);
    if (lds > 64 * 1024)
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_band_kernel<K, W, TC, false>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((dtw_band_kernel<K, W, TC, false>), dim3((unsigned)blocks), dim3(kDtwWin), lds, st, mfcc, frame_pitch,
                       frame_pitch, (unsigned)ft, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                       t.chunks, t.dup, t.T, score_ref, scores, avg, flat, S, gl);
    return hipGetLastError();
}

// chunks of ONE template: two windows per lane (dtw_band2_kernel)
template <int K, int W>
static hipError_t launch_dtw_single_chunks(hipStream_t st, const TemplatesDev &t, int chunk_base, int n_chunks, const float *mfcc, size_t S,
                                           size_t frame_pitch, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref,
                                           float *scores, float *avg, bool few_windows, GateList gl = GateList{}) {
    if (n_chunks <= 0) return hipSuccess;
    dtw_mark(gl.work(), kDtwRanRegister);
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    constexpr int NW = 2 * kDtwWin;
    if ((few_windows || gl.list) && KP == K) {
        const size_t ft = (S * n_win * (gl.list ? (size_t)gl.list_mult : 1) + NW - 1) / NW;
        const size_t blocks = ft * (size_t)n_chunks;
        if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
        hipLaunchKernelGGL((dtw_band2_kernel<K, W, (KP == K)>), dim3((unsigned)blocks), dim3(kDtwWin), 0, st, mfcc, frame_pitch,
                           frame_pitch, (unsigned)ft, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                           t.chunks, t.dup, t.T, score_ref, scores, avg, 1, S, gl);
        return hipGetLastError();
    }
    if (gl.list) return hipErrorNotSupported;
    const size_t tiles = (n_win + NW - 1) / NW;
    const int flat = (n_win >= (size_t)NW && S > 1) ? 1 : 0;
    const size_t ft = flat ? (S * n_win + NW - 1) / NW : tiles;
    const size_t blocks = flat ? ft * (size_t)n_chunks : tiles * (size_t)n_chunks * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)(NW + 2 * (t.max_len + W)) * KP * sizeof(float);
    if (lds > 64 * 1024)
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_band2_kernel<K, W, false>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((dtw_band2_kernel<K, W, false>), dim3((unsigned)blocks), dim3(kDtwWin), lds, st, mfcc, frame_pitch,
                       frame_pitch, (unsigned)ft, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                       t.chunks, t.dup, t.T, score_ref, scores, avg, flat, S, gl);
    return hipGetLastError();
}

template <int K, int W, int TC>
static hipError_t launch_dtw_wide(hipStream_t st, const TemplatesDev &t, int cls, int n_chunks, const float *mfcc, size_t S,
                                  size_t frame_pitch, size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch,
                                  float score_ref, float *scores, float *avg, bool few_windows = false, GateList gl = GateList{}, int chunk_base = -1) {
    if (n_chunks <= 0) return hipSuccess;
    dtw_mark(gl.work(), kDtwRanRegister);
    if (chunk_base < 0) chunk_base = t.class_first[cls];
    if (few_windows || gl.list) {
        // lanes span streams (few windows per stream) or come from the gate's list: frames read from global memory
        const size_t ft = (S * n_win + kDtwWin - 1) / kDtwWin;
        const size_t blocks = ft * (size_t)n_chunks;
        if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
        hipLaunchKernelGGL((dtw_band_wide_kernel<K, W, TC, true>), dim3((unsigned)blocks), dim3(kDtwWin), 0, st, mfcc, frame_pitch,
                           frame_pitch, (unsigned)ft, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                           t.chunks, t.dup, t.T, score_ref, scores, avg, S, gl);
        return hipGetLastError();
    }
    const size_t blocks = tiles * (size_t)n_chunks * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;
    const size_t lds = (size_t)(kDtwWin + t.max_len + W) * KP * sizeof(float);
    if (lds > 64 * 1024)
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_band_wide_kernel<K, W, TC, false>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((dtw_band_wide_kernel<K, W, TC, false>), dim3((unsigned)blocks), dim3(kDtwWin), lds, st, mfcc, frame_pitch,
                       frame_pitch, (unsigned)tiles, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch,
                       t.chunks, t.dup, t.T, score_ref, scores, avg, (size_t)0, gl);
    return hipGetLastError();
}

// wide frames (mfcc_size 13 / 16): single-template chunks (class 3, n1 of them) one template per lane, pairs (class 0) two
template <int K, int W>
static hipError_t launch_dtw_wide_all(hipStream_t st, const TemplatesDev &t, int n1, const float *mfcc, size_t S, size_t frame_pitch,
                                      size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores,
                                      float *avg, bool few, GateList gl = GateList{}, bool padded = false) {
    // every length at least three times, band 5, rows with slack behind them: the sample templates on the matrix cores
    // (rp_dtw_mfma_wide.hip), the averaged template -- if it is to be scored here -- through the one-template register kernel
    const bool rows_ok = W == 5 && (padded || few || gl.list);
    const bool three_part = rows_ok && dtw_mfma_wide3_supported(t, W);                       // the default arithmetic: chunks of four
    const bool two_part = rows_ok && !three_part && dtw_mfma_wide_supported(t, W, score_ref);  // RP_ARITH_FAST_SPLIT: chunks of eight
    if (three_part || two_part) {
        if (t.has_avg && n1 == t.class_count[3])
            if (hipError_t e = launch_dtw_wide<K, W, 1>(st, t, 3, 1, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg,
                                                        few, gl, t.class_first[3] + t.class_count[3] - 1); e != hipSuccess) return e;
        return (three_part ? launch_dtw_mfma_wide3 : launch_dtw_mfma_wide)(st, gl.work(), t, W, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref,
                                                                          scores, avg, gl.list, gl.count, gl.dense_min, gl.abandon_nc);
    }
    if (hipError_t e = launch_dtw_wide<K, W, 1>(st, t, 3, n1, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl); e != hipSuccess) return e;
    return launch_dtw_wide<K, W, 2>(st, t, 0, t.class_count[0], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
}

// dispatch on the (mfcc_size, band) pairs the wide kernels are built for
#define RP_WIDE_DISPATCH(CALL)                                                                 \
    do {                                                                                       \
        if (t.K == 16) {                                                                       \
            switch (band) { case 3: return CALL(16, 3); case 4: return CALL(16, 4); case 5: return CALL(16, 5); default: return CALL(16, 6); } \
        } else {                                                                               \
            switch (band) { case 3: return CALL(13, 3); case 4: return CALL(13, 4); case 5: return CALL(13, 5); default: return CALL(13, 6); } \
        }                                                                                      \
    } while (0)

// Largest template tile the register kernels are built for at this (mfcc_size, band) (0 = only the generic
// kernel applies).  Built: mfcc_size 5 with band 3..6 (tile 8), mfcc_size 13 and 16 with band 3..6 (tile 2).
// ---- one DTW per wave, for a handful of windows (the single-stream API: three new windows per 30 ms chunk) -----
// The register kernels above give every lane a whole DTW: with 3 windows x 5 templates that is 15 busy lanes walking
// ~100 dependent rows each (35 us).  Here a workgroup of one wave owns ONE (window, template) pair: all 64 lanes
// normalise the window and form the band's cosine costs, then the 2W lanes of the band walk the recurrence along
// anti-diagonals (lane q handles band offset q; cell (r, q) is due at step 2r + q, when its left neighbour -- lane q-1,
// one step ago -- and its upper neighbour -- lane q+1, one step ago -- are one DPP lane shift away).  Same operations
// per cell as dtw_band_kernel, same results.
__global__ __launch_bounds__(64) void dtw_single_kernel(
    const float *__restrict__ mfcc, size_t n_frames_total, size_t first_win, unsigned n_win, size_t out_win_pitch,
    const int *__restrict__ lens, const float *__restrict__ unit, int Lpad, int K, int T, int t_first, int t_count, int max_len, int W,
    float score_ref, float *__restrict__ scores, float *__restrict__ avg, const float *__restrict__ raw, int force_ref, uint32_t *fix) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const unsigned wi = blockIdx.x / t_count;
    const int t = t_first + (int)(blockIdx.x - wi * t_count);
    const int m = lens[t];
    const int L = m < max_len ? m : max_len;  // m == n == L (checked by the launcher): the window is cut to L frames
    const int B = 2 * W, KP = K | 1;
    float *ys = reinterpret_cast<float *>(smem);   // [L][KP] window frames, then normalised in place
    float *mu = ys + (size_t)L * KP;               // [K]
    float *dm = mu + ((K + 3) & ~3);               // [2L + B + 16][B] band costs by due step
    float *ts = dm + (size_t)(2 * L + B + 16) * B;                // [L][KP] unit template rows (one coalesced read instead of a
                                                   // dependent global load per multiply-add in the cost loop)
    const float *src = mfcc + (first_win + wi) * (size_t)K;
    const float *trow = unit + (size_t)t * Lpad * K;
    for (int i = lane; i < L * K; i += 64) {
        const int f = i / K, k = i - f * K;
        ys[f * KP + k] = (first_win + wi + f) < n_frames_total ? src[(size_t)f * K + k] : 0.f;
        ts[f * KP + k] = trow[i];
    }
    __syncthreads();
    // MfccNormalizer::normalize: sequential column sums (one lane per coefficient), like the register kernels
    for (int k = lane; k < K; k += 64) {
        float sum = 0.f;
        int i = 0;
        for (; i + 8 <= L; i += 8) {  // eight reads in flight, the adds stay in frame order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ys[(i + j) * KP + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[j];
        }
        for (; i < L; ++i) sum += ys[i * KP + k];
        mu[k] = sum / (float)L;
    }
    __syncthreads();
    // the norm-range test of the batch kernels (dtw_band_kernel), healed in place: when a frame of the window leaves
    // kDtwNormLo..kDtwFixLimit (or the template set has such a row: force_ref) this wave forms its costs as the reference does
    // (comparator.rs:28-48) from the rows as given and the centred frames -- the one-wave-per-DTW form has no second pass
    float chk = 0.f;
    for (int f = lane; f < L; f += 64) {
        float bb = 0.f;
        for (int k = 0; k < K; ++k) { const float y = ys[f * KP + k] - mu[k]; bb = fmaf(y, y, bb); }
        chk = fmaxf(fmaxf(chk, bb > 0.f ? rsqrtf(bb) : 0.f), bb);
    }
    const bool ref_cell = force_ref || __any(chk > kDtwFixLimit);
    if (ref_cell && lane == 0) atomicAdd(dtw_fix_stats(fix), 1ull);
    for (int f = lane; f < L; f += 64) {
        float bb = 0.f;
        for (int k = 0; k < K; ++k) { const float y = ys[f * KP + k] - mu[k]; bb = fmaf(y, y, bb); }
        const float inv = ref_cell ? 1.f : (bb > 0.f ? rsqrtf(bb) : 0.f);
        for (int k = 0; k < K; ++k) ys[f * KP + k] = (ys[f * KP + k] - mu[k]) * inv;
    }
    if (ref_cell) {
        const float *rrow = raw + (size_t)t * Lpad * K;
        for (int i = lane; i < L * K; i += 64) { const int f = i / K; ts[f * KP + (i - f * K)] = rrow[i]; }
    }
    __syncthreads();
    // band costs d[r][q] = 1 - a_r . y_c, c = r - W + q, stored by the step they are due at: dm[(2r + q)*B + q];
    // cells outside 1 <= c <= n are never part of a path (+inf)
    const int steps = 2 * (L - 1) + B;  // due steps run from 2 to steps - 1
    for (int i = lane; i < (L - 1) * B; i += 64) {
        const int r = 1 + i / B, q = i - (r - 1) * B, c = r - W + q;
        float d = RP_INF;
        if (c >= 1 && c <= L) {
            const float *a = ts + (r - 1) * KP, *y = ys + (c - 1) * KP;
            if (!ref_cell) {
                d = 1.f;
                for (int k = 0; k < K; ++k) d = fmaf(-a[k], y[k], d);
            } else {
                float dot_ab = 0.f, dot_a = 0.f, dot_b = 0.f;
                for (int k = 0; k < K; ++k) { dot_ab += a[k] * y[k]; dot_a += a[k] * a[k]; dot_b += y[k] * y[k]; }
                const float magnitude = sqrtf(dot_a * dot_b);
                d = 1.f - (magnitude == 0.f ? 0.f : dot_ab / magnitude);
            }
        }
        dm[(2 * r + q) * B + q] = d;
    }
    __syncthreads();
    if (lane >= 16) return;  // the band lives in one DPP row
    const int q = lane < B ? lane : B - 1;
    const bool band_lane = lane < B;
    float cur = lane == W ? 0.f : RP_INF;  // row 0: D[0][0] = 0 sits at band offset W
    const int first_due = q + 2, last_due = 2 * (L - 1) + q;
    const float *dq = dm + q;
    for (int tau0 = 2; tau0 < steps; tau0 += 16) {
        float dreg[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) dreg[j] = dq[(tau0 + j) * B];  // 16 steps' worth of costs, one wait
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int tau = tau0 + j;
            const int ci = __float_as_int(cur);
            const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(RP_INF), ci, 0x111, 0xf, 0xf, false));  // row_shr:1
            // row_shl:1; lane B (never due) stays +inf, so the last band lane's upper neighbour is out of band by itself
            const float up = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(RP_INF), ci, 0x101, 0xf, 0xf, false));
            const bool due = band_lane && ((tau - q) & 1) == 0 && tau >= first_due && tau <= last_due;
            const float nv = dreg[j] + fminf(fminf(up, left), cur);
            cur = due ? nv : cur;
        }
    }
    if (lane == W + 1) {  // D[m-1][n]
        const float nc = cur / (float)(L + L);
        const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
        const size_t row = (size_t)wi;  // S == 1
        if (t < T) scores[row * T + t] = sc;
        else avg[row] = sc;
    }
}

int dtw_register_tile(int K, int band) {
    if (K == 5 && band >= 3 && band <= 6) return 8;
    if ((K == 13 || K == 16) && band >= 3 && band <= 6) return 2;
    return 0;
}

template <int W>
static hipError_t launch_dtw_k5(hipStream_t st, const TemplatesDev &t, int n1, const float *mfcc, size_t S, size_t frame_pitch,
                                size_t tiles, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref,
                                float *scores, float *avg, bool few, GateList gl = GateList{}) {
    hipError_t e;
    // templates whose length occurs once or twice: the matrix-core kernel for unequal lengths (rp_dtw_ragged.hip) when every window of
    // the call is scored from LDS-staged tiles; the averaged template (its own output array) keeps its register launch.  The windows that
    // kernel lists (a frame it cannot resolve within the parity gate: digital silence behind speech, wild scales) are scored again by the
    // register kernels in list mode -- two launches that leave at once when nothing is listed -- or, without slack behind the frame
    // array, by dtw_ref_kernel
    if (!few && !gl.list && !gl.count && t.rag_count > 0 && W <= 5 && gl.wk_all.rag_prep && gl.wk_all.rag_streams >= S &&
        dtw_ragged_supported(t, W, n_win, score_ref)) {
        const int n_avg = (t.has_avg && n1 == t.class_count[3] && n1 > 0) ? 1 : 0;
        if (n_avg)
            if ((e = launch_dtw_single_chunks<5, W>(st, t, t.class_first[3] + t.class_count[3] - 1, 1, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl)) != hipSuccess) return e;
        // (every ragged chunk lists on its own: a window may be listed once per chunk -- the list holds rows x chunks entries)
        const bool list_rows = gl.padded && gl.wk_all.rag_list && gl.wk_all.rag_rows >= S * n_win * (size_t)t.rag_count && out_win_pitch == n_win;
        if ((e = launch_dtw_ragged(st, gl.work(), t, W, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref, scores, gl.abandon_nc, list_rows)) != hipSuccess) return e;
        if (list_rows) {
            GateList g2 = gl;
            g2.list = gl.wk_all.rag_list + 1; g2.count = gl.wk_all.rag_list; g2.dense_min = 0; g2.fuse = nullptr;
            g2.list_mult = (uint32_t)t.rag_count;   // every ragged chunk lists on its own (round-5 advice: a grid for S x n_win entries dropped the tail)
            if ((e = launch_dtw_single_chunks<5, W>(st, t, t.class_first[3], t.class_count[3] - (t.has_avg ? 1 : 0), mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref, scores, avg, false, g2)) != hipSuccess) return e;
            if ((e = launch_dtw_class<5, W, 2>(st, t, t.class_first[0], t.class_count[0], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, false, g2)) != hipSuccess) return e;
        }
    } else {
        // n1: single-template chunks to score (class 3; the averaged template is its last chunk)
        if ((e = launch_dtw_single_chunks<5, W>(st, t, t.class_first[3], n1, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl)) != hipSuccess) return e;
        if ((e = launch_dtw_class<5, W, 2>(st, t, t.class_first[0], t.class_count[0], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl)) != hipSuccess) return e;
    }
    // chunks of 3..8 templates: the matrix-core kernel (rp_dtw_mfma.hip) in every mode (LDS-staged, frames from global memory for
    // live-stream batches and the gate's list, early abandon): 5..8 templates at band 3..5 with eight template slots per wave, 3..4 at
    // band 5 with four.
    {
        const bool from_global = few || gl.list != nullptr;
        if (t.class_count[1] > 0 && dtw_mfma_supported(t, W, n_win, from_global, 4, score_ref)) {
            if ((e = launch_dtw_mfma(st, gl.work(), t, W, 4, t.class_first[1], t.class_count[1], mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref,
                                     scores, avg, from_global, gl.list, gl.count, gl.dense_min, gl.abandon_nc, gl.fuse)) != hipSuccess) return e;
        } else if ((e = launch_dtw_class<5, W, 4>(st, t, t.class_first[1], t.class_count[1], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl)) != hipSuccess) return e;
        if (t.class_count[2] > 0 && dtw_mfma_supported(t, W, n_win, from_global, 8, score_ref)) {
            // several chunks of one length, every window of a long batch scored: the workgroups that share a column's B operand among four
            // (two) chunks (rp_dtw_mfma_group.hip: same bits); the chunks outside a group, and every other mode, keep dtw_mfma_kernel
            if (!from_global && !gl.count && !gl.fuse && !(gl.abandon_nc < RP_INF) && dtw_mfma_group_supported(t, W, n_win, S, score_ref)) {
                if ((e = launch_dtw_mfma_group(st, gl.work(), t, W, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref, scores)) != hipSuccess) return e;
                for (int r = 0; r < t.rest_runs; ++r)
                    if ((e = launch_dtw_mfma(st, gl.work(), t, W, 8, t.rest_first[r], t.rest_count[r], mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref,
                                             scores, avg, false, nullptr, nullptr, 0, gl.abandon_nc, nullptr)) != hipSuccess) return e;
                return hipSuccess;
            }
            return launch_dtw_mfma(st, gl.work(), t, W, 8, t.class_first[2], t.class_count[2], mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, score_ref,
                                   scores, avg, from_global, gl.list, gl.count, gl.dense_min, gl.abandon_nc, gl.fuse);
        }
    }
    // Small batches: tc-8 waves run two per SIMD; a launch that fills those slots 2.x times leaves the chip mostly idle in
    // its last round.  The same templates as tc-4 half chunks are twice as many waves of 0.83 of the length (measured at C2), three per SIMD
    // (146 VGPRs).  Taken when the modelled makespan is shorter; large batches (>= 3 rounds of tc-8 waves) never are.
    if (t.class_count[2] > 0 && t.split_count == 2 * t.class_count[2] && std::getenv("RP_DTW_NO_SPLIT") == nullptr) {
        const double waves8 = (double)((S * n_win + kDtwWin - 1) / kDtwWin) * t.class_count[2];
        const double slots = 4.0 * device_cu_count();
        const double cost8 = std::ceil(waves8 / (2.0 * slots)), cost4 = 0.85 * std::ceil(2.0 * waves8 / (3.0 * slots));
        if (waves8 < 3.0 * 2.0 * slots && cost4 < cost8)
            return launch_dtw_class<5, W, 4>(st, t, t.split_first, t.split_count, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
    }
    return launch_dtw_class<5, W, 8>(st, t, t.class_first[2], t.class_count[2], mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
}

// ---- the averaged-template gate as a skip (wakeword_comp.rs:85-93) -------------------------------------------------
// The reference scores the window against the averaged template first and returns None when that score is below
// avg_threshold: the T sample templates are never compared.  Batched: pass 1 scores EVERY window against the averaged
// template only, gate_compact_kernel lists the rows that pass, pass 3 runs the sample templates on the listed rows
// (one lane per listed window, frames read from global memory).  Rows that are not listed keep whatever `scores` held.
// One wave lists 16 x 64 consecutive rows with a single atomic (one atomic per wave-row of 64 would serialise on the
// counter: ~300 k atomics at C3); the order inside a wave's block is preserved, so neighbouring windows stay neighbours.
__global__ __launch_bounds__(256) void gate_compact_kernel(const float *__restrict__ avg, size_t rows, float avg_threshold,
                                                           uint32_t *__restrict__ list, uint32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t r0 = wave * 1024;
    if (r0 >= rows) return;
    unsigned long long m[16];
    unsigned total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t r = r0 + (size_t)i * 64 + lane;
        const bool pass = r < rows && !(avg[r] < avg_threshold);  // `avg_score < avg_threshold -> None`
        m[i] = __ballot(pass);
        total += (unsigned)__popcll(m[i]);
    }
    if (total == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, total);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if ((m[i] >> lane) & 1ull) list[base + (unsigned)__popcll(m[i] & ((1ull << lane) - 1ull))] = (uint32_t)(r0 + (size_t)i * 64 + lane);
        base += (unsigned)__popcll(m[i]);
    }
}

bool dtw_gate_supported(const TemplatesDev &t, int band, size_t rows) {
    return t.has_avg && !t.ref_only && dtw_register_tile(t.K, band) > 0 && t.max_diff == 0 && t.chunks && rows > 0 && rows < 0xffffffffULL &&
           (size_t)(2 * kDtwWin + 2 * (t.max_len + 8)) * (size_t)(t.K | 1) * sizeof(float) <= 160 * 1024;
}

template <int W>
static hipError_t gated_k5(hipStream_t st, const DtwWork &wk, int band, const TemplatesDev &t, int avg_chunk, const float *mfcc, size_t S, size_t frame_pitch,
                           size_t first_win, size_t n_win, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                           uint32_t *count, bool few, float abandon_nc) {
    const size_t rows = S * n_win, tiles = (n_win + kDtwWin - 1) / kDtwWin;
    // pass 1: the averaged template over every window (and, before the gate looks at them, the reference-shaped rescoring of the
    // windows whose frames left the norm range: dtw_ref_kernel)
    GateList g1;
    g1.fix = wk.fix; g1.sched = wk.sched; g1.ran = wk.ran;
    hipError_t e = launch_dtw_single_chunks<5, W>(st, t, avg_chunk, 1, mfcc, S, frame_pitch, first_win, n_win, n_win, score_ref, scores, avg, few, g1);
    if (e != hipSuccess) return e;
    if ((e = launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, n_win, band, score_ref, scores, avg, false, t.T, 1)) != hipSuccess) return e;
    // pass 2: list the rows whose avg_score is not below the threshold
    const size_t waves = (rows + 1023) / 1024, blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(gate_compact_kernel, dim3((unsigned)blocks), dim3(256), 0, st, avg, rows, avg_threshold, list, count);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // pass 3: the sample templates on the listed rows -- or, when (nearly) every row is listed, on all rows through the
    // ordinary staged launch (GateList; with few windows per stream both forms read global memory: list mode only)
    GateList gl;
    gl.list = list; gl.count = count; gl.abandon_nc = abandon_nc; gl.fix = wk.fix; gl.sched = wk.sched; gl.ran = wk.ran;
    gl.dense_min = few ? 0u : (uint32_t)(rows - rows / 10);
    e = launch_dtw_k5<W>(st, t, t.class_count[3] - 1, mfcc, S, frame_pitch, tiles, first_win, n_win, n_win, score_ref, scores, avg, false, gl);
    if (e != hipSuccess) return e;
    if (!few) {
        gl.list = nullptr;
        if ((e = launch_dtw_k5<W>(st, t, t.class_count[3] - 1, mfcc, S, frame_pitch, tiles, first_win, n_win, n_win, score_ref, scores, avg, false, gl)) != hipSuccess) return e;
    }
    return launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, n_win, band, score_ref, scores, avg, false, 0, t.T);
}

template <int K, int W>
static hipError_t gated_wide(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int avg_chunk, const float *mfcc, size_t S, size_t frame_pitch,
                             size_t first_win, size_t n_win, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                             uint32_t *count, bool few, float abandon_nc) {
    const size_t rows = S * n_win, tiles = (n_win + kDtwWin - 1) / kDtwWin;
    GateList g1;
    g1.fix = wk.fix; g1.sched = wk.sched; g1.ran = wk.ran;
    hipError_t e = launch_dtw_wide<K, W, 1>(st, t, 3, 1, mfcc, S, frame_pitch, tiles, first_win, n_win, n_win, score_ref, scores, avg, few, g1, avg_chunk);
    if (e != hipSuccess) return e;
    if ((e = launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, n_win, W, score_ref, scores, avg, false, t.T, 1)) != hipSuccess) return e;
    const size_t waves = (rows + 1023) / 1024, blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(gate_compact_kernel, dim3((unsigned)blocks), dim3(256), 0, st, avg, rows, avg_threshold, list, count);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    GateList gl;
    gl.list = list; gl.count = count; gl.abandon_nc = abandon_nc; gl.fix = wk.fix; gl.sched = wk.sched; gl.ran = wk.ran;
    gl.dense_min = few ? 0u : (uint32_t)(rows - rows / 10);
    e = launch_dtw_wide_all<K, W>(st, t, t.class_count[3] - 1, mfcc, S, frame_pitch, tiles, first_win, n_win, n_win, score_ref, scores, avg, false, gl, true);
    if (e != hipSuccess) return e;
    if (!few) {
        gl.list = nullptr;
        if ((e = launch_dtw_wide_all<K, W>(st, t, t.class_count[3] - 1, mfcc, S, frame_pitch, tiles, first_win, n_win, n_win, score_ref, scores, avg, false, gl, true)) != hipSuccess) return e;
    }
    return launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, n_win, W, score_ref, scores, avg, false, 0, t.T);
}

// first_win / few_windows as in launch_dtw (live-stream batches score the few newest windows of every stream: then pass 1
// also reads its frames from global memory); scores / avg rows have pitch n_win.
static hipError_t launch_dtw_gated_impl(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                        size_t n_win, int band, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                                        uint32_t *count, bool few_windows, float abandon_nc);
hipError_t launch_dtw_gated(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                            size_t n_win, int band, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                            uint32_t *count, bool few_windows, float abandon_nc) {
    if (!dtw_gate_supported(t, band, S * n_win)) return hipErrorNotSupported;
    const hipError_t e = launch_dtw_gated_impl(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, band, score_ref, avg_threshold, scores, avg, list, count,
                                               few_windows, abandon_nc);
    return e == hipSuccess ? e : dtw_abort(st, wk, e);
}
static hipError_t launch_dtw_gated_impl(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                        size_t n_win, int band, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                                        uint32_t *count, bool few_windows, float abandon_nc) {
    const size_t rows = S * n_win;
    if (!dtw_gate_supported(t, band, rows)) return hipErrorNotSupported;
    // (as in launch_dtw: one stream alone is scored like a batch when the matrix-core kernel serves its templates)
    const bool mfma_batch = few_windows && ((t.K == 5 && ((t.class_count[2] > 0 && dtw_mfma_supported(t, band, n_win, true, 8, score_ref)) ||
                                                          (t.class_count[1] > 0 && dtw_mfma_supported(t, band, n_win, true, 4, score_ref)))) ||
                                            dtw_mfma_wide_supported(t, band, score_ref) || dtw_mfma_wide3_supported(t, band));
    const bool few = few_windows && (S > 1 || mfma_batch) && n_win < (size_t)kDtwWin;
    const int avg_chunk = t.class_first[3] + t.class_count[3] - 1;  // the averaged template: last of the single-template chunks
    hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (t.K == 5) {
        switch (band) {
        case 3: return gated_k5<3>(st, wk, band, t, avg_chunk, mfcc, S, frame_pitch, first_win, n_win, score_ref, avg_threshold, scores, avg, list, count, few, abandon_nc);
        case 4: return gated_k5<4>(st, wk, band, t, avg_chunk, mfcc, S, frame_pitch, first_win, n_win, score_ref, avg_threshold, scores, avg, list, count, few, abandon_nc);
        case 5: return gated_k5<5>(st, wk, band, t, avg_chunk, mfcc, S, frame_pitch, first_win, n_win, score_ref, avg_threshold, scores, avg, list, count, few, abandon_nc);
        default: return gated_k5<6>(st, wk, band, t, avg_chunk, mfcc, S, frame_pitch, first_win, n_win, score_ref, avg_threshold, scores, avg, list, count, few, abandon_nc);
        }
    }
#define RP_WIDE_CALL(KK, WW) gated_wide<KK, WW>(st, wk, t, avg_chunk, mfcc, S, frame_pitch, first_win, n_win, score_ref, avg_threshold, scores, avg, list, count, few, abandon_nc)
    RP_WIDE_DISPATCH(RP_WIDE_CALL);
#undef RP_WIDE_CALL
}

// Normalised-cost bound above which a DTW cannot reach `threshold` any more: score = 1 / (1 + exp((nc - ref) / ref)) > thr
// <=> nc < ref * (1 + ln(1/thr - 1)) (comparator.rs:18-26), with a margin for the rounding of the running costs.
float dtw_abandon_nc(float threshold, float score_ref) {
    if (!(score_ref > 0.f) || !(threshold > 0.f)) return __builtin_inff();  // everything can fire (or the formula is degenerate): never abandon
    if (threshold >= 1.f) return 0.f;                                         // a score is < 1: nothing can fire
    const float nc = score_ref * (1.f + logf(1.f / threshold - 1.f));
    return nc > 0.f ? nc * 1.001f + 1e-4f : nc + 1e-4f;
}

static hipError_t launch_dtw_fast(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                                  size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, int with_avg,
                                  float *scores, float *avg, bool padded_rows, float abandon_nc, DtwFusedAgg *fuse, bool *self_healing);


hipError_t launch_dtw(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                      size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, int with_avg,
                      float *scores, float *avg, bool padded_rows, float abandon_nc, DtwFusedAgg *fuse) {
    if (fuse) fuse->done = false;
    if (S == 0 || n_win == 0) return hipSuccess;
    if (!wk.fix || !wk.sched) return hipErrorInvalidValue;
    if (t.n_chunks_total > kDtwSchedChunks) return hipErrorInvalidValue;
    const int Ttot = t.T + ((with_avg && t.has_avg) ? 1 : 0);
    bool self_healing = false;
    if (!t.ref_only || (S == 1 && n_win <= 8)) {   // (a handful of windows of one stream: dtw_single_kernel takes force_ref itself)
        if (hipError_t e = launch_dtw_fast(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, with_avg, scores, avg,
                                           padded_rows, abandon_nc, fuse, &self_healing); e != hipSuccess) return dtw_abort(st, wk, e);
        if (self_healing) return hipSuccess;
    }
    // the windows the fast kernels listed (a frame outside the norm range), or -- a template set with such a row -- every window
    const hipError_t e = launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, scores, avg, t.ref_only != 0, 0, Ttot,
                                        (fuse && fuse->done) ? fuse : nullptr);
    return e == hipSuccess ? e : dtw_abort(st, wk, e);
}

static hipError_t launch_dtw_fast(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                                  size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, int with_avg,
                                  float *scores, float *avg, bool padded_rows, float abandon_nc, DtwFusedAgg *fuse, bool *self_healing) {
    GateList gl;
    gl.abandon_nc = abandon_nc; gl.fix = wk.fix; gl.sched = wk.sched; gl.ran = wk.ran; gl.wk_all = wk; gl.padded = padded_rows;
    // many streams with few windows each (streaming batches): cross-stream waves reading frames from global memory;
    // needs `padded_rows` (slack after the last stream's frames for the never-used out-of-band columns)
    // (one stream alone is a batch too when the matrix-core kernel serves its templates: a stream's bits must not depend on the
    // batch it is scored in, live or offline -- only the single-stream mirror, which never passes padded_rows, keeps dtw_single_kernel)
    const bool mfma_batch = padded_rows && ((t.K == 5 && ((t.class_count[2] > 0 && dtw_mfma_supported(t, band, n_win, true, 8, score_ref)) ||
                                                           (t.class_count[1] > 0 && dtw_mfma_supported(t, band, n_win, true, 4, score_ref)))) ||
                                            dtw_mfma_wide_supported(t, band, score_ref) || dtw_mfma_wide3_supported(t, band));
    const bool few = padded_rows && (S > 1 || mfma_batch) && n_win < (size_t)kDtwWin;
    const bool do_avg = with_avg && t.has_avg;
    const int Ttot = t.T + (do_avg ? 1 : 0);
    // a handful of windows of one stream (the single-stream API): one wave per DTW, band lanes on anti-diagonals
    if (S == 1 && n_win <= 8 && !mfma_batch && t.max_diff == 0 && band >= 1 && 2 * band <= 16) {
        const int KP = t.K | 1;
        const size_t lds = (2 * (size_t)t.max_len * KP + ((t.K + 3) & ~3) + (size_t)(2 * t.max_len + 2 * band + 16) * 2 * band) * sizeof(float);
        if (lds <= 64 * 1024) {
            hipLaunchKernelGGL(dtw_single_kernel, dim3((unsigned)(n_win * Ttot)), dim3(64), lds, st, mfcc, frame_pitch, first_win,
                               (unsigned)n_win, out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, 0, Ttot, t.max_len, band, score_ref,
                               scores, avg, t.raw, t.ref_only, wk.fix);
            *self_healing = true;
            dtw_mark(wk, kDtwRanSingle);
            return hipGetLastError();
        }
    }
    if (t.ref_only) return hipSuccess;  // launch_dtw scores every window with dtw_ref_kernel
    const size_t tiles = (n_win + kDtwWin - 1) / kDtwWin;
    // the register kernels assume m == n (no template longer than the window)
    // (a register kernel stages 64..128 windows + two template lengths of frames in LDS: templates beyond ~4 000 frames at
    // mfcc_size 5 -- 40 s -- do not fit the CU's 160 KB; the generic kernel stages one length and takes them up to ~8 000)
    const size_t reg_lds = (size_t)(2 * kDtwWin + 2 * (t.max_len + 8)) * (size_t)(t.K | 1) * sizeof(float);
    if (dtw_register_tile(t.K, band) > 0 && t.max_diff == 0 && t.chunks && (few || reg_lds <= 160 * 1024)) {
        const int n2 = t.class_count[3] - ((t.has_avg && !do_avg) ? 1 : 0);  // single-template chunks to score
        if (t.K == 5) {
            // ScoreMode::Max inside the matrix-core kernel: one chunk of 3..8 templates is all there is to score
            if (fuse && fuse->agg && n2 == 0 && t.class_count[0] == 0 && t.class_count[1] + t.class_count[2] == 1 && band >= 3 && band <= 5 &&
                std::getenv("RP_DTW_NO_FUSED_MAX") == nullptr &&
                dtw_mfma_supported(t, band, n_win, few, t.class_count[2] == 1 ? 8 : 4, score_ref)) {
                gl.fuse = fuse;
                fuse->done = true;
            }
            switch (band) {
            case 3: return launch_dtw_k5<3>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
            case 4: return launch_dtw_k5<4>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
            case 5: return launch_dtw_k5<5>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
            default: return launch_dtw_k5<6>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl);
            }
        }
#define RP_WIDE_CALL(KK, WW) launch_dtw_wide_all<KK, WW>(st, t, n2, mfcc, S, frame_pitch, tiles, first_win, n_win, out_win_pitch, score_ref, scores, avg, few, gl, padded_rows)
        RP_WIDE_DISPATCH(RP_WIDE_CALL);
#undef RP_WIDE_CALL
    }
    const size_t blocks = tiles * (size_t)Ttot * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const int KP = t.K | 1;
    // the band is widened to |m-n| inside the kernel; size for the worst case over templates
    const int Wmax = band > t.max_diff ? band : t.max_diff;
    const size_t lds = ((size_t)(64 + t.max_len - 1) * KP + (size_t)t.K * 64 + (size_t)(2 * Wmax + 1) * 64) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorMemoryAllocation;  // reported as "template too long" by the callers' hip_ok text
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_generic_kernel), 160 * 1024); e != hipSuccess) return e;
    dtw_mark(wk, kDtwRanGeneric);
    hipLaunchKernelGGL(dtw_generic_kernel, dim3((unsigned)blocks), dim3(64), lds, st, mfcc, frame_pitch, frame_pitch,
                       (unsigned)tiles, first_win, n_win, out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, 0, Ttot,
                       t.max_len, band, score_ref, scores, avg, static_cast<const float *>(nullptr), 0.f, wk.fix);
    return hipGetLastError();
}

// true when launch_dtw serves this template set with dtw_generic_kernel (any mfcc_size / band / an averaged template longer
// than the window)
bool dtw_uses_generic(const TemplatesDev &t, int band, size_t S, size_t n_win) {
    if (S == 1 && n_win <= 8 && t.max_diff == 0 && band >= 1 && 2 * band <= 16) {   // dtw_single_kernel, if its LDS fits
        const size_t lds1 = (2 * (size_t)t.max_len * (t.K | 1) + ((t.K + 3) & ~3) + (size_t)(2 * t.max_len + 2 * band + 16) * 2 * band) * sizeof(float);
        if (lds1 <= 64 * 1024) return false;
    }
    if (t.ref_only) return true;   // dtw_ref_kernel: launch_dtw_generic_gated is the gated form
    const size_t reg_lds = (size_t)(2 * kDtwWin + 2 * (t.max_len + 8)) * (size_t)(t.K | 1) * sizeof(float);
    return !(dtw_register_tile(t.K, band) > 0 && t.max_diff == 0 && t.chunks && reg_lds <= 160 * 1024);
}

// The averaged-template gate behind the generic kernel: pass 1 scores every window against the averaged template (-> avg),
// pass 2 the sample templates, each wave leaving at once when none of its 64 windows passed (scores of such rows are not
// written; the aggregate pass gives them 0).
static hipError_t launch_dtw_generic_gated_impl(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                                                size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, float avg_threshold,
                                                float *scores, float *avg);
hipError_t launch_dtw_generic_gated(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                    size_t n_win, size_t out_win_pitch, int band, float score_ref, float avg_threshold, float *scores,
                                    float *avg) {
    if (S == 0 || n_win == 0) return hipSuccess;
    if (!t.has_avg || !avg) return hipErrorInvalidValue;
    const hipError_t e = launch_dtw_generic_gated_impl(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, avg_threshold, scores, avg);
    return e == hipSuccess ? e : dtw_abort(st, wk, e);
}
static hipError_t launch_dtw_generic_gated_impl(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                                                size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, float avg_threshold,
                                                float *scores, float *avg) {
    if (t.ref_only) {  // a template row outside the norm range: every window reference-shaped, no skipping (the aggregate pass writes 0 for rejected rows)
        if (hipError_t e = launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, scores, avg, true, t.T, 1); e != hipSuccess) return e;
        return launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, scores, avg, true, 0, t.T);
    }
    const size_t tiles = (n_win + kDtwWin - 1) / kDtwWin;
    if (tiles * (size_t)t.T * S > 0x7fffffffULL) return hipErrorInvalidValue;
    const int KP = t.K | 1;
    const int Wmax = band > t.max_diff ? band : t.max_diff;
    const size_t lds = ((size_t)(64 + t.max_len - 1) * KP + (size_t)t.K * 64 + (size_t)(2 * Wmax + 1) * 64) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorMemoryAllocation;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_generic_kernel), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL(dtw_generic_kernel, dim3((unsigned)(tiles * S)), dim3(64), lds, st, mfcc, frame_pitch, frame_pitch, (unsigned)tiles,
                       first_win, n_win, out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, t.T, 1, t.max_len, band, score_ref, scores, avg,
                       static_cast<const float *>(nullptr), 0.f, wk.fix);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    // the gate must see reference-shaped avg scores: rescoring of the listed windows first
    if (hipError_t e = launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, scores, avg, false, t.T, 1); e != hipSuccess) return e;
    hipLaunchKernelGGL(dtw_generic_kernel, dim3((unsigned)(tiles * (size_t)t.T * S)), dim3(64), lds, st, mfcc, frame_pitch, frame_pitch,
                       (unsigned)tiles, first_win, n_win, out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, 0, t.T, t.max_len, band, score_ref,
                       scores, avg, static_cast<const float *>(avg), avg_threshold, wk.fix);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    return launch_dtw_ref(st, wk, t, mfcc, S, frame_pitch, first_win, n_win, out_win_pitch, band, score_ref, scores, avg, false, 0, t.T);
}

// A handful of windows of ONE stream (the single-stream API), templates t_first .. t_first + t_count - 1 only (index T = the
// averaged template): the caller scores the averaged template first and the sample templates only when a window passed
// the gate.  hipErrorNotSupported when dtw_single_kernel does not take this set (the caller then scores everything).
hipError_t launch_dtw_single_part(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t frame_pitch, size_t first_win, size_t n_win,
                                  size_t out_win_pitch, int band, float score_ref, int t_first, int t_count, float *scores, float *avg) {
    if (n_win == 0 || t_count <= 0) return hipSuccess;
    if (!(n_win <= 8 && t.max_diff == 0 && band >= 1 && 2 * band <= 16) || t_first + t_count > t.T + (t.has_avg ? 1 : 0)) return hipErrorNotSupported;
    const int KP = t.K | 1;
    const size_t lds = (2 * (size_t)t.max_len * KP + ((t.K + 3) & ~3) + (size_t)(2 * t.max_len + 2 * band + 16) * 2 * band) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorNotSupported;
    hipLaunchKernelGGL(dtw_single_kernel, dim3((unsigned)(n_win * t_count)), dim3(64), lds, st, mfcc, frame_pitch, first_win, (unsigned)n_win,
                       out_win_pitch, t.lens, t.unit, t.Lpad, t.K, t.T, t_first, t_count, t.max_len, band, score_ref, scores, avg, t.raw, t.ref_only, wk.fix);
    return hipGetLastError();
}

// -------------------------------------------------------------------- aggregate
// src/wakewords/comp/wakeword_comp.rs:38-49 (get_percentile) and :108-139
__device__ inline float percentile_sorted(const float *v, int n, float percentile) {
    float index = percentile / 100.0f * (float)(n - 1);
    float fl = floorf(index);
    if (fl == index) return v[(int)index];
    int i = (int)fl;
    float d = index - fl;
    return v[i] * (1.0f - d) + v[i + 1] * d;
}

constexpr int kAggMaxT = 256;

// Max / Average: no sort buffer, so no scratch memory to set up (the single-stream path launches this for 3 rows).
// A workgroup owns 64 consecutive rows: the [64][T] block of scores is one contiguous range, copied to LDS with
// coalesced loads (row pitch T+1: conflict-free), then lane r walks row r in template order.
// What a row's aggregate means downstream: a window the averaged-template gate rejected was never compared with the
// sample templates (its `scores` row holds nothing), so its aggregate is written as 0 here instead of leaving that to the
// scan's own avg test; and the first window of a stream that can fire (agg > threshold, gate passed) raises the stream's
// `hot` flag, so that scan_kernel does not have to sweep the rows of quiet streams to learn that nothing can happen
// (src/detector.rs:398-430: without such a window no partial detection ever exists).
__device__ __forceinline__ void agg_store(float *__restrict__ agg, size_t row, float a, const AggExtra &x) {
    const bool gated = x.gate_avg && x.gate_avg[row] < x.gate_threshold;
    if (gated) a = 0.f;
    agg[row] = a;
    if (x.hot && !gated && a > x.threshold) x.hot[row / x.n_win] = 1u;   // every writer stores the same value
}

__global__ __launch_bounds__(64) void aggregate_kernel(const float *__restrict__ scores, size_t n_rows, int T, int mode,
                                                       float *__restrict__ agg, AggExtra x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [64][T + 1]
    const size_t row0 = (size_t)blockIdx.x * 64;
    const size_t nr = n_rows - row0 < 64 ? n_rows - row0 : 64;
    const float *src = scores + row0 * T;
    const int total = (int)nr * T, P = T + 1;
    for (int i = threadIdx.x; i < total; i += 64) {
        const int r = i / T, t = i - r * T;
        tile[r * P + t] = src[i];
    }
    __syncthreads();
    const int r = threadIdx.x;
    if ((size_t)r >= nr) return;
    const float *v = tile + r * P;
    if (mode == 1) {  // Max
        float m = v[0];
        for (int i = 1; i < T; ++i) m = fmaxf(m, v[i]);
        agg_store(agg, row0 + r, m, x);
    } else {  // Average: sequential sum in template order
        float sum = 0.f;
        for (int i = 0; i < T; ++i) sum += v[i];
        agg_store(agg, row0 + r, sum / (float)T, x);
    }
}

// Median / percentiles: sort ascending, then the reference's f32 interpolation (wakeword_comp.rs:38-49).
// T <= 64: a workgroup copies its 64 rows [64][T] (one contiguous block) to LDS with coalesced loads, lane r takes row r
// into NT registers (padded with +inf) and sorts them with a compile-time bitonic network -- no scratch memory, no
// data-dependent loop; the percentile position depends only on T and the mode, so it is wave-uniform.
template <int NT> __device__ __forceinline__ void bitonic_sort(float (&v)[NT]) {
#pragma unroll
    for (int k = 2; k <= NT; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const float a = v[i], b = v[l];
                    const float lo = fminf(a, b), hi = fmaxf(a, b);
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
}

__device__ __forceinline__ float percentile_of_mode(int mode) {
    switch (mode) {
    case 3: return 25.f;
    case 5: return 75.f;
    case 6: return 80.f;
    case 7: return 90.f;
    case 8: return 95.f;
    default: return 50.f;  // Median, P50
    }
}

template <int NT>
__global__ __launch_bounds__(64) void aggregate_sorted_reg_kernel(const float *__restrict__ scores, size_t n_rows, int T, int mode,
                                                                  float *__restrict__ agg, AggExtra x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [64][T + 1]
    const size_t row0 = (size_t)blockIdx.x * 64;
    const size_t nr = n_rows - row0 < 64 ? n_rows - row0 : 64;
    const float *src = scores + row0 * T;
    const int total = (int)nr * T, P = T + 1;
    for (int i = threadIdx.x; i < total; i += 64) {
        const int r = i / T, t = i - r * T;
        tile[r * P + t] = src[i];
    }
    __syncthreads();
    const int r = threadIdx.x;
    if ((size_t)r >= nr) return;
    float v[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) v[i] = i < T ? tile[r * P + i] : RP_INF;
    bitonic_sort<NT>(v);
    // get_percentile: index = p/100 * (T-1) in f32; an exact integer takes the element, else linear interpolation
    const float index = percentile_of_mode(mode) / 100.0f * (float)(T - 1);
    const float fl = floorf(index);
    const int i0 = (int)fl;  // wave-uniform
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        lo = i == i0 ? v[i] : lo;
        hi = i == i0 + 1 ? v[i] : hi;
    }
    const float d = index - fl;
    agg_store(agg, row0 + r, fl == index ? lo : lo * (1.0f - d) + hi * d, x);
}

// T > 64: per-lane insertion sort in scratch memory
__global__ __launch_bounds__(64) void aggregate_sorted_kernel(const float *__restrict__ scores, size_t n_rows, int T, int mode,
                                                              float *__restrict__ agg, AggExtra x) {
    size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= n_rows) return;
    const float *v = scores + row * T;
    float tmp[kAggMaxT];
    for (int i = 0; i < T; ++i) {  // insertion sort ascending (total_cmp order for non-NaN scores)
        float x = v[i];
        int j = i - 1;
        while (j >= 0 && tmp[j] > x) { tmp[j + 1] = tmp[j]; --j; }
        tmp[j + 1] = x;
    }
    agg_store(agg, row, percentile_sorted(tmp, T, percentile_of_mode(mode)), x);
}

hipError_t launch_aggregate(hipStream_t st, const float *scores, size_t n_rows, int T, int mode, float *agg, AggExtra x) {
    if (n_rows == 0) return hipSuccess;
    if (T < 1 || T > kAggMaxT) return hipErrorInvalidValue;
    size_t blocks = (n_rows + 63) / 64;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    if (mode == 0 || mode == 1)
        hipLaunchKernelGGL(aggregate_kernel, dim3((unsigned)blocks), dim3(64), (size_t)64 * (T + 1) * sizeof(float), st, scores, n_rows, T, mode, agg, x);
    else if (T <= 64) {
        const size_t lds = (size_t)64 * (T + 1) * sizeof(float);
#define RP_AGG_SORTED(NT) hipLaunchKernelGGL(aggregate_sorted_reg_kernel<NT>, dim3((unsigned)blocks), dim3(64), lds, st, scores, n_rows, T, mode, agg, x)
        if (T <= 2) RP_AGG_SORTED(2);
        else if (T <= 4) RP_AGG_SORTED(4);
        else if (T <= 8) RP_AGG_SORTED(8);
        else if (T <= 16) RP_AGG_SORTED(16);
        else if (T <= 32) RP_AGG_SORTED(32);
        else RP_AGG_SORTED(64);
#undef RP_AGG_SORTED
    } else hipLaunchKernelGGL(aggregate_sorted_kernel, dim3((unsigned)blocks), dim3(64), 0, st, scores, n_rows, T, mode, agg, x);
    return hipGetLastError();
}

}  // namespace rp
