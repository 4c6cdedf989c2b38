// rp_dtw_mfma_wide.hip -- dtw_mfma_wide_kernel: the matrix-core DTW (rp_dtw_mfma.hip) for mfcc_size 13 and 16.
// Same sweep, same recurrence, same lane layout (32 windows x 8 template slots per wave, lane = (window, half), two template pairs per
// lane, 12 circular row slots = 3 MFMA tiles); what changes is the K axis of the product:
//   * lane half h owns CHM = ceil(K / 2) components (half 1 of mfcc_size 13: six components and a zero);
//   * per component the three products x0 a0, x1 a0, x0 a1 of the f16 two-way splits: component pairs (a, b) fill three registers
//     [(x0a, x0b), (x1a, x1b), (x0a, x0b)] against [(a0, a0), (a0, a0), (a1, a1)]; an odd last component two: [(x0, x1), (x0, c)]
//     against [(a0, a0), (a1, c)]; c = 1.0 in half 1 (the constant of 1 - a.x), else 0.  An even count (mfcc_size 16) has no slot left
//     over: its 1 is the C operand of the first k-step (round 5; through round 4 it took a thirteenth register = a fourth k-step = 12
//     MFMAs per column and a 1 KB template row.  The compiler keeps the sixteen ones in registers -- it folds a splat into the
//     instruction's inline constant only for a single use -- which is still 4 registers fewer than the fourth k-step's operands);
//   * 12 (mfcc_size 16) / 11 (mfcc_size 13) registers per half = three k-steps of v_mfma_f32_32x32x16_f16 per tile, chained on one
//     accumulator: 9 MFMAs per column.
// The wide register kernels (dtw_band_wide_kernel: two templates per wave, 13 or 16 FMAs per cell) spend 80 % of their cycles on the
// cost FMAs; here the vector pipe runs the recurrence and the frame work only.
// Frames are always read from global memory (a frame is 52 / 64 bytes: staging 32 + 2 L of them per wave next to an A image of 0.75 / 1 KB
// per template row would not fit the LDS): the caller leaves slack behind the last stream's frames (launch_dtw `padded_rows`).
#include "rp_device.h"

#include <cstdlib>

namespace rp {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWWin = 32, kWSlots = 12, kWTiles = 3;

__device__ __forceinline__ unsigned pkrtz(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
// x0 = rtz_f16(x) as f32: x with the low 13 mantissa bits cleared (a full-rate v_and; below the f16 normal range the two differ by less
// than the f16 subnormal spacing, 6e-8)
__device__ __forceinline__ float x0f(float x) { return __uint_as_float(__float_as_uint(x) & 0xffffe000u); }

template <int W>
__host__ __device__ constexpr int wide_last_use(int u, int g) {
    int last = -1;
    for (int q = 0; q < 2 * W; ++q)
        if (((u + q + kWSlots - W + 2) % kWSlots) / 4 == g) last = q;
    return last;
}

// The order and place of a column's MFMAs: a tile is refilled (KS chained k-steps) once its last cell of the column has read it; issued in
// one burst the second and third k-step wait for the matrix pipe (32 cycles each) with the wave's own vector work queued behind them.
// Here at most one instruction goes out per band cell: tiles in the order they come free, k-steps in order, the rest after the last cell.
struct WideIssue { int tile[12], ks[12], at[12]; };
template <int W, int KS>
__host__ __device__ constexpr WideIssue wide_issue(int u) {
    WideIssue s{};
    int order[kWTiles] = {0, 1, 2}, lu[kWTiles] = {wide_last_use<W>(u, 0), wide_last_use<W>(u, 1), wide_last_use<W>(u, 2)};
    for (int i = 0; i < kWTiles; ++i)
        for (int j = i + 1; j < kWTiles; ++j)
            if (lu[order[j]] < lu[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int n = 0, prev = -1;
    for (int i = 0; i < kWTiles; ++i)
        for (int ks = 0; ks < KS; ++ks) {
            int at = lu[order[i]] > prev + 1 ? lu[order[i]] : prev + 1;
            if (at < 0) at = 0;
            if (at > 2 * W - 1) at = 2 * W - 1;
            s.tile[n] = order[i]; s.ks[n] = ks; s.at[n] = at;
            prev = at; ++n;
        }
    return s;
}
struct WideIssueTable { int at[kWSlots][kWTiles][4]; };   // band cell after which k-step ks of tile g goes out in column phase u
template <int W, int KS>
__host__ __device__ constexpr WideIssueTable wide_issue_table() {
    WideIssueTable t{};
    for (int u = 0; u < kWSlots; ++u) {
        const WideIssue s = wide_issue<W, KS>(u);
        for (int i = 0; i < kWTiles * KS; ++i) t.at[u][s.tile[i]][s.ks[i]] = s.at[i];
    }
    return t;
}

}  // namespace

template <int K, int W, int NW>
__global__ __launch_bounds__(64 * NW, 1) void dtw_mfma_wide_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t total_tiles, unsigned n_chunks, int chunk_base, size_t first_win,
    size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks, const uint4 *__restrict__ aimg, int T, float score_ref,
    float *__restrict__ scores, float *__restrict__ avg, size_t n_streams, const uint32_t *__restrict__ list,
    const uint32_t *__restrict__ count, uint32_t dense_min, float abandon_nc, uint32_t *__restrict__ sched, unsigned static_rounds, uint32_t *__restrict__ fix) {
    constexpr int B = 2 * W, NS = kWSlots, NTILE = kWTiles;
    constexpr int CHM = dtw_mfma_wide_chm(K), NPAIR = CHM / 2, ODD = CHM % 2, KS = dtw_mfma_wide_ksteps(K);
    constexpr int kRowBytes = dtw_mfma_wide_row_bytes(K);
    constexpr WideIssueTable kIssue = wide_issue_table<W, KS>();
    static_assert(B + 2 <= NS, "the band and its two neighbours must fit the 12 row slots");
    size_t total_entries = n_streams * n_win;
    if (list) {
        const uint32_t n_listed = *count;
        if (dense_min && n_listed >= dense_min) return;
        total_entries = n_listed;
        total_tiles = ((size_t)n_listed + kWWin - 1) / kWWin;
    } else if (count && *count < dense_min) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned ci = blockIdx.x % n_chunks;
    const unsigned n_groups = gridDim.x / n_chunks;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const u32x4 *asrc = reinterpret_cast<const u32x4 *>(aimg) + ch->aimg_off;
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < (L + 16) * kRowBytes / 16; i += 64 * NW) adst[i] = asrc[i];
    }
    __syncthreads();
    const int n = lane & 31, h = lane >> 5;
    const int mrow = lane & 31, jj = mrow >> 3, tA = ((mrow >> 2) & 1) * 4 + (mrow & 3);
    const unsigned a_lane = (unsigned)(h * 128 + tA * 16);
    unsigned dl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dl[e] = (unsigned)(((e - jj + NS) % NS) * kRowBytes);
    const int nvalid = h ? K - CHM : CHM;  // components this half really has (mfcc_size 13: 7 and 6)
    const unsigned sel_c = h ? 0x07060100u : 0x0c0c0100u;             // (x0, c) of an odd last component: c = 1.0 (half 1) or 0
    const float abandon_cost = abandon_nc * (float)(L + L);
    bool slot_real[4], slot_avg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { slot_real[e] = 4 * h + e < ch->count; slot_avg[e] = slot_real[e] && ch->tid[4 * h + e] >= T; }

    uint32_t *next_tile = sched + 2 * (chunk_base + ci);
    unsigned round = 0;
    const size_t chunk_waves = (size_t)n_groups * NW;  // waves working on this chunk
    for (;;) {
        // the first static_rounds tiles of a wave are its own index among the chunk's waves (+ a round's worth each time), the following
        // ones come from the counter: 3 072 waves asking one address for a ticket at the same moment queue up behind each other (a
        // launch of two tiles per wave -- a live-stream call -- lost a quarter of its time there); the host keeps the counter for the
        // rounds in which balancing matters (mfma_static_rounds)
        size_t tile;
        if (round < static_rounds) {
            tile = (size_t)round * chunk_waves + (size_t)(blockIdx.x / n_chunks) * NW + (size_t)(tid >> 6);
            ++round;
        } else {
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(next_tile, 1u);
            tile = (size_t)__builtin_amdgcn_readfirstlane(ticket) + (size_t)static_rounds * chunk_waves;
        }
        if (tile >= total_tiles) break;
        size_t f = tile * kWWin + n;
        const bool valid = f < total_entries;
        if (list) f = list[valid ? f : total_entries - 1];
        const size_t s = valid ? f / n_win : 0;
        const int w = valid ? (int)(f - s * n_win) : 0;
        const float *xh = mfcc + (s * frame_pitch + first_win + (size_t)w) * K + h * CHM;  // this half's components of the window's first frame

// this half's components of window frame cc (1-based) -> fl[]
#define RP_LOADF(cc)                                                                                                          \
    do {                                                                                                                      \
        const float *p_ = xh + (size_t)((cc) - 1) * K;                                                                        \
        if (K == 16) {                                                                                                        \
            const float4 a_ = reinterpret_cast<const float4 *>(p_)[0], b_ = reinterpret_cast<const float4 *>(p_)[1];          \
            fl[0] = a_.x; fl[1] = a_.y; fl[2] = a_.z; fl[3] = a_.w; fl[4] = b_.x; fl[5] = b_.y; fl[6] = b_.z; fl[7] = b_.w;   \
        } else {                                                                                                              \
            /* unconditional loads (a half's zero component reads the next frame's first one: in bounds), then the select */   \
            _Pragma("unroll") for (int j = 0; j < CHM; ++j) { const float t_ = p_[j]; fl[j] = j < nvalid ? t_ : 0.f; }        \
        }                                                                                                                     \
    } while (0)

        // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums (of this half's components)
        float mu[CHM], fl[CHM];
#pragma unroll
        for (int j = 0; j < CHM; ++j) mu[j] = 0.f;
        // ten frames requested per wait: left rolled, every frame paid a whole memory round trip before its eight adds (same sums, same order)
#ifndef RP_WIDE_MEAN_UNROLL
#define RP_WIDE_MEAN_UNROLL 10
#endif
#pragma unroll RP_WIDE_MEAN_UNROLL
        for (int i = 1; i <= L; ++i) {
            RP_LOADF(i);
#pragma unroll
            for (int j = 0; j < CHM; ++j) mu[j] += fl[j];
        }
#pragma unroll
        for (int j = 0; j < CHM; ++j) mu[j] = mu[j] / (float)L;

        v2f Q[2][B + 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[p][q] = (v2f){RP_INF, RP_INF};
            Q[p][W - 1] = (v2f){0.f, 0.f};
        }
        u32x4 Areg[NTILE][KS];
#pragma unroll
        for (int g = 0; g < NTILE; ++g) {
            const int slot = 4 * g + jj;
            int r = W - ((W - slot + NS) % NS);
            r = r < 1 ? 1 : r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                Areg[g][ks] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)(r - 1) * kRowBytes + ks * 256);
        }
        v16f acc[NTILE];
        u32x4 bop[2][KS];
        float chk = 0.f;

// the frame in fl[] (column cc) -> B operand bop[par]: centre, scale to unit length (the two halves' squared norms meet through
// v_permlane32_swap; zero frame -> zero vector -> cost 1, comparator.rs:43-47), split in two f16 parts, pack
#define RP_PREP(par)                                                                                                          \
    do {                                                                                                                      \
        float d_[CHM], own_ = 0.f;                                                                                            \
        _Pragma("unroll") for (int j = 0; j < CHM; ++j) { d_[j] = fl[j] - mu[j]; own_ = fmaf(d_[j], d_[j], own_); }            \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(own_), __float_as_uint(own_), false, false);        \
        const float bb_ = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);                                                  \
        const float inv_ = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;                                                      \
        chk = fmaxf(fmaxf(chk, inv_), bb_); /* one v_max3_f32: the norm-range test (kDtwFixLimit, rp_kernels.h) */            \
        unsigned v_[4 * KS];                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4 * KS; ++i) v_[i] = 0u;                                                        \
        _Pragma("unroll") for (int j = 0; j < NPAIR; ++j) {                                                                   \
            const float ua_ = d_[2 * j] * inv_, ub_ = d_[2 * j + 1] * inv_;                                                   \
            const unsigned p_ = pkrtz(ua_, ub_);                                                                              \
            v_[3 * j] = p_; v_[3 * j + 2] = p_;                                                                               \
            v_[3 * j + 1] = pk_f16_second(ua_ - x0f(ua_), ub_ - x0f(ub_));                                                    \
        }                                                                                                                     \
        if (ODD) {                                                                                                            \
            const float us_ = d_[CHM - 1] * inv_;                                                                             \
            const unsigned t_ = pkrtz(us_, 0.f);                                                                              \
            v_[3 * NPAIR] = pk_f16_second(x0f(us_), us_ - x0f(us_)); /* x0 is already an f16 value */                         \
            v_[3 * NPAIR + 1] = __builtin_amdgcn_perm(0x3c000000u, t_, sel_c);                                                \
        }                                                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                                     \
            bop[par][ks] = (u32x4){v_[4 * ks], v_[4 * ks + 1], v_[4 * ks + 2], v_[4 * ks + 3]};                               \
    } while (0)
// the A tile that receives template row cc + W (cc = 1 + uu mod 12)
#define RP_AREF(cc, uu, GUARD)                                                                                                \
    {                                                                                                                         \
        const int sn = ((uu) + 1 + W) % NS, g = sn / 4, e = sn % 4;                                                           \
        int off = ((cc) + W - 1) * kRowBytes - (int)dl[e];                                                                    \
        if (GUARD) off = off < 0 ? 0 : off;                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                                     \
            Areg[g][ks] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)off + ks * 256);                         \
    }
#define RP_MFMA(g, par)                                                                                                       \
    do {                                                                                                                      \
        constexpr float c0_ = ODD ? 0.f : 1.f;   /* an even component count has no product slot for the 1 of 1 - a.x: it starts the sum */ \
        const v16f init16 = {c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_, c0_};                  \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g][0]), __builtin_bit_cast(f16x8, bop[par][0]), init16, 0, 0, 0); \
        _Pragma("unroll") for (int ks = 1; ks < KS; ++ks)                                                                     \
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g][ks]), __builtin_bit_cast(f16x8, bop[par][ks]), acc[g], 0, 0, 0); \
    } while (0)
#define RP_MFMA1(g, ks, par)                                                                                                  \
    do {                                                                                                                      \
        constexpr float c1_ = ODD ? 0.f : 1.f;                                                                                \
        const v16f in16_ = {c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_};                   \
        if ((ks) == 0) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g][0]), __builtin_bit_cast(f16x8, bop[par][0]), in16_, 0, 0, 0); \
        else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g][ks]), __builtin_bit_cast(f16x8, bop[par][ks]), acc[g], 0, 0, 0); \
    } while (0)
// column c (c = 1 + u mod 12): the B operand of column c + 2 is built first (its frame was requested one column earlier), then the
// frame of column c + 3 is requested, then the cells; each tile's MFMAs for column c + 1 go out after the last cell that reads the tile
#define RP_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        RP_AREF(c + 1, (u + 1) % NS, GUARD)                                                                                   \
        RP_PREP((u + 1) & 1);                                                                                                 \
        RP_LOADF(c + 3);   /* (requested two columns ahead through a ring of two: measured, no faster -- the L2 is not what this kernel waits for) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        v2f up[2] = {(v2f){RP_INF, RP_INF}, (v2f){RP_INF, RP_INF}};                                                           \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                   \
                const v2f cost = (v2f){acc[sl / 4][4 * (sl % 4) + 2 * p], acc[sl / 4][4 * (sl % 4) + 2 * p + 1]};             \
                v2f m, v;                                                                                                     \
                m.x = fminf(fminf(up[p].x, Q[p][q + 1].x), Q[p][q].x);                                                        \
                m.y = fminf(fminf(up[p].y, Q[p][q + 1].y), Q[p][q].y);                                                        \
                v.x = cost.x + m.x; v.y = cost.y + m.y;                                                                       \
                if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                              \
                Q[p][q] = v;                                                                                                  \
                up[p] = v;                                                                                                    \
            }                                                                                                                 \
            _Pragma("unroll") for (int g = 0; g < NTILE; ++g)                                                                 \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                             \
                    if (kIssue.at[u][g][ks] == q) RP_MFMA1(g, ks, u & 1);                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
    } while (0)

        RP_AREF(1, 0, true)
        RP_LOADF(1);
        RP_PREP(1);
        RP_MFMA(0, 1); RP_MFMA(1, 1); RP_MFMA(2, 1);
        RP_LOADF(2);
        RP_PREP(0);
        RP_LOADF(3);  // step c builds the B operand of column c + 2 from fl[] at its top: fl[] holds column 3 for step 1
        __builtin_amdgcn_sched_barrier(0);
        int c0 = 1;
        bool dead = false;
        {   // first block: cells of rows < 1 stay +inf (L >= 12)
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(true); }
        }
#define RP_ABANDON_CHECK()                                                                                                    \
    if (abandon_nc < RP_INF) {                                                                                                \
        bool alive = false;                                                                                                   \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                       \
            v2f m = Q[p][0];                                                                                                  \
            _Pragma("unroll") for (int q = 1; q < B; ++q) m = (v2f){fminf(m.x, Q[p][q].x), fminf(m.y, Q[p][q].y)};             \
            alive = alive || (slot_real[2 * p] && (m.x <= abandon_cost || slot_avg[2 * p])) ||                                \
                    (slot_real[2 * p + 1] && (m.y <= abandon_cost || slot_avg[2 * p + 1]));                                   \
        }                                                                                                                     \
        if (!__any(alive && valid)) dead = true;                                                                              \
    }
        for (c0 = 1 + NS; c0 + NS - 1 <= L; c0 += NS) {
            RP_ABANDON_CHECK()
            if (dead) break;
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(false); }
        }
        if (!dead && c0 <= L) {
            RP_ABANDON_CHECK()
            if (!dead) {
#pragma unroll
                for (int u = 0; u < NS - 1; ++u) {
                    const int c = c0 + u;
                    if (c <= L) RP_STEP(false);
                }
            }
        }
#undef RP_ABANDON_CHECK
#undef RP_STEP
#undef RP_MFMA
#undef RP_MFMA1
#undef RP_AREF
#undef RP_PREP
#undef RP_LOADF

        if (valid) {
            const size_t row = s * out_win_pitch + (size_t)w;
            const float denom = (float)(L + L);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int slot = 4 * h + 2 * p + e;
                    if (slot < ch->count) {
                        const float cost = e ? Q[p][W - 2].y : Q[p][W - 2].x;
                        const float nc = cost / denom;
                        const float sc = dead ? 0.f : 1.f / (1.f + expf((nc - score_ref) / score_ref));
                        const int t = ch->tid[slot];
                        if (t < T) scores[row * T + t] = sc;
                        else if (!dead) avg[row] = sc;
                    }
                }
            }
            // a frame outside the norm range (both lane halves saw the same squared norms): listed for dtw_ref_kernel
            if (h == 0 && chk > kDtwFixLimit) dtw_fix_append(fix, row, (uint32_t)(chunk_base + (int)ci));
        }
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(next_tile + 1, 1u) == n_groups - 1) {
            next_tile[0] = 0;
            next_tile[1] = 0;
        }
    }
}

bool dtw_mfma_wide_supported(const TemplatesDev &t, int band, float score_ref) {
    if (!(score_ref >= kDtwMfmaMinScoreRef)) return false;   // as dtw_mfma_supported
    if (t.arith_mode() != kArithFastSplit || (t.K != 13 && t.K != 16) || band != 5 || !t.aimg || t.wide8_count <= 0 || t.max_diff != 0) return false;
    if (t.mfma_min_len < kWSlots) return false;
    return (size_t)(t.max_len + 16) * dtw_mfma_wide_row_bytes(t.K) <= 160 * 1024;
}

hipError_t launch_dtw_mfma_wide(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg, const uint32_t *list,
                                const uint32_t *count, uint32_t dense_min, float abandon_nc) {
    const int n_chunks = t.wide8_count;
    if (n_chunks <= 0 || S == 0 || n_win == 0) return hipSuccess;
    if (band != 5 || !wk.sched || !wk.fix) return hipErrorNotSupported;
    dtw_mark(wk, kDtwRanMfmaWide | kDtwRanF16x2);
    const size_t total_tiles = (S * n_win + kWWin - 1) / kWWin;
    constexpr int NW = 8;
    const size_t lds = (size_t)(t.max_len + 16) * dtw_mfma_wide_row_bytes(t.K);
    size_t groups = (size_t)device_cu_count() / (size_t)n_chunks;
    if (groups < 1) groups = 1;
    const size_t need = (total_tiles + NW - 1) / NW;
    if (groups > need) groups = need;
    const size_t blocks = groups * (size_t)n_chunks;
    const unsigned static_rounds = mfma_static_rounds(total_tiles, groups * (size_t)NW, list != nullptr);
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
#define RP_LAUNCH_WIDE(KK)                                                                                                          \
    do {                                                                                                                            \
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_mfma_wide_kernel<KK, 5, NW>), 160 * 1024); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dtw_mfma_wide_kernel<KK, 5, NW>), dim3((unsigned)blocks), dim3(64 * NW), lds, st, mfcc, frame_pitch, total_tiles, \
                           (unsigned)n_chunks, t.wide8_first, first_win, n_win, out_win_pitch, t.chunks, reinterpret_cast<const uint4 *>(t.aimg), \
                           t.T, score_ref, scores, avg, S, list, count, dense_min, abandon_nc, wk.sched, static_rounds, wk.fix);         \
    } while (0)
    if (t.K == 16) RP_LAUNCH_WIDE(16);
    else RP_LAUNCH_WIDE(13);
#undef RP_LAUNCH_WIDE
    return hipGetLastError();
}

}  // namespace rp
