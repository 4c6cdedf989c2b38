// rp_dtw_mfma_wide3.hip -- dtw_mfma_wide3_kernel: the matrix-core DTW for mfcc_size 13 and 16 in the f32-GRADE arithmetic (RP_ARITH_F32_MATRIX,
// round 6).  Same scoring as every other DTW kernel (src/mfcc/dtw.rs:56-105 + comparator.rs:15-48 + normalizer.rs:17-29 + wakeword_comp.rs:22-37),
// same sweep and recurrence as dtw_mfma_wide_kernel (rp_dtw_mfma_wide.hip), other products and another shape:
//   * products: both operands as THREE bf16 parts, exactly (x0 = x & 0xffff0000, r = x - x0, x1 = r & 0xffff0000, x2 = r - x1; the template
//     rows on the host, rounded to nearest), and the six partial products x_i a_j with i + j <= 2 of every component: 6 x 16 = 96 k-slots for
//     mfcc_size 16 = SIX v_mfma_f32_32x32x16_bf16 per tile and column, chained on one accumulator that starts at the 1 of 1 - a.x (mfcc_size 13:
//     78 slots + the constant's own).  What is dropped is below 2^-22 of a product (2^-25.7 rms; an f32 multiply rounds by up to 2^-24).
//   * shape: FOUR template slots per wave, not eight -- the A image of eight would be 1 664 bytes per template row, 193 KB at 100 frames, beyond
//     the CU's LDS; of four it is 768 + 64 bytes (96 KB at 100 frames; the 64 put the four row slots a ds_read_b128 serves on different
//     banks).  A tile is 8 row slots x 4 templates, 16 circular row slots = 2 tiles, a lane (window, half h) runs ONE template pair
//     (templates 2h, 2h + 1): the C/D layout puts row 8 G + 4 h + 2 sp + e = (row slot 2 G + sp, template 2 h + e) into register 4 G + 2 sp + e.
//     Columns are unrolled 16 at a time.  12 matrix instructions per column and four templates.
//   * lane half h owns CHM = ceil(K / 2) components.  Per component pair (a, b) the window side has three registers P0 = (x0a, x0b), P1, P2; the
//     products need P0 three times (against a0, a1, a2), P1 twice, P2 once -- 24 register slots per half and k-step run, but only twelve different
//     registers.  They are kept as ONE run and the six k-steps read overlapping four-register pieces of it (w3_run_piece; the layouts of both
//     frame sizes and the A image's side of every slot: dtw_mfma_wide3_slot, rp_kernels.h).  mfcc_size 13's odd seventh component rides in two
//     registers, (x0s, x1s) read three times and (x2s, the constant 1.0 of half 1) -- an even component count starts the sum at 1 instead.
//   * schedule: two run buffers (column c's operand in brun[c & 1]).  In every column one of the two tiles is still read by the last or last but
//     one band cell, so its chain of six dependent matrix instructions cannot leave before the cells are done; it goes out between the pieces
//     of the NEXT frame's preparation (its head, one piece per component pair, the odd component), which writes the other buffer
//     (w3_issue_table).  6.27-6.29 ms against 6.39-6.41 for one buffer at 8 192 streams x 8 templates of mfcc_size 16.
// Frames are read from global memory (the caller leaves slack behind the last stream's frames, launch_dtw `padded_rows`), as in the two-part
// kernel.  Two waves per SIMD (eight per workgroup), nothing spilled.
// BUILD: the 16-column block (192 matrix instructions, ~2 700 instructions) is a `#pragma unroll` loop far beyond the compiler's default budget for
// pragma-requested full unrolling; this file is compiled with -mllvm -pragma-unroll-threshold=200000 (Makefile FILE_FLAGS_rp_dtw_mfma_wide3.hip).
// Without it the loop stays rolled, the row-slot -> accumulator mapping becomes a run-time index and the build shows 256 registers with
// 850-1 400 spilled values -- which is how this kernel was first written off.
#include "rp_device.h"

#include <cstdlib>

namespace rp {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x12 __attribute__((ext_vector_type(12)));

// k-step ks of the window side's operand: a four-register piece of the half's run of twelve (see above)
__device__ __forceinline__ u32x4 w3_run_piece(const u32x12 &r, int ks) {
    switch (ks) {
    case 0: return __builtin_shufflevector(r, r, 0, 1, 2, 3);
    case 1: return __builtin_shufflevector(r, r, 2, 3, 4, 5);
    case 2: case 3: return __builtin_shufflevector(r, r, 4, 5, 6, 7);
    case 4: return __builtin_shufflevector(r, r, 6, 7, 8, 9);
    default: return __builtin_shufflevector(r, r, 8, 9, 10, 11);
    }
}

constexpr int kW3Win = 32, kW3Slots = 16, kW3Tiles = 2, kW3SPT = 8, kW3KS = kDtwWide3KSteps;

// accumulator register of (row slot, pair element e) in its tile (see above)
__host__ __device__ constexpr int w3_acc_reg(int slot, int e) { return 4 * ((slot % 8) >> 1) + 2 * (slot & 1) + e; }

template <int W>
__host__ __device__ constexpr int w3_last_use(int u, int g) {
    int last = -1;
    for (int q = 0; q < 2 * W; ++q)
        if (((u + q + kW3Slots - W + 2) % kW3Slots) / kW3SPT == g) last = q;
    return last;
}

// issue point of k-step ks of tile g in column phase u: points 0 .. 2W - 1 lie behind the band cells, 2W .. 2W + tail - 1 behind the pieces of the
// next frame's preparation.  A tile's chain goes out one instruction per point from the cell that reads the tile last.
struct W3IssueTable { int at[kW3Slots][kW3Tiles][kW3KS]; };
template <int W>
__host__ __device__ constexpr W3IssueTable w3_issue_table(int tail) {
    W3IssueTable t{};
    for (int u = 0; u < kW3Slots; ++u)
        for (int g = 0; g < kW3Tiles; ++g)
            for (int ks = 0; ks < kW3KS; ++ks) {
                int at = w3_last_use<W>(u, g) + ks;
                if (at < 0) at = 0;
                if (at > 2 * W + tail - 1) at = 2 * W + tail - 1;
                t.at[u][g][ks] = at;
            }
    return t;
}

__device__ __forceinline__ unsigned hi2(float hi, float lo) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

}  // namespace

template <int K, int W, int NW>
__global__ __launch_bounds__(64 * NW, 1) void dtw_mfma_wide3_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t total_tiles, unsigned n_chunks, int chunk_base, size_t first_win,
    size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks, const uint4 *__restrict__ aimg, int T, float score_ref,
    float *__restrict__ scores, float *__restrict__ avg, size_t n_streams, const uint32_t *__restrict__ list,
    const uint32_t *__restrict__ count, uint32_t dense_min, float abandon_nc, uint32_t *__restrict__ sched, unsigned static_rounds, uint32_t *__restrict__ fix) {
    constexpr int B = 2 * W, NS = kW3Slots, NTILE = kW3Tiles, SPT = kW3SPT, KS = kW3KS;
    constexpr int CHM = dtw_mfma_wide_chm(K), NPAIR = CHM / 2, ODD = CHM % 2;
    constexpr int kRowBytes = kDtwWide3RowBytes;
    constexpr int TAIL = 1 + NPAIR + ODD;         // issue points behind the cells: the preparation's head, one per component pair, the odd component
    constexpr W3IssueTable kIssue = w3_issue_table<W>(TAIL);
    static_assert(B + 2 <= NS, "the band and its two neighbours must fit the 16 row slots");
    static_assert(K == 13 || K == 16, "dtw_mfma_wide3_slot knows these two layouts");
    size_t total_entries = n_streams * n_win;
    if (list) {
        const uint32_t n_listed = *count;
        if (dense_min && n_listed >= dense_min) return;
        total_entries = n_listed;
        total_tiles = ((size_t)n_listed + kW3Win - 1) / kW3Win;
    } else if (count && *count < dense_min) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned ci = blockIdx.x % n_chunks;
    const unsigned n_groups = gridDim.x / n_chunks;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const u32x4 *asrc = reinterpret_cast<const u32x4 *>(aimg) + ch->aimg3_off;
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < (L + 16) * kRowBytes / 16; i += 64 * NW) adst[i] = asrc[i];
    }
    __syncthreads();
    const int n = lane & 31, h = lane >> 5;
    // A operand: this lane supplies row m = lane & 31 of a tile, k half = lane >> 5: m = 8 G + 4 h' + 2 sp + e = (row slot 2 G + sp, template 2 h' + e)
    const int mrow = lane & 31, jj = 2 * (mrow >> 3) + ((mrow >> 1) & 1), tA = ((mrow >> 2) & 1) * 2 + (mrow & 1);
    const unsigned a_lane = (unsigned)(h * 64 + tA * 16);
    // rows back from the newest row (in slot e of its tile) to the row this lane's slot holds: (e - jj) mod 16
    const int nvalid = h ? K - CHM : CHM;  // components this half really has (mfcc_size 13: 7 and 6)
    const float abandon_cost = abandon_nc * (float)(L + L);
    bool slot_real[2], slot_avg[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { slot_real[e] = 2 * h + e < ch->count; slot_avg[e] = slot_real[e] && ch->tid[2 * h + e] >= T; }

    uint32_t *next_tile = sched + 2 * (chunk_base + ci);
    unsigned round = 0;
    const size_t chunk_waves = (size_t)n_groups * NW;  // waves working on this chunk
    for (;;) {
        size_t tile;
        if (round < static_rounds) {   // as dtw_mfma_kernel: the first round(s) by index, the rest from the chunk's counter
            tile = (size_t)round * chunk_waves + (size_t)(blockIdx.x / n_chunks) * NW + (size_t)(tid >> 6);
            ++round;
        } else {
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(next_tile, 1u);
            tile = (size_t)__builtin_amdgcn_readfirstlane(ticket) + (size_t)static_rounds * chunk_waves;
        }
        if (tile >= total_tiles) break;
        size_t f = tile * kW3Win + n;
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
#pragma unroll 10
        for (int i = 1; i <= L; ++i) {
            RP_LOADF(i);
#pragma unroll
            for (int j = 0; j < CHM; ++j) mu[j] += fl[j];
        }
#pragma unroll
        for (int j = 0; j < CHM; ++j) mu[j] = mu[j] / (float)L;

        v2f Q[B + 1];
#pragma unroll
        for (int q = 0; q <= B; ++q) Q[q] = (v2f){RP_INF, RP_INF};
        Q[W - 1] = (v2f){0.f, 0.f};
        u32x4 Areg[NTILE][KS];
#pragma unroll
        for (int g = 0; g < NTILE; ++g) {
            const int slot = SPT * g + jj;
            int r = W - ((W - slot + NS) % NS);
            r = r < 1 ? 1 : r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                Areg[g][ks] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)(r - 1) * kRowBytes + ks * 128);
        }
        v16f acc[NTILE];
        u32x12 brun[2];     // the window side's operand: column c's in brun[c & 1]
        float chk = 0.f;
        float pd_[CHM], pinv_ = 0.f;   // the frame being prepared: centred components, 1 / norm

// the frame in fl[] (column cc) -> the window side's operand brun[par]: centre, scale to unit length (the two halves' squared norms meet through
// v_permlane32_swap; zero frame -> zero vector -> cost 1, comparator.rs:43-47), split in three bf16 parts, pack
#define RP_PREP_HEAD()                                                                                                        \
    do {                                                                                                                      \
        float own_ = 0.f;                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < CHM; ++j) { pd_[j] = fl[j] - mu[j]; own_ = fmaf(pd_[j], pd_[j], own_); }         \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(own_), __float_as_uint(own_), false, false);        \
        const float bb_ = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);                                                  \
        pinv_ = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;                                                                 \
        chk = fmaxf(fmaxf(chk, pinv_), bb_); /* one v_max3_f32: the norm-range test (kDtwFixLimit, rp_kernels.h) */           \
    } while (0)
// component pair j of the prepared frame -> its three registers of brun[par]: pairs 0, 1 from the front of the run, 2, 3 from its back
#define RP_PREP_PAIR(par, j)                                                                                                  \
    do {                                                                                                                      \
        const float ua_ = pd_[2 * (j)] * pinv_, ub_ = pd_[2 * (j) + 1] * pinv_;                                               \
        const float ra_ = ua_ - top16(ua_), rb_ = ub_ - top16(ub_);                                                           \
        brun[par][4 + (j)] = hi2(ub_, ua_);                                                                                   \
        brun[par][(j) < 2 ? 2 + (j) : 6 + (j)] = hi2(rb_, ra_);                                                               \
        brun[par][(j) < 2 ? (j) : 8 + (j)] = hi2(rb_ - top16(rb_), ra_ - top16(ra_));                                         \
    } while (0)
// mfcc_size 13's seventh component: (x0s, x1s) where the fourth pair's P0 would be, (x2s, the constant 1.0 of half 1) for its P1, 0 for its P2
#define RP_PREP_ODD(par)                                                                                                      \
    do {                                                                                                                      \
        const float us_ = pd_[CHM - 1] * pinv_, rs_ = us_ - top16(us_);                                                       \
        brun[par][7] = hi2(rs_, us_);                                                                                         \
        brun[par][9] = hi2(h ? 1.f : 0.f, rs_ - top16(rs_));                                                                  \
        brun[par][11] = 0u;                                                                                                   \
    } while (0)
#define RP_PREP(par)                                                                                                          \
    do {                                                                                                                      \
        RP_PREP_HEAD();                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < NPAIR; ++j) RP_PREP_PAIR(par, j);                                               \
        if (ODD) RP_PREP_ODD(par);                                                                                            \
    } while (0)
// the A tile that receives template row cc + W (cc = 1 + uu mod 16)
#define RP_AREF(cc, uu, GUARD)                                                                                                \
    {                                                                                                                         \
        const int sn = ((uu) + 1 + W) % NS, g = sn / SPT, e = sn % SPT;                                                       \
        int off = ((cc) + W - 1 - ((e - jj + NS) & (NS - 1))) * kRowBytes;                                                    \
        if (GUARD) off = off < 0 ? 0 : off;                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                                     \
            Areg[g][ks] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)off + ks * 128);                         \
    }
#define RP_MFMA1(g, ks, par)                                                                                                  \
    do {                                                                                                                      \
        constexpr float c1_ = ODD ? 0.f : 1.f;   /* an even component count has no product slot for the 1 of 1 - a.x: it starts the sum */ \
        const v16f in16_ = {c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_, c1_};                   \
        const u32x4 b4_ = w3_run_piece(brun[par], ks);                                                        \
        if ((ks) == 0) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g][0]), __builtin_bit_cast(bf16x8, b4_), in16_, 0, 0, 0); \
        else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g][ks]), __builtin_bit_cast(bf16x8, b4_), acc[g], 0, 0, 0); \
    } while (0)
// column c (c = 1 + u mod 16): the B operand of column c + 2 is built first (its frame was requested one column earlier), then the
// frame of column c + 3 is requested, then the cells; a tile's k-steps for column c + 1 go out after the last cell that reads the tile
#define RP_ISSUE(at_)                                                                                                         \
    _Pragma("unroll") for (int g = 0; g < NTILE; ++g)                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                                     \
            if (kIssue.at[u][g][ks] == (at_)) RP_MFMA1(g, ks, MPAR);
#define RP_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        /* operand buffers (c0 is odd): column c + 1's in brun[(c + 1) & 1] = brun[u & 1], column c + 2's goes to the other */        \
        const int MPAR = u & 1, PPAR = (u + 1) & 1;                                                                            \
        RP_AREF(c + 1, (u + 1) % NS, GUARD)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        v2f up = (v2f){RP_INF, RP_INF};                                                                                       \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            const v2f cost = (v2f){acc[sl / SPT][w3_acc_reg(sl, 0)], acc[sl / SPT][w3_acc_reg(sl, 1)]};                       \
            v2f m, v;                                                                                                         \
            m.x = fminf(fminf(up.x, Q[q + 1].x), Q[q].x);                                                                     \
            m.y = fminf(fminf(up.y, Q[q + 1].y), Q[q].y);                                                                     \
            v.x = cost.x + m.x; v.y = cost.y + m.y;                                                                           \
            if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                                  \
            Q[q] = v;                                                                                                         \
            up = v;                                                                                                           \
            RP_ISSUE(q)                                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
        /* column c + 2's operand, piece by piece, the rest of column c + 1's matrix instructions between the pieces */           \
        RP_PREP_HEAD();                                                                                                       \
        RP_ISSUE(B)                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NPAIR; ++j) {                                                                   \
            RP_PREP_PAIR(PPAR, j);                                                                                            \
            RP_ISSUE(B + 1 + j)                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
        if (ODD) {                                                                                                            \
            RP_PREP_ODD(PPAR);                                                                                                \
            RP_ISSUE(B + 1 + NPAIR)                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
        RP_LOADF(c + 3);                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    } while (0)

        RP_AREF(1, 0, true)
        RP_LOADF(1);
        RP_PREP(1);                // column 1's operand (column c's lives in brun[c & 1])
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { RP_MFMA1(0, ks, 1); RP_MFMA1(1, ks, 1); }
        RP_LOADF(2);
        __builtin_amdgcn_sched_barrier(0);
        RP_PREP(0);                // column 2's
        RP_LOADF(3);  // step c builds the B operand of column c + 2 from fl[] at its end: fl[] holds column 3 for step 1
        __builtin_amdgcn_sched_barrier(0);
        int c0 = 1;
        bool dead = false;
        {   // first block: cells of rows < 1 stay +inf (L >= 16)
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(true); }
        }
#define RP_ABANDON_CHECK()                                                                                                    \
    if (abandon_nc < RP_INF) {                                                                                                \
        v2f m = Q[0];                                                                                                         \
        _Pragma("unroll") for (int q = 1; q < B; ++q) m = (v2f){fminf(m.x, Q[q].x), fminf(m.y, Q[q].y)};                       \
        const bool alive = (slot_real[0] && (m.x <= abandon_cost || slot_avg[0])) || (slot_real[1] && (m.y <= abandon_cost || slot_avg[1])); \
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
#undef RP_ISSUE
#undef RP_MFMA1
#undef RP_AREF
#undef RP_PREP
#undef RP_PREP_ODD
#undef RP_PREP_PAIR
#undef RP_PREP_HEAD
#undef RP_LOADF

        if (valid) {
            const size_t row = s * out_win_pitch + (size_t)w;
            const float denom = (float)(L + L);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int slot = 2 * h + e;
                if (slot < ch->count) {
                    const float cost = e ? Q[W - 2].y : Q[W - 2].x;   // D[m - 1][n]: band position W - 2 (dtw.rs:101)
                    const float nc = cost / denom;
                    const float sc = dead ? 0.f : 1.f / (1.f + expf((nc - score_ref) / score_ref));
                    const int t = ch->tid[slot];
                    if (t < T) scores[row * T + t] = sc;
                    else if (!dead) avg[row] = sc;
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

bool dtw_mfma_wide3_supported(const TemplatesDev &t, int band) {
    if (t.arith_mode() != kArithF32Matrix || (t.K != 13 && t.K != 16) || band != 5 || !t.aimg3 || t.wide4_count <= 0 || t.max_diff != 0) return false;
    if (t.wide4_min_len < kW3Slots) return false;   // the first 16 columns are one guarded block
    return (size_t)(t.max_len + 16) * kDtwWide3RowBytes <= 160 * 1024;
}

hipError_t launch_dtw_mfma_wide3(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                 size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg, const uint32_t *list,
                                 const uint32_t *count, uint32_t dense_min, float abandon_nc) {
    const int n_chunks = t.wide4_count;
    if (n_chunks <= 0 || S == 0 || n_win == 0) return hipSuccess;
    if (band != 5 || !wk.sched || !wk.fix) return hipErrorNotSupported;
    dtw_mark(wk, kDtwRanMfmaWide | kDtwRanBf16x3);
    const size_t total_tiles = (S * n_win + kW3Win - 1) / kW3Win;
    constexpr int NW = 8;
    const size_t lds = (size_t)(t.max_len + 16) * kDtwWide3RowBytes;
    size_t groups = (size_t)device_cu_count() / (size_t)n_chunks;
    if (groups < 1) groups = 1;
    const size_t need = (total_tiles + NW - 1) / NW;
    if (groups > need) groups = need;
    const size_t blocks = groups * (size_t)n_chunks;
    const unsigned static_rounds = mfma_static_rounds(total_tiles, groups * (size_t)NW, list != nullptr);
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
#define RP_LAUNCH_WIDE3(KK)                                                                                                         \
    do {                                                                                                                            \
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_mfma_wide3_kernel<KK, 5, NW>), 160 * 1024); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dtw_mfma_wide3_kernel<KK, 5, NW>), dim3((unsigned)blocks), dim3(64 * NW), lds, st, mfcc, frame_pitch, total_tiles, \
                           (unsigned)n_chunks, t.wide4_first, first_win, n_win, out_win_pitch, t.chunks, reinterpret_cast<const uint4 *>(t.aimg3), \
                           t.T, score_ref, scores, avg, S, list, count, dense_min, abandon_nc, wk.sched, static_rounds, wk.fix);         \
    } while (0)
    if (t.K == 16) RP_LAUNCH_WIDE3(16);
    else RP_LAUNCH_WIDE3(13);
#undef RP_LAUNCH_WIDE3
    return hipGetLastError();
}

}  // namespace rp
