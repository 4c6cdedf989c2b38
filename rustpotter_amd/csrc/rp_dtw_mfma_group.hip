// rp_dtw_mfma_group.hip -- dtw_mfma_group_kernel: dtw_mfma_kernel (rp_dtw_mfma.hip, eight template slots) for references with SEVERAL chunks of
// the same template length -- BASELINE config C4: 64 templates of 100 frames = eight chunks.  There every chunk's wave centres, scales and
// splits the SAME frames of the same windows: 28 of the 108 vector instructions of a column, and the vector pipe is what binds the kernel.
//
//   * A workgroup of twelve waves holds SH (4 or 2) chunks of one length -- their A images side by side in LDS -- and 12 / SH tiles of 32
//     windows: wave (tile j, chunk i).  The SH waves of a tile run its columns together; the B operand of a column (64 lanes x 16 bytes) is
//     built by ONE of them and read by all through an LDS ring: columns in blocks of four, wave i builds column 4 b + 1 + i of block b (two chunks: also 4 b + 3 + i) while
//     block b - 1 is consumed, one workgroup barrier per block (two ring buffers of 4 KB per tile).  With SH = 4 a wave does the frame work
//     of every fourth column: 7 instead of 28 instructions per column.
//   * Everything else is dtw_mfma_kernel's: lane = (window, half), two template pairs per lane, twelve circular row slots = three tiles, the
//     A tile of the next column re-read per step, each tile's MFMA issued behind the last cell that reads it, the first twelve columns
//     guarded, D[m - 1][n] at band position W - 2.  Same operations on the same values: the scores are BIT-IDENTICAL to dtw_mfma_kernel's
//     (tests/test_gpu_dtw_mfma.py), including the list of windows outside the norm range (the partial range tests of a tile's waves meet
//     in LDS; columns 1 .. L + 2 are tested, as there).
//   * Whole-batch calls only (frames staged in LDS, every window scored, no early abandon -- a wave that stops would leave its tile's
//     barrier short); tile groups are handed out by index (the launches this form is for are hundreds of rounds long).
// Reference arithmetic: src/mfcc/dtw.rs:56-105, comparator.rs:15-48, normalizer.rs:17-29, wakeword_comp.rs:22-37 (as rp_dtw_mfma.hip).
#include "rp_device.h"

#include <cstdlib>

namespace rp {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGK = 5, kGWin = 32, kGSlots = 12, kGTiles = 3, kGWaves = 12, kGAhead = 8;   // kGAhead: columns staged behind the window (built, never used)

__device__ __forceinline__ unsigned pkrtz_g(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ float x0f_g(float x) { return __uint_as_float(__float_as_uint(x) & 0xffffe000u); }

template <int W>
__host__ __device__ constexpr int group_last_use(int u, int g) {
    int last = -1;
    for (int q = 0; q < 2 * W; ++q)
        if (((u + q + kGSlots - W + 2) % kGSlots) / 4 == g) last = q;
    return last;
}

__host__ __device__ inline int group_stage_floats(int L) { return ((kGWin + 2 * (L + kGAhead)) * kGK + 3) & ~3; }

}  // namespace

size_t dtw_mfma_group_lds_bytes(int L, int sh) {
    const int tiles = kGWaves / sh;
    return (size_t)sh * (size_t)(L + 16) * kDtwMfmaRowBytes + (size_t)tiles * (size_t)group_stage_floats(L) * sizeof(float) +
           (size_t)tiles * 2 * 4 * 1024 + (size_t)kGWaves * 64 * sizeof(float);
}

#ifdef RP_GROUP_NO_BARRIER   // timing experiment only (results wrong): what the per-block barrier costs
#define RP_GROUP_SYNC() __builtin_amdgcn_s_waitcnt(0)
#else
#define RP_GROUP_SYNC() __syncthreads()
#endif
template <int W, int SH>
__global__ __launch_bounds__(64 * kGWaves, 1) void dtw_mfma_group_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, size_t total_tiles, unsigned n_cgroups, const int *__restrict__ grp_first,
    size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks, const uint4 *__restrict__ aimg, int T, float score_ref,
    float *__restrict__ scores, size_t n_streams, uint32_t *__restrict__ fix) {
    constexpr int K = kGK, B = 2 * W, NS = kGSlots, NTILE = kGTiles, NP = 2, TPW = kGWaves / SH;
    constexpr int kRowBytes = kDtwMfmaRowBytes;
    static_assert(B + 2 <= NS, "the band and its two neighbours must fit the circular row slots");
    static_assert(SH == 4 || SH == 2, "chunks per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned cg = blockIdx.x % n_cgroups, n_groups = gridDim.x / n_cgroups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tj = wave / SH, ci = wave % SH;   // this wave: tile slot, chunk of the group
    const int chunk_id = grp_first[cg] + ci;
    const DtwChunk *ch = chunks + chunk_id;
    const int L = ch->len;  // the group's: m == n == L
    const int a_bytes = (L + 16) * kRowBytes;
    const int xs_floats = group_stage_floats(L);
    // ---- LDS: SH A images | TPW frame stages | TPW rings of 2 x 4 columns x 1 KB | the waves' range tests
    for (int g = 0; g < SH; ++g) {
        const u32x4 *asrc = reinterpret_cast<const u32x4 *>(aimg) + chunks[grp_first[cg] + g].aimg_off;
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem + (size_t)g * a_bytes);
        for (int i = tid; i < a_bytes / 16; i += 64 * kGWaves) adst[i] = asrc[i];
    }
    const unsigned char *a_img = smem + (size_t)ci * a_bytes;
    float *xs = reinterpret_cast<float *>(smem + (size_t)SH * a_bytes) + (size_t)tj * xs_floats;
    unsigned char *ring = smem + (size_t)SH * a_bytes + (size_t)TPW * xs_floats * sizeof(float) + (size_t)tj * 8192;
    float *chk_lds = reinterpret_cast<float *>(smem + (size_t)SH * a_bytes + (size_t)TPW * xs_floats * sizeof(float) + (size_t)TPW * 8192);
    const int n = lane & 31, h = lane >> 5;
    const int mrow = lane & 31;
    const int jj = mrow >> 3;
    const int tA = ((mrow >> 2) & 1) * 4 + (mrow & 3);
    const unsigned a_lane = (unsigned)(h * 128 + tA * 16);
    unsigned dl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dl[e] = (unsigned)(((e - jj + NS) % NS) * kRowBytes);
    const unsigned sel_one = h ? 0x07060100u : 0x03020100u;

    const size_t total_entries = n_streams * n_win;
    const size_t n_tg = (total_tiles + TPW - 1) / TPW;
    for (size_t tg = blockIdx.x / n_cgroups; tg < n_tg; tg += n_groups) {
        const size_t tile_raw = tg * TPW + (size_t)tj;
        const bool tile_ok = tile_raw < total_tiles;
        const size_t tile = tile_ok ? tile_raw : total_tiles - 1;   // a tile slot past the end repeats the last tile (its barriers are needed) and writes nothing
        const size_t f0 = tile * kGWin;
        // ---- stage the tile's frames (up to two stream segments), a share per wave of the tile
        const size_t sA = f0 / n_win;
        const int wA = (int)(f0 - sA * n_win);
        const int nA = (int)n_win - wA < kGWin ? (int)n_win - wA : kGWin;
        const int nB = (nA < kGWin && sA + 1 < n_streams) ? kGWin - nA : 0;
        const int segA = nA + L + kGAhead;
        {
            auto stage = [&](const float *src, size_t g0, int n_floats, float *dst) {
                for (int i0 = lane + 64 * ci; i0 < n_floats; i0 += 256 * SH) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = i0 + 64 * SH * j;
                        v[j] = (i < n_floats && g0 + (size_t)(i / K) < n_frames_total) ? src[g0 * K + i] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i0 + 64 * SH * j < n_floats) dst[i0 + 64 * SH * j] = v[j];
                }
            };
            stage(mfcc + sA * frame_pitch * K, first_win + wA, segA * K, xs);
            if (nB > 0) stage(mfcc + (sA + 1) * frame_pitch * K, first_win, (nB + L + kGAhead) * K, xs + segA * K);
        }
        __syncthreads();   // (also: the A images are in place; the previous tile's ring and range tests are done with)
        const bool inA = n < nA;
        const bool valid = tile_ok && (inA || (n - nA < nB)) && f0 + (size_t)n < total_entries;
        const size_t s = inA ? sA : sA + 1;
        const int w = inA ? wA + n : n - nA;
        const float *xw = xs + (inA ? n : ((n - nA < nB) ? segA + n - nA : 0)) * K;
        const float *xa = xw + (h ? 3 : 0);
        const float *x2 = xw + 2;
        // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums of this lane's three components -- the SH waves of a
        // tile need the same means: with four chunks wave 0 / 1 / 2 sums one component each for all of them (same additions in the same order)
        float mua = 0.f, mub = 0.f, mu2 = 0.f;
        if (SH == 4) {
            if (ci < 3) {
                const float *xp = ci == 0 ? xa : ci == 1 ? xa + 1 : x2;
                float acc_ = 0.f;
#pragma unroll 8
                for (int i = 0; i < L; ++i) acc_ += xp[i * K];
                chk_lds[(tj * SH + ci) * 64 + lane] = acc_ / (float)L;
            }
            __syncthreads();
            mua = chk_lds[(tj * SH + 0) * 64 + lane]; mub = chk_lds[(tj * SH + 1) * 64 + lane]; mu2 = chk_lds[(tj * SH + 2) * 64 + lane];
            // (the range tests reuse these words at the end of the tile, behind the column loop's barriers)
        } else {
#pragma unroll 8
            for (int i = 0; i < L; ++i) { mua += xa[i * K]; mub += xa[i * K + 1]; mu2 += x2[i * K]; }
            mua = mua / (float)L; mub = mub / (float)L; mu2 = mu2 / (float)L;
        }

        v2f Q[NP][B + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[p][q] = (v2f){RP_INF, RP_INF};
            Q[p][W - 1] = (v2f){0.f, 0.f};
        }
        u32x4 Areg[NTILE];
#pragma unroll
        for (int g = 0; g < NTILE; ++g) {
            const int slot = 4 * g + jj;
            int r = W - ((W - slot + NS) % NS);
            r = r < 1 ? 1 : r;
            Areg[g] = *reinterpret_cast<const u32x4 *>(a_img + a_lane + (unsigned)(r - 1) * kRowBytes);
        }
        v16f acc[NTILE];
        u32x4 bop;
        float chk_ = 0.f;
        unsigned r0 = 0;   // ring buffer of the block that holds column c0 (flips every twelve columns: three blocks)

// the B operand of column cbase + pos (wave-uniform) of this wave's tile into ring buffer byte offset wb, position pos: dtw_mfma_kernel's
// pieces P0..P9 -- in a step they stand between the cells like there (a chain of ~25 dependent instructions in one piece leaves the wave
// waiting on itself; measured: the whole gain of the shared operand)
#define GP0(pos, cbase) cp_ = (cbase) + (pos); fa_ = xa[(cp_ - 1) * K]; fb_ = xa[(cp_ - 1) * K + 1]; f2_ = x2[(cp_ - 1) * K];
#define GP1() da_ = fa_ - mua; db_ = fb_ - mub; d2_ = f2_ - mu2;
#define GP2() own_ = fmaf(da_, da_, db_ * db_);
#define GP3() { const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(own_), __float_as_uint(own_), false, false); \
                bb_ = fmaf(d2_, d2_, __uint_as_float(sw_[0]) + __uint_as_float(sw_[1])); }
#define GP4() inv_ = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;
#define GP5() ua_ = da_ * inv_; ub_ = db_ * inv_; u2_ = d2_ * inv_;
#define GP6() o_.x = pkrtz_g(ua_, ub_); o_.z = o_.x;
#define GP7() o_.y = pk_f16_second(ua_ - x0f_g(ua_), ub_ - x0f_g(ub_));
#define GP8() if (cp_ <= L + 2) chk_ = fmaxf(fmaxf(chk_, inv_), bb_);   /* the norm-range test covers the columns dtw_mfma_kernel's does */
#define GP9(pos, wb) { const float x0_ = x0f_g(u2_); o_.w = __builtin_amdgcn_perm(0x3c000000u, pk_f16_second(x0_, u2_ - x0_), sel_one); \
                       *reinterpret_cast<u32x4 *>(ring + (wb) + (pos) * 1024 + lane * 16) = o_; }
#define RG_PRODUCE(pos, cbase, wb) do { GP0(pos, cbase) GP1() GP2() GP3() GP4() GP5() GP6() GP7() GP8() GP9(pos, wb) } while (0)
#define RG_AREF(cc, uu, GUARD)                                                                                                \
    {                                                                                                                         \
        const int sn = ((uu) + 1 + W) % NS, g = sn / 4, e = sn % 4;                                                           \
        int off = ((cc) + W - 1) * kRowBytes - (int)dl[e];                                                                    \
        if (GUARD) off = off < 0 ? 0 : off;                                                                                   \
        Areg[g] = *reinterpret_cast<const u32x4 *>(a_img + a_lane + (unsigned)off);                                           \
    }
#define RG_MFMA(g)                                                                                                            \
    do {                                                                                                                      \
        const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g]), __builtin_bit_cast(f16x8, bop), zero16, 0, 0, 0); \
    } while (0)
// column c = c0 + u (c0 - 1 a multiple of 12): a block boundary lies before the columns with c % 4 == 1, i.e. the barrier stands at the top of
// the steps with (1 + u) % 4 == 0; the steps with u % 4 == 0 build this wave's column of the block after next
#define RG_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        if ((1 + u) % 4 == 0) RP_GROUP_SYNC();                                                                                \
        bop = *reinterpret_cast<const u32x4 *>(ring + (r0 ^ ((((1 + u) / 4) & 1) * 4096u)) + ((1 + u) % 4) * 1024 + lane * 16);   /* column c + 1 */ \
        RG_AREF(c + 1, (u + 1) % NS, GUARD)                                                                                   \
        /* this step builds a column of the block after next: every wave in the steps with u % 4 == 0 (position ci), with two chunks per \
           workgroup also in those with u % 4 == 2 (position ci + 2) */                                                                        \
        const bool build_ = (u % 4 == 0) || (SH == 2 && u % 4 == 2);                                                          \
        const int ppos_ = (u % 4 == 0) ? ci : ci + 2;                                                                         \
        const unsigned pwb_ = r0 ^ (((1 + u / 4) & 1) * 4096u);                                                               \
        if (build_) { GP0(ppos_, (u % 4 == 0) ? c + 4 : c + 2) }                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        v2f up[2] = {(v2f){RP_INF, RP_INF}, (v2f){RP_INF, RP_INF}};                                                           \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                  \
                const v2f cost = (v2f){acc[sl / 4][4 * (sl % 4) + 2 * p], acc[sl / 4][4 * (sl % 4) + 2 * p + 1]};             \
                v2f m, v;                                                                                                     \
                m.x = fminf(fminf(up[p].x, Q[p][q + 1].x), Q[p][q].x);                                                        \
                m.y = fminf(fminf(up[p].y, Q[p][q + 1].y), Q[p][q].y);                                                        \
                v.x = cost.x + m.x; v.y = cost.y + m.y;                                                                       \
                if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                              \
                Q[p][q] = v;                                                                                                  \
                up[p] = v;                                                                                                    \
            }                                                                                                                 \
            if (build_) {                                                                                                     \
                if (q == (0 * B) / 10) { GP1() } if (q == (2 * B) / 10) { GP2() } if (q == (3 * B) / 10) { GP3() }            \
                if (q == (4 * B) / 10) { GP4() } if (q == (5 * B) / 10) { GP5() } if (q == (6 * B) / 10) { GP6() }            \
                if (q == (7 * B) / 10) { GP7() } if (q == (8 * B) / 10) { GP8() } if (q == (9 * B) / 10) { GP9(ppos_, pwb_) } \
            }                                                                                                                 \
            _Pragma("unroll") for (int g = 0; g < NTILE; ++g)                                                                 \
                if (group_last_use<W>(u, g) == q) RG_MFMA(g);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < NTILE; ++g)                                                                     \
            if (group_last_use<W>(u, g) < 0) RG_MFMA(g);                                                                      \
    } while (0)

        int cp_;
        float fa_, fb_, f2_, da_, db_, d2_, own_, bb_, inv_, ua_, ub_, u2_;
        u32x4 o_;
        // block 0 (columns 1 .. 4) into buffer 0; column 1's costs
        RG_PRODUCE(ci, 1, 0u);
        if (SH == 2) { RG_PRODUCE(ci + 2, 1, 0u); }
        __syncthreads();
        RG_AREF(1, 0, true)
        bop = *reinterpret_cast<const u32x4 *>(ring + lane * 16);
        RG_MFMA(0); RG_MFMA(1); RG_MFMA(2);
        __builtin_amdgcn_sched_barrier(0);
        int c0 = 1;
        {   // first block of twelve: cells of rows < 1 stay +inf (L >= 12)
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RG_STEP(true); }
        }
        r0 ^= 4096u;
        for (c0 = 1 + NS; c0 + NS - 1 <= L; c0 += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RG_STEP(false); }
            r0 ^= 4096u;
        }
        if (c0 <= L) {
#pragma unroll
            for (int u = 0; u < NS - 1; ++u) {  // the last L mod 12 columns
                const int c = c0 + u;
                if (c <= L) RG_STEP(false);
            }
        }
#undef RG_STEP
#undef RG_MFMA
#undef RG_AREF
#undef RG_PRODUCE
#undef GP0
#undef GP1
#undef GP2
#undef GP3
#undef GP4
#undef GP5
#undef GP6
#undef GP7
#undef GP8
#undef GP9

        // the range tests of the tile's waves meet (each saw a quarter of the columns)
        chk_lds[wave * 64 + lane] = chk_;
        __syncthreads();
        float chk_all = 0.f;
#pragma unroll
        for (int g = 0; g < SH; ++g) chk_all = fmaxf(chk_all, chk_lds[(tj * SH + g) * 64 + lane]);
        // D[m - 1][n] with m == n == L (dtw.rs:101): band position q = W - 2
        if (valid) {
            const size_t row = s * out_win_pitch + (size_t)w;
            const float denom = (float)(L + L);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int slot = 2 * NP * h + 2 * p + e;
                    if (slot < ch->count) {
                        const float cost = e ? Q[p][W - 2].y : Q[p][W - 2].x;
                        const float nc = cost / denom;
                        const float sc = 1.f / (1.f + expf((nc - score_ref) / score_ref));
                        const int t = ch->tid[slot];
                        if (t < T) scores[row * T + t] = sc;
                    }
                }
            }
            if (h == 0 && chk_all > kDtwFixLimit) dtw_fix_append(fix, row, (uint32_t)chunk_id);
        }
    }
}

bool dtw_mfma_group_supported(const TemplatesDev &t, int band, size_t n_win, size_t S, float score_ref) {
    // RP_DTW_GROUP (read per call): "0" = every chunk through dtw_mfma_kernel (the A/B switch of the bit-equality tests), "2" = also for
    // launches below the size rule (tests)
    const char *env = std::getenv("RP_DTW_GROUP");
    // the two-part f16 arithmetic only (RP_ARITH_FAST_SPLIT): four three-part A images (4 x 58 KB at 100 frames) do not fit a CU's LDS
    if ((env && env[0] == '0') || t.grp_count <= 0 || !t.grp_first || t.arith_mode() != kArithFastSplit) return false;
    if (!dtw_mfma_supported(t, band, n_win, false, 8, score_ref)) return false;
    if (env && env[0] == '2') return true;
    // hundreds of tile rounds, or the index hand-out leaves the chip waiting for the last workgroups
    const size_t tiles = (S * n_win + kGWin - 1) / kGWin;
    return tiles >= (size_t)64 * (size_t)device_cu_count();
}

// Scores the chunk groups t.grp_* (rp_ctx.cpp: runs of 4 class-2 chunks of one length); the caller sends the other chunks to dtw_mfma_kernel.
hipError_t launch_dtw_mfma_group(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch,
                                 size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores) {
    if (t.grp_count <= 0 || S == 0 || n_win == 0) return hipSuccess;
    if (!wk.fix) return hipErrorInvalidValue;
    dtw_mark(wk, kDtwRanMfmaGroup | kDtwRanF16x2);
    const size_t total_tiles = (S * n_win + kGWin - 1) / kGWin;
    // every group is a run of FOUR chunks of one length (rp_ctx.cpp builds no other shape: see below)
    {
        const int sh = 4, first = 0, count = t.grp_count;
        const size_t lds = dtw_mfma_group_lds_bytes(t.grp4_max_len, sh);
        size_t groups = (size_t)device_cu_count() / (size_t)count;
        if (groups < 1) groups = 1;
        const size_t n_tg = (total_tiles + (size_t)(kGWaves / sh) - 1) / (size_t)(kGWaves / sh);
        if (groups > n_tg) groups = n_tg;
        const size_t blocks = groups * (size_t)count;
#define RP_LAUNCH_GROUP(WW, SHH)                                                                                                    \
    do {                                                                                                                            \
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_mfma_group_kernel<WW, SHH>), 160 * 1024); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dtw_mfma_group_kernel<WW, SHH>), dim3((unsigned)blocks), dim3(64 * kGWaves), lds, st, mfcc, frame_pitch, frame_pitch, total_tiles, \
                           (unsigned)count, t.grp_first + first, first_win, n_win, out_win_pitch, t.chunks, reinterpret_cast<const uint4 *>(t.aimg), t.T, \
                           score_ref, scores, S, wk.fix);                                                                           \
    } while (0)
        // (the two-chunk shape, SH = 2, compiles and passes the same tests; measured 3.32 against 2.95 ms for dtw_mfma_kernel at 8 192 streams x
        // 16 templates -- half the sharing does not pay for the ring and the lockstep -- so rp_ctx.cpp builds no pairs and it is not instantiated)
        switch (band * 10 + sh) {
        case 34: RP_LAUNCH_GROUP(3, 4); break;
        case 44: RP_LAUNCH_GROUP(4, 4); break;
        case 54: RP_LAUNCH_GROUP(5, 4); break;
        default: return hipErrorNotSupported;
        }
#undef RP_LAUNCH_GROUP
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rp
