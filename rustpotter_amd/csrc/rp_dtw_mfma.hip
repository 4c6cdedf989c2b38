// rp_dtw_mfma.hip -- dtw_mfma_kernel: the banded DTW of mfcc_size 5 with the cosine costs on the matrix cores (DESIGN.md §4.2,
// round 3).  Same scoring as dtw_band_kernel (src/mfcc/dtw.rs:56-105 + comparator.rs:15-48 + normalizer.rs:17-29 +
// wakeword_comp.rs:22-37), other arithmetic for the cell cost:
//
//   * dtw_band_kernel spends 5 of the 8 issue slots of a band cell of a template pair on v_pk_fma_f32 for `1 - a.x` -- at the f32
//     FMA peak of the vector pipe (tools/scratch/valu_rate_probe.hip: 4.7 cycles per packed op, 4.2 per v_min3_f32).  Here the costs of
//     a whole band COLUMN come out of v_mfma_f32_32x32x16_f16 and the vector pipe only runs the recurrence.
//   * A wave owns 32 windows x one chunk of up to 8 same-length templates.  Lane l = (window l & 31, half h = l >> 5) runs the
//     recurrence of templates 4h..4h+3 (two packed pairs) of its window: the MFMA's C/D layout (col = lane & 31,
//     row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) puts exactly those costs into that lane, pairs in consecutive registers.
//   * Column-major sweep: step c takes window frame c against the 2W template rows of its band (rows c-W+1 .. c+W).  M = 8
//     templates x 12 circular row slots (row r lives in slot r mod 12) = 3 tiles of 32 rows, N = 32 windows, K = 16 f16 slots
//     = one instruction per tile.  A (the negated unit rows, split on the host, 256 B per template row in LDS) changes by one
//     row per template and column: one tile's operand is re-read per column.
//   * Precision: x = x0 + x1, a = a0 + a1 with x0 = rtz_f16(x), x1 = rtz_f16(x - x0) (22 significant bits; round 4: the template side rounds
//     both parts to nearest and carries the gain that undoes the window side's truncation, rp_device.h pk_f16_second); the slots hold
//     x0 a0, x1 a0, x0 a1 for the five components (15) and 1.0 x 1.0, accumulated in f32 on C = 0: the instruction leaves
//     1 - a.x with an error below 2^-20 (measured against the f32 CPU restatement: scores within 1e-6; the parity gate is 1e-5).
//     Lane (n, h) centres, scales and splits only components (0, 1) or (3, 4) and component 2; the two partial squared norms
//     meet through v_permlane32_swap.
//   * Software pipeline per column c: the A tile of column c+1 is re-read, the cells of column c run (two independent chains of
//     v_min3_f32 x2 + add x2 per cell), each tile's MFMA for column c+1 is issued right after the last cell that reads the tile, and
//     the frame of column c+2 is prepared in ten pieces between the cells.  Columns are unrolled 12 at a time so that every
//     slot, tile and band index is a compile-time register.
//   * Two arithmetics (P3, round 6; chosen by the context, rp_ctx_set_arithmetic).  P3 = true, the default (RP_ARITH_F32_MATRIX): f32-GRADE products.
//     Both operands are split into THREE bf16 parts, exactly (x0 = x & 0xffff0000, r = x - x0, x1 = r & 0xffff0000, x2 = r - x1: 3 x 8 significant
//     bits = an f32's 24; the template rows on the host, rounded to nearest), and the six partial products x_i a_j with i + j <= 2 of the five
//     components + 1.0 x 1.0 fill 31 of the 32 k-slots of TWO v_mfma_f32_32x32x16_bf16 chained on one accumulator (the second one band cell
//     after the first).  What is dropped (x1 a2 + x2 a1 + x2 a2) is below 2^-22 of a product, 2^-25.7 rms -- an f32 multiply rounds by up to
//     2^-24; tests/test_gpu_dtw_f64.py holds the scores to the strict-f32 oracle's own distance from an f64 evaluation.  The window side's two
//     operands are one run of six registers (the middle two shared), the A image is 512 bytes per template row (append_mfma_image3,
//     rp_ctx.cpp); 219 registers = two waves per SIMD, eight per workgroup (a twelve-wave build exists: RP_MFMA3_WAVES=12, see the launcher).  P3 = false (RP_ARITH_FAST_SPLIT, opt-in): the two-part f16 form
//     described above, 22-bit products.
//   * Two shapes (NT): eight template slots as described (chunks of 5..8 templates, band 3..5), or four (chunks of 3..4, band 5): a
//     tile is then 8 row slots x 4 templates, 16 circular row slots = 2 tiles, one template pair per lane, columns unrolled 16 at a time, twelve
//     waves per workgroup (eight where twelve waves' frame stages no longer fit beside the A image) in both arithmetics.  mfcc_size 13 / 16 have their own K axis: rp_dtw_mfma_wide3.hip / rp_dtw_mfma_wide.hip.
//   * BUILD: the 12 / 16-column blocks are `#pragma unroll` loops whose bodies exceed the compiler's budget for pragma-requested full
//     unrolling; this file is compiled with -mllvm -pragma-unroll-threshold=200000 (Makefile FILE_FLAGS_rp_dtw_mfma.hip).  Without it the
//     three-part four-slot build keeps a rolled loop, indexes its accumulators at run time and spills 13 709 values.
// Measured (tools/scratch/dtw_mfma_probe2.hip, 8 192 streams x 288 windows x 8 templates of 100 frames): 1.62 ms against 2.36 ms at
// dtw_band_kernel's C3 rate; VALU-issue bound (SQ_ACTIVE_INST_VALU = 100 % of the SIMD cycles), matrix pipe 21 % busy.  In the product
// at C3: 19.5 ms (vector kernels) -> 11.6-11.9 (two-part form) / 14.4-15.3 (three-part form, matrix pipe 39 % busy) (DESIGN.md §4.2,
// profiles/r06_final_*).
#include "rp_device.h"

#include <cstdlib>

namespace rp {

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));   // the window side's run of six registers (P3)

constexpr int kMK = 5;         // MFCC coefficients per frame
constexpr int kMWin = 32;      // windows per wave
// NT template slots per chunk: 8 (chunks of 5..8 templates; a tile of 32 rows = 4 row slots x 8 templates, 12 circular row slots = 3
// tiles, a lane runs two template pairs) or 4 (chunks of 3..4; a tile = 8 row slots x 4 templates, 16 row slots = 2 tiles, one pair)
constexpr int mfma_slots(int nt) { return nt == 8 ? 12 : 16; }
constexpr int mfma_tiles(int nt) { return nt == 8 ? 3 : 2; }
constexpr int kMSlotsMax = 16;  // rows of zero padding behind a chunk's A image
// accumulator register of (row slot, template pair p, pair element e) in its tile: the C/D layout puts row (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5) into `reg` of lane half lane >> 5.  NT = 8: row = 8 (slot % 4) + 4 h + (2 p + e); NT = 4: row = 8 (s >> 1) + 4 h +
// 2 (s & 1) + e with s = slot % 8.
constexpr int mfma_acc_reg(int nt, int slot, int p, int e) {
    return nt == 8 ? 4 * (slot % 4) + 2 * p + e : 4 * ((slot % 8) >> 1) + 2 * (slot & 1) + e;
}

__device__ __forceinline__ unsigned pkrtz(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
#ifdef RP_MFMA_CVT_BACK  // A/B build only: x0 as f32 by converting the f16 back (see RP_X0F)
__device__ __forceinline__ float lo_f32(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p)[0]; }
__device__ __forceinline__ float hi_f32(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p)[1]; }
#endif

// last band position q of column phase u whose MFMA row slot (u + q + NS - W + 2) mod NS lies in tile g; -1: the tile is not read
template <int W, int NT>
__host__ __device__ constexpr int mfma_last_use(int u, int g) {
    int last = -1;
    for (int q = 0; q < 2 * W; ++q)
        if (((u + q + mfma_slots(NT) - W + 2) % mfma_slots(NT)) / (32 / NT) == g) last = q;
    return last;
}

}  // namespace

// One workgroup = NW waves on one chunk (its A image is staged once); waves take the 32-entry tiles of the flattened
// (stream, window) space from an atomic counter.  GX = false: a tile's frames are staged in LDS, a tile may straddle two streams
// (n_win >= 32).  GX = true: lanes read their window's frames from global memory (live-stream batches: a few windows per
// stream; LIST mode of the averaged-template gate: list[] holds the rows that passed, *count of them).  list == nullptr with a
// count: DENSE mode of the gate -- the launch does nothing unless *count >= dense_min; LIST mode does nothing when the list is
// dense (rp_dtw.hip GateList).  abandon_nc < inf: early abandon of detect-only calls -- every 12 (16) columns a wave stops when the
// cheapest band cell of every (window, template) it holds is past abandon_nc * (m + n), writing score 0 (cell costs are >= 0 and
// every warping path crosses every column, so that cell bounds the final cost from below; the averaged template never stops).
// static_rounds: tiles a wave takes by its own index before it turns to the counter (mfma_static_rounds, rp_kernels.h).  agg_out != null:
// the chunk holds every sample template of the reference -- the kernel also writes ScoreMode::Max of a window's scores and raises the
// stream's hot flag (DtwFusedAgg, rp_kernels.h).
// RP_MFMA_TRACE (variant builds only, tools/r4_mfma_timeline.py): wave 0 and the last wave of every workgroup stamp the constant
// 100 MHz clock at the kernel's phase boundaries into the tail of the counter block (DtwWork::sched words 1024..4095, the first 96 workgroups: the
// list words of DtwWork::fix start right behind): where a short
// launch spends its fixed cost.  Never defined in the product.
#ifdef RP_MFMA_TRACE
#define RP_TRACE(slot)                                                                                                        \
    do {                                                                                                                      \
        if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == NW - 1) && blockIdx.x < 96)   /* 96 x 2 x 8 stamps end with the counter block */          \
            reinterpret_cast<unsigned long long *>(sched + 1024)[(blockIdx.x * 2 + ((threadIdx.x >> 6) ? 1 : 0)) * 8 + (slot)] = \
                __builtin_amdgcn_s_memrealtime();                                                                             \
    } while (0)
#else
#define RP_TRACE(slot)
#endif

#ifndef RP_MFMA_WAVES_PER_EU   // experiment builds: cap every instantiation at the registers of N waves per SIMD (3: 168), e.g. to leave room
#define RP_MFMA_WAVES_PER_EU 0 // for another kernel's waves beside an eight-wave workgroup
#endif
#if RP_MFMA_WAVES_PER_EU
#define RP_MFMA_OCC __attribute__((amdgpu_waves_per_eu(RP_MFMA_WAVES_PER_EU, RP_MFMA_WAVES_PER_EU)))
#else
#define RP_MFMA_OCC
#endif
template <int W, int NW, bool GX, int NT, bool P3>
RP_MFMA_OCC __global__ __launch_bounds__(64 * NW, 1) void dtw_mfma_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_frames_total, size_t total_tiles, unsigned n_chunks, int chunk_base,
    size_t first_win, size_t n_win, size_t out_win_pitch, const DtwChunk *__restrict__ chunks, const uint4 *__restrict__ aimg, int T,
    float score_ref, float *__restrict__ scores, float *__restrict__ avg, size_t n_streams, int max_len, const uint32_t *__restrict__ list,
    const uint32_t *__restrict__ count, uint32_t dense_min, float abandon_nc, uint32_t *__restrict__ sched, unsigned static_rounds,
    float *__restrict__ agg_out, uint32_t *__restrict__ agg_hot, float agg_threshold, uint32_t *__restrict__ fix) {
    constexpr int K = kMK, B = 2 * W, NS = mfma_slots(NT), NTILE = mfma_tiles(NT), SPT = 32 / NT, NP = NT / 4;
    constexpr int kRowBytes = P3 ? kDtwMfma3RowBytes : kDtwMfmaRowBytes;
#ifndef RP_MFMA_GX_PD  // A/B builds: 1 = the one-column look-ahead of the staged form
    constexpr int PD = GX ? (NS % 3 == 0 ? 3 : 4) : 1;  // columns a frame is requested ahead of its use (RP_P0)
#else
    constexpr int PD = GX ? RP_MFMA_GX_PD : 1;
#endif
    static_assert(NS % PD == 0, "the frame ring's slot must be a compile-time index");
    static_assert(NT == 8 || NT == 4, "template slots per chunk");
    static_assert(B + 2 <= NS, "the band and its two neighbours must fit the circular row slots");
    size_t total_entries = n_streams * n_win;
    if (list) {
        const uint32_t n_listed = *count;
        if (dense_min && n_listed >= dense_min) return;
        total_entries = n_listed;
        total_tiles = ((size_t)n_listed + kMWin - 1) / kMWin;
    } else if (count && *count < dense_min) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RP_TRACE(0);
    const unsigned ci = blockIdx.x % n_chunks;
    const unsigned n_groups = gridDim.x / n_chunks;
    const DtwChunk *ch = chunks + chunk_base + ci;
    const int L = ch->len;  // m == n == L
    const int a_bytes = (max_len + kMSlotsMax) * kRowBytes;
    const int xs_floats = dtw_mfma_stage_floats(max_len);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const u32x4 *asrc = reinterpret_cast<const u32x4 *>(aimg) + (P3 ? ch->aimg3_off : ch->aimg_off);
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < (L + kMSlotsMax) * kRowBytes / 16; i += 64 * NW) adst[i] = asrc[i];
    }
    __syncthreads();
    RP_TRACE(1);
    float *xs = reinterpret_cast<float *>(smem + a_bytes) + wave * xs_floats;
    (void)xs;
    const int n = lane & 31, h = lane >> 5;
    // A operand: this lane supplies row m = lane & 31 of a tile, k half = lane >> 5.  NT = 8: m = 8 jj + 4 h' + r' = (row slot jj of the
    // tile, template 4 h' + r'); NT = 4: m = 8 G + 4 h' + 2 sp + e = (row slot 2 G + sp, template 2 h' + e).
    const int mrow = lane & 31;
    const int jj = NT == 8 ? mrow >> 3 : 2 * (mrow >> 3) + ((mrow >> 1) & 1);
    const int tA = NT == 8 ? ((mrow >> 2) & 1) * 4 + (mrow & 3) : ((mrow >> 2) & 1) * 2 + (mrow & 1);
    const unsigned a_lane = (unsigned)(h * 128 + tA * 16);
    unsigned dl[SPT];  // byte offset back to the row this lane's slot holds when the newest row sits in slot e of its tile
#pragma unroll
    for (int e = 0; e < SPT; ++e) dl[e] = (unsigned)(((e - jj + NS) % NS) * kRowBytes);
    const unsigned sel_one = h ? 0x07060100u : 0x03020100u;  // slot 7: x1 of component 2 (half 0) / the constant 1.0 (half 1)
    // P3 (three bf16 parts): register 3 of the first operand holds (x0, x1) of component 2 in half 0 and (x2, x0) in half 1 -- one v_perm
    // of (t, x0) with a per-half selector, t = r1 - (r1 & mask_c2): r1 itself in half 0 (its upper 16 bits ARE x1), x2 in half 1
    const unsigned sel_c2 = h ? 0x03020706u : 0x07060302u;
    const unsigned mask_c2 = h ? 0xffff0000u : 0u;
    (void)sel_c2; (void)mask_c2;
    const float abandon_cost = abandon_nc * (float)(L + L);
    // which of this lane's templates (NT / 2 of them) can keep a wave alive: real ones; the averaged template (tid >= T) always does
    bool slot_real[2 * NP], slot_avg[2 * NP];
#pragma unroll
    for (int e = 0; e < 2 * NP; ++e) {
        slot_real[e] = 2 * NP * h + e < ch->count;
        slot_avg[e] = slot_real[e] && ch->tid[2 * NP * h + e] >= T;
    }

    // Tiles are handed out by an atomic counter per chunk (sched[2 ci]): a small batch is a few tiles per wave, and a static
    // split leaves most of the chip waiting for the waves that got one tile more (BASELINE config C2: 3.09 tiles per wave).
    // The last workgroup of a chunk to finish (sched[2 ci + 1] counts them) puts both words back to zero for the next launch.
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
            tile = (size_t)round * chunk_waves + (size_t)(blockIdx.x / n_chunks) * NW + (size_t)wave;
            ++round;
        } else {
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(next_tile, 1u);
            tile = (size_t)__builtin_amdgcn_readfirstlane(ticket) + (size_t)static_rounds * chunk_waves;
        }
        if (tile >= total_tiles) break;
        // ---- lanes -> (stream, window) ----
        const size_t f0 = tile * kMWin;
        bool valid;
        size_t s;
        int w;
        const float *xw;
        if (GX) {
            size_t f = f0 + n;
            valid = f < total_entries;
            if (list) f = list[valid ? f : total_entries - 1];  // row ids s * n_win + w of the windows that passed the gate
            s = valid ? f / n_win : 0;
            w = valid ? (int)(f - s * n_win) : 0;
            xw = mfcc + (s * frame_pitch + first_win + (size_t)w) * K;  // the caller leaves W * K floats of slack after the last frame
        } else {
            // stage the frames of up to two stream segments (columns L + 1 .. L + 3 are read ahead, never used)
            const size_t sA = f0 / n_win;
            const int wA = (int)(f0 - sA * n_win);
            const int nA = (int)n_win - wA < kMWin ? (int)n_win - wA : kMWin;
            const int nB = (nA < kMWin && sA + 1 < n_streams) ? kMWin - nA : 0;
            const int segA = nA + L + 3;
#ifndef RP_AB_STAGE1   // A/B builds: the one-load-per-wait staging loop of round 3
            // four loads in flight per wait: left one by one, a tile's ~11 loads per lane were as many L2 round trips -- nothing covers
            // them in the first round of a short launch, where every wave of the chip stages at the same time
            auto stage = [&](const float *src, size_t g0, int n_floats, float *dst) {
                for (int i0 = lane; i0 < n_floats; i0 += 256) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = i0 + 64 * j;
                        v[j] = (i < n_floats && g0 + (size_t)(i / K) < n_frames_total) ? src[g0 * K + i] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i0 + 64 * j < n_floats) dst[i0 + 64 * j] = v[j];
                }
            };
            stage(mfcc + sA * frame_pitch * K, first_win + wA, segA * K, xs);
            if (nB > 0) stage(mfcc + (sA + 1) * frame_pitch * K, first_win, (nB + L + 3) * K, xs + segA * K);
#else
            {
                const float *src = mfcc + sA * frame_pitch * K;
                const size_t g0 = first_win + wA;
                for (int i = lane; i < segA * K; i += 64) {
                    const int f = i / K;
                    xs[i] = g0 + f < n_frames_total ? src[g0 * K + i] : 0.f;
                }
            }
            if (nB > 0) {
                const float *src = mfcc + (sA + 1) * frame_pitch * K;
                const int segB = nB + L + 3;
                for (int i = lane; i < segB * K; i += 64) {
                    const int f = i / K;
                    xs[segA * K + i] = first_win + f < n_frames_total ? src[first_win * K + i] : 0.f;
                }
            }
#endif
            wave_lds_sync();
            const bool inA = n < nA;
            valid = inA || (n - nA < nB);
            s = inA ? sA : sA + 1;
            w = inA ? wA + n : n - nA;
            xw = xs + (inA ? n : (valid ? segA + n - nA : 0)) * K;
        }
        const float *xa = xw + (h ? 3 : 0);  // this half's two components; component 2 at xw + 2
        const float *x2 = xw + 2;
        // MfccNormalizer::normalize, src/mfcc/normalizer.rs:17-29: sequential column sums (of this lane's three components)
        float mua = 0.f, mub = 0.f, mu2 = 0.f;
#ifndef RP_MFMA_GX_MEAN_UNROLL
#define RP_MFMA_GX_MEAN_UNROLL 20
#endif
#ifndef RP_MFMA_MEAN_UNROLL
#define RP_MFMA_MEAN_UNROLL 8
#endif
        constexpr int kMeanUnroll = GX ? RP_MFMA_GX_MEAN_UNROLL : RP_MFMA_MEAN_UNROLL;  // from global memory: 20 frames in flight per wait (an L2 round trip each), sums in the same order
#pragma unroll kMeanUnroll
        for (int i = 0; i < L; ++i) { mua += xa[i * K]; mub += xa[i * K + 1]; mu2 += x2[i * K]; }
        mua = mua / (float)L; mub = mub / (float)L; mu2 = mu2 / (float)L;
        if (round == 1) RP_TRACE(2);

        // Q[p][q] = D[(c - 1) - W + 1 + q][c - 1] of the template pair p (band position, as P[] of dtw_band_kernel with rows and
        // columns swapped); column 0: D[0][0] = 0 sits at q = W - 1.  Q[p][B] stays +inf (the cell below the band).
        v2f Q[NP][B + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[p][q] = (v2f){RP_INF, RP_INF};
            Q[p][W - 1] = (v2f){0.f, 0.f};
        }
        u32x4 Areg[NTILE], Areg2[P3 ? NTILE : 1];   // P3: the row's second 256 bytes = the A operand of the second k-step
#pragma unroll
        for (int g = 0; g < NTILE; ++g) {
            const int slot = SPT * g + jj;
            int r = W - ((W - slot + NS) % NS);  // 1-based template row in this slot for the state "newest row = W"
            r = r < 1 ? 1 : r;
            Areg[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)(r - 1) * kRowBytes);
            if (P3) Areg2[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + 256u + (unsigned)(r - 1) * kRowBytes);
        }
        v16f acc[NTILE];   // costs of the current column; a tile is refilled for the next column as soon as its last cell is done
        u32x4 bop[2];  // B operand of column cc in bop[cc & 1]: built two columns ahead, in pieces between the cells
        // P3: the two k-steps' B operands as ONE run of six registers per column, [0..3] the first k-step's and [2..5] the second's: the
        // registers both need -- (x0a, x0b) and (x1a, x1b) -- sit in the middle and are written once (slot order: append_mfma_image3,
        // rp_ctx.cpp).  Measured neutral against two separate operands with three copies per column (15.0 ms either way), ten registers fewer.
        u32x6 bv[P3 ? 2 : 1];
        (void)bv;

// The frame work of column cc, cut into ten pieces P0..P9 that are placed between the cells of the recurrence.
// P0 requests the frame of column cc into ring slot rs, P1 takes it out PD columns later (rs = cc mod PD, spelled out by the caller:
// c0 - 1 is a multiple of NS and PD divides NS, so the slot is a compile-time register).  LDS-staged tiles look one column ahead;
// frames from global memory (GX) three or four: at one column (~0.6 us of a SIMD shared by three waves) the loads of the
// L2 / Infinity Cache (> 1 us under load) were what a live-stream launch waited for.
#define RP_P0(cc, rs) fa_[rs] = xa[((cc) - 1) * K]; fb_[rs] = xa[((cc) - 1) * K + 1]; f2_[rs] = x2[((cc) - 1) * K];
#define RP_P1(cc, rs) da_ = fa_[rs] - mua; db_ = fb_[rs] - mub; d2_ = f2_[rs] - mu2;
#define RP_P2(cc) own_ = fmaf(da_, da_, db_ * db_);
#define RP_P3(cc) { const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(own_), __float_as_uint(own_), false, false); \
                    bb_ = fmaf(d2_, d2_, __uint_as_float(sw_[0]) + __uint_as_float(sw_[1])); }
#define RP_P4(cc) inv_ = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;  /* zero frame -> zero vector -> cost 1 (comparator.rs:43-47) */
#define RP_P5(cc) ua_ = da_ * inv_; ub_ = db_ * inv_; u2_ = d2_ * inv_;
// P3: x = x0 + x1 + x2 EXACTLY with x0 = x & 0xffff0000 (its first 8 significant bits = a bf16), r1 = x - x0 (exact), x1 = r1 & 0xffff0000,
// x2 = r1 - x1 (exact, at most 8 significant bits: a bf16 too).  A register of the B operand is the upper halves of two f32 values: one v_perm.
#define RP_HI2(hi, lo) __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u)
#define RP_AND(x, m) __uint_as_float(__float_as_uint(x) & (m))
#define RP_BSET(par, i, v) bv[par][i] = (v)
#define RP_P6(cc, par) if (P3) { x0a_ = RP_AND(ua_, 0xffff0000u); x0b_ = RP_AND(ub_, 0xffff0000u); x0c_ = RP_AND(u2_, 0xffff0000u); RP_BSET(par, 2, RP_HI2(ub_, ua_)); } \
                       else { bop[par].x = pkrtz(ua_, ub_); bop[par].z = bop[par].x; }
// x1 = rtz_f16(x - x0): x0 as f32 is x with the low 13 mantissa bits cleared (one full-rate v_and instead of a half-rate v_cvt_f32_f16;
// below the f16 normal range, |x| < 6.1e-5, the two differ by less than the f16 subnormal spacing 6e-8 -- far below the kernel's error)
#ifndef RP_MFMA_CVT_BACK
#define RP_X0F(x, packed, hi) __uint_as_float(__float_as_uint(x) & 0xffffe000u)
#else
#define RP_X0F(x, packed, hi) ((hi) ? hi_f32(packed) : lo_f32(packed))
#endif
#define RP_P7(cc, par) if (P3) { r1a_ = ua_ - x0a_; r1b_ = ub_ - x0b_; r1c_ = u2_ - x0c_; RP_BSET(par, 3, RP_HI2(r1b_, r1a_)); } \
                       else { bop[par].y = pk_f16_second(ua_ - RP_X0F(ua_, bop[par].x, 0), ub_ - RP_X0F(ub_, bop[par].x, 1)); }
#ifndef RP_AB_NO_RANGE_TEST   /* A/B builds only (results wrong for out-of-range frames): what the test costs */
#define RP_P8(cc) chk_ = fmaxf(fmaxf(chk_, inv_), bb_);  /* one v_max3_f32: the norm-range test (kDtwFixLimit, rp_kernels.h) */ \
                  if (P3) { x1a_ = RP_AND(r1a_, 0xffff0000u); x1b_ = RP_AND(r1b_, 0xffff0000u); tc_ = RP_AND(r1c_, mask_c2); }
#else
#define RP_P8(cc) if (P3) { x1a_ = RP_AND(r1a_, 0xffff0000u); x1b_ = RP_AND(r1b_, 0xffff0000u); tc_ = RP_AND(r1c_, mask_c2); }
#endif
#define RP_P9(cc, par) if (P3) { RP_BSET(par, 0, RP_HI2(r1b_ - x1b_, r1a_ - x1a_)); RP_BSET(par, 1, __builtin_amdgcn_perm(__float_as_uint(r1c_ - tc_), __float_as_uint(u2_), sel_c2)); } \
                       else { const float x0_ = RP_X0F(u2_, pkrtz(u2_, 0.f), 0); /* (x0, x1) of component 2: x0 is already an f16 value, x1 rounds to nearest */ \
                         bop[par].w = __builtin_amdgcn_perm(0x3c000000u, pk_f16_second(x0_, u2_ - x0_), sel_one); }
// P3: the second k-step's operand: (x0, x0 | x1, x1 | x0, x0) of the half's two components against (a1, a1 | a1, a1 | a2, a2), and register 3 =
// (x0, x1) of component 2 against (a1, a1) in half 0, the constant (1.0, 0) in half 1
#define RP_P10(cc, par) if (P3) { bv[par][4] = bv[par][2]; bv[par][5] = h ? 0x00003f80u : bv[par][1]; }
#define RP_PREP_ALL(cc, par) RP_P0(cc, (cc) % PD) RP_P1(cc, (cc) % PD) RP_P2(cc) RP_P3(cc) RP_P4(cc) RP_P5(cc) RP_P6(cc, par) RP_P7(cc, par) RP_P8(cc) RP_P9(cc, par) RP_P10(cc, par)
// the A tile that receives template row cc + W (cc = 1 + uu mod 12)
#define RP_AREF(cc, uu, GUARD)                                                                                                \
    {                                                                                                                         \
        const int sn = ((uu) + 1 + W) % NS, g = sn / SPT, e = sn % SPT;                                                       \
        int off = ((cc) + W - 1) * kRowBytes - (int)dl[e];                                                                    \
        if (GUARD) off = off < 0 ? 0 : off;                                                                                   \
        Areg[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)off);                                            \
        if (P3) Areg2[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + 256u + (unsigned)off);                            \
    }
#define RP_B1(par) __builtin_shufflevector(bv[P3 ? (par) : 0], bv[P3 ? (par) : 0], 0, 1, 2, 3)
#define RP_B2(par) __builtin_shufflevector(bv[P3 ? (par) : 0], bv[P3 ? (par) : 0], 2, 3, 4, 5)
#define RP_MFMA(g, par)                                                                                                       \
    do {                                                                                                                      \
        const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  \
        if (P3) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g]), __builtin_bit_cast(bf16x8, RP_B1(par)), zero16, 0, 0, 0); \
        else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g]), __builtin_bit_cast(f16x8, bop[par]), zero16, 0, 0, 0); \
    } while (0)
// P3: the second k-step on the same accumulator, one band cell after the first (its eight passes are over by then)
#define RP_MFMA2(g, par)                                                                                                      \
    do {                                                                                                                      \
        if (P3) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg2[g]), __builtin_bit_cast(bf16x8, RP_B2(par)), acc[g], 0, 0, 0); \
    } while (0)

#ifndef RP_P3_GAP   // band cells between a tile's two k-steps (A/B builds)
#define RP_P3_GAP 1
#endif
// column c (c = 1 + u mod 12): rows r_q = c - W + 1 + q, q = 0..2W-1, sit in MFMA row slot (u + q + 14 - W) mod 12
#define RP_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        RP_AREF(c + 1, (u + 1) % NS, GUARD)                                                                                   \
        v2f up[2] = {(v2f){RP_INF, RP_INF}, (v2f){RP_INF, RP_INF}};                                                           \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) { /* NT = 8: two independent chains, interleaved */                \
                const v2f cost = (v2f){acc[sl / SPT][mfma_acc_reg(NT, sl, p, 0)], acc[sl / SPT][mfma_acc_reg(NT, sl, p, 1)]}; \
                v2f m, v;                                                                                                     \
                m.x = fminf(fminf(up[p].x, Q[p][q + 1].x), Q[p][q].x);                                                        \
                m.y = fminf(fminf(up[p].y, Q[p][q + 1].y), Q[p][q].y);                                                        \
                v.x = cost.x + m.x; v.y = cost.y + m.y; /* two plain adds (2.4 cycles each) beat v_pk_add_f32 (4.7 + a wait state) */ \
                if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                              \
                Q[p][q] = v;                                                                                                  \
                up[p] = v;                                                                                                    \
            }                                                                                                                 \
            /* piece k of the frame of column c + 2 after cell (k B) / 10 (its values were requested one column earlier, P0): the  \
               pieces fill the wait states between a cell's adds and the next cell's v_min3 */                                                                                  \
            if (!P3) {                                                                                                        \
            if (q == (0 * B) / 10) { RP_P1(c + 2, (u + 3) % PD) RP_P0(c + 2 + PD, (u + 3) % PD) } if (q == (2 * B) / 10) { RP_P2(c + 2) }                       \
            if (q == (3 * B) / 10) { RP_P3(c + 2) } if (q == (4 * B) / 10) { RP_P4(c + 2) } if (q == (5 * B) / 10) { RP_P5(c + 2) } \
            if (q == (6 * B) / 10) { RP_P6(c + 2, (u + 1) & 1) } if (q == (7 * B) / 10) { RP_P7(c + 2, (u + 1) & 1) }         \
            if (q == (8 * B) / 10) { RP_P8(c + 2) } if (q == (9 * B) / 10) { RP_P9(c + 2, (u + 1) & 1) }                      \
            } else { /* eleven pieces: the three-part split is twenty instructions against ten */                            \
            if (q == (0 * B) / 11) { RP_P1(c + 2, (u + 3) % PD) RP_P0(c + 2 + PD, (u + 3) % PD) } if (q == (1 * B) / 11) { RP_P2(c + 2) }                       \
            if (q == (2 * B) / 11) { RP_P3(c + 2) } if (q == (3 * B) / 11) { RP_P4(c + 2) } if (q == (4 * B) / 11) { RP_P5(c + 2) } \
            if (q == (5 * B) / 11) { RP_P6(c + 2, (u + 1) & 1) } if (q == (6 * B) / 11) { RP_P7(c + 2, (u + 1) & 1) }         \
            if (q == (7 * B) / 11) { RP_P8(c + 2) } if (q == (8 * B) / 11) { RP_P9(c + 2, (u + 1) & 1) }                      \
            if (q == (9 * B) / 11) { RP_P10(c + 2, (u + 1) & 1) }                                                             \
            }                                                                                                                 \
            _Pragma("unroll") for (int g = 0; g < NTILE; ++g) {                                                               \
                if (mfma_last_use<W, NT>(u, g) == q) RP_MFMA(g, u & 1);                                                       \
                if (q >= RP_P3_GAP && mfma_last_use<W, NT>(u, g) == q - RP_P3_GAP) RP_MFMA2(g, u & 1);                        \
            }                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < NTILE; ++g) {                                                                   \
            if (mfma_last_use<W, NT>(u, g) < 0) { RP_MFMA(g, u & 1); RP_MFMA2(g, u & 1); }                                    \
            else if (RP_P3_GAP > 0 && mfma_last_use<W, NT>(u, g) > B - 1 - RP_P3_GAP) RP_MFMA2(g, u & 1);                     \
        }                                                                                                                     \
    } while (0)

        float fa_[PD], fb_[PD], f2_[PD], da_, db_, d2_, own_, bb_, inv_, ua_, ub_, u2_, chk_ = 0.f;
        float x0a_ = 0.f, x0b_ = 0.f, x0c_ = 0.f, r1a_ = 0.f, r1b_ = 0.f, r1c_ = 0.f, x1a_ = 0.f, x1b_ = 0.f, tc_ = 0.f;   // P3 only
        (void)x0a_; (void)x0b_; (void)x0c_; (void)r1a_; (void)r1b_; (void)r1c_; (void)x1a_; (void)x1b_; (void)tc_;
        RP_AREF(1, 0, true)
        RP_PREP_ALL(1, 1)
        RP_MFMA(0, 1); RP_MFMA(1, 1);
        if (NTILE > 2) RP_MFMA(NTILE - 1, 1);
        RP_MFMA2(0, 1); RP_MFMA2(1, 1);
        if (NTILE > 2) RP_MFMA2(NTILE - 1, 1);
        RP_PREP_ALL(2, 0)
#pragma unroll
        for (int a = 0; a < PD; ++a) { RP_P0(3 + a, (3 + a) % PD) }
        __builtin_amdgcn_sched_barrier(0);
        int c0 = 1;
        bool dead = false;
        {   // first block: cells of rows < 1 stay +inf (L >= NS)
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(true); }
        }
// early abandon: wave-uniform, once per 12 columns.  (RP_MFMA_PRICE_NO_ABANDON: tools/isa_mix.py prices the hot loop as the headline call
// runs it -- abandon_nc = +inf jumps over this block with one scalar branch -- by compiling the block out; never defined in the product.)
#ifdef RP_MFMA_PRICE_NO_ABANDON
#define RP_ABANDON_ON false
#else
#define RP_ABANDON_ON (abandon_nc < RP_INF)
#endif
#define RP_ABANDON_CHECK()                                                                                                    \
    if (RP_ABANDON_ON) {                                                                                                \
        bool alive = false;                                                                                                   \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                      \
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
                for (int u = 0; u < NS - 1; ++u) {  // the last L mod 12 columns
                    const int c = c0 + u;
                    if (c <= L) RP_STEP(false);
                }
            }
        }
#undef RP_ABANDON_CHECK
#undef RP_ABANDON_ON
#undef RP_STEP
#undef RP_MFMA
#undef RP_MFMA2
#undef RP_P10
#undef RP_HI2
#undef RP_AND
#undef RP_BSET
#undef RP_B1
#undef RP_B2
#undef RP_AREF
#undef RP_PREP_ALL
#undef RP_P0
#undef RP_P1
#undef RP_P2
#undef RP_P3
#undef RP_P4
#undef RP_P5
#undef RP_P6
#undef RP_P7
#undef RP_P8
#undef RP_P9
#undef RP_X0F

        if (round == 1) RP_TRACE(3);
        // D[m - 1][n] with m == n == L (dtw.rs:101): band position q = (L - 1) - (L - W + 1) = W - 2
        float best = 0.f;  // ScoreMode::Max over this lane's templates (scores are > 0; an abandoned wave reports 0 like its scores)
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
                        const float sc = dead ? 0.f : 1.f / (1.f + expf((nc - score_ref) / score_ref));
                        const int t = ch->tid[slot];
                        if (t < T) { scores[row * T + t] = sc; best = fmaxf(best, sc); }
                        else if (!dead) avg[row] = sc;
                    }
                }
            }
            // a frame outside the norm range (both lane halves saw the same squared norms): listed for dtw_ref_kernel
            if (h == 0 && chk_ > kDtwFixLimit) dtw_fix_append(fix, row, (uint32_t)(chunk_base + (int)ci));
        }
        if (agg_out) {  // the chunk holds every sample template (launch_dtw): the two lanes of a window hold all its scores
            const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
            const float m = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
            if (valid && h == 0) {
                agg_out[s * out_win_pitch + (size_t)w] = m;
                if (agg_hot && m > agg_threshold) agg_hot[s] = 1u;  // as agg_store: every writer stores the same value
            }
        }
        if (!GX) wave_lds_sync();  // the next tile restages xs
        if (round == 1) RP_TRACE(4);
    }
    RP_TRACE(5);
    __syncthreads();
    RP_TRACE(6);
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(next_tile + 1, 1u) == n_groups - 1) {  // every workgroup of this chunk has taken its last ticket
            next_tile[0] = 0;
            next_tile[1] = 0;
        }
    }
}

bool dtw_mfma_supported(const TemplatesDev &t, int band, size_t n_win, bool from_global, int slots, float score_ref) {
    const int mode = t.arith_mode();   // the context's arithmetic (rp_ctx_set_arithmetic), read per call
    if (mode == kArithStrictF32 || t.K != kMK || !(mode == kArithFastSplit ? t.aimg : t.aimg3) || t.max_diff != 0) return false;
    const bool p3 = mode != kArithFastSplit;
    // the two-part form: the score's sensitivity to the cost grows like 1 / score_ref (rp_kernels.h); the three-part form's products are
    // f32-grade, it needs no floor
    if (!p3 && !(score_ref >= kDtwMfmaMinScoreRef)) return false;
    if (slots == 8 ? (band < 3 || band > 5) : band != 5) return false;  // 12 (16) row slots hold 2 band + 2 rows; 4 slots: band 5 only
    if (!from_global && n_win < (size_t)kMWin) return false;            // a staged tile holds at most two stream segments
    // the first 12 (16) columns are one guarded block
    if (slots == 8 ? t.mfma_min_len < mfma_slots(8) : t.mfma_min_len4 < mfma_slots(4)) return false;
    // long templates: the A image leaves room for eight waves' frame stages only
    return dtw_mfma_lds_bytes(t.max_len, 8, p3 ? kDtwMfma3RowBytes : kDtwMfmaRowBytes) <= 160 * 1024;
}

hipError_t launch_dtw_mfma(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, int slots, int chunk_base, int n_chunks, const float *mfcc, size_t S,
                           size_t frame_pitch, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg,
                           bool from_global, const uint32_t *list, const uint32_t *count, uint32_t dense_min, float abandon_nc,
                           const DtwFusedAgg *fuse) {
    if (n_chunks <= 0 || S == 0 || n_win == 0) return hipSuccess;
    const bool p3 = t.arith_mode() != kArithFastSplit;
    dtw_mark(wk, kDtwRanMfma | (p3 ? kDtwRanBf16x3 : kDtwRanF16x2));
    if (fuse && (n_chunks != 1 || list || count)) return hipErrorInvalidValue;  // launch_dtw only asks for it with one chunk, every row scored
    float *agg_out = fuse ? fuse->agg : nullptr;
    uint32_t *agg_hot = fuse ? fuse->hot : nullptr;
    const float agg_threshold = fuse ? fuse->threshold : 0.f;
    if (list && !from_global) return hipErrorNotSupported;
    const size_t total_tiles = (S * n_win + kMWin - 1) / kMWin;
    const int row_bytes = p3 ? kDtwMfma3RowBytes : kDtwMfmaRowBytes;
    int nw = dtw_mfma_lds_bytes(t.max_len, 12, row_bytes) <= 160 * 1024 ? 12 : 8;
    if (p3 && slots == 8) {
        // the three-part form runs two waves per SIMD (8 per workgroup, 219 registers, nothing spilled).  Its twelve-wave build (168 registers,
        // 49 values spilled) is 1.3 % faster at BASELINE C3 -- 14.93-14.94 ms against 15.11-15.21, alternated three times -- and pays for it with
        // three scratch stores and three reloads per 12-column block that reach the HBM: 5.7 GB per launch against 1.47 (the algorithmic bytes
        // are 1.17 GB).  Harmless for the time (0.38 TB/s), but it is waste on the one counter this path is judged against: not the default.
        // RP_MFMA3_WAVES=12 selects it (same bits)
        static const int env_nw = [] { const char *e = std::getenv("RP_MFMA3_WAVES"); return e ? std::atoi(e) : 0; }();
        if (env_nw != 12) nw = 8;
    }
    const size_t lds = dtw_mfma_lds_bytes(t.max_len, nw, row_bytes);
    const void *image = p3 ? t.aimg3 : t.aimg;
    // one workgroup per CU and chunk group; the waves take tiles from the chunk's counter
    if (!wk.sched || !wk.fix) return hipErrorInvalidValue;
    size_t groups = (size_t)device_cu_count() / (size_t)n_chunks;
    if (groups < 1) groups = 1;
    const size_t need = (total_tiles + nw - 1) / nw;
    if (groups > need) groups = need;
    const size_t blocks = groups * (size_t)n_chunks;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const unsigned static_rounds = mfma_static_rounds(total_tiles, groups * (size_t)nw, list != nullptr);
#define RP_LAUNCH_MFMA_P(WW, NW, GXV, NT, PP)                                                                                       \
    do {                                                                                                                            \
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(dtw_mfma_kernel<WW, NW, GXV, NT, PP>), 160 * 1024); e != hipSuccess) return e; \
        hipLaunchKernelGGL((dtw_mfma_kernel<WW, NW, GXV, NT, PP>), dim3((unsigned)blocks), dim3(64 * NW), lds, st, mfcc, frame_pitch, frame_pitch, \
                           total_tiles, (unsigned)n_chunks, chunk_base, first_win, n_win, out_win_pitch, t.chunks,                   \
                           reinterpret_cast<const uint4 *>(image), t.T, score_ref, scores, avg, S, t.max_len, list, count, dense_min, \
                           abandon_nc, wk.sched, static_rounds, agg_out, agg_hot, agg_threshold, wk.fix);                                            \
    } while (0)
#define RP_LAUNCH_MFMA(WW, NW, GXV, NT)                                                                                             \
    do {                                                                                                                            \
        if (p3) RP_LAUNCH_MFMA_P(WW, NW, GXV, NT, true); else RP_LAUNCH_MFMA_P(WW, NW, GXV, NT, false);                             \
    } while (0)
#define RP_LAUNCH_MFMA_W(WW, NT)                                                                                                    \
    do {                                                                                                                            \
        if (from_global) { if (nw == 12) RP_LAUNCH_MFMA(WW, 12, true, NT); else RP_LAUNCH_MFMA(WW, 8, true, NT); }                  \
        else { if (nw == 12) RP_LAUNCH_MFMA(WW, 12, false, NT); else RP_LAUNCH_MFMA(WW, 8, false, NT); }                            \
    } while (0)
    if (slots == 4) {
        if (band != 5) return hipErrorNotSupported;
        RP_LAUNCH_MFMA_W(5, 4);
    } else {
        switch (band) {
        case 3: RP_LAUNCH_MFMA_W(3, 8); break;
        case 4: RP_LAUNCH_MFMA_W(4, 8); break;
        case 5: RP_LAUNCH_MFMA_W(5, 8); break;
        default: return hipErrorNotSupported;
        }
    }
#undef RP_LAUNCH_MFMA_W
#undef RP_LAUNCH_MFMA
#undef RP_LAUNCH_MFMA_P
    return hipGetLastError();
}

}  // namespace rp
