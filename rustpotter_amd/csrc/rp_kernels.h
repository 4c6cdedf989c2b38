// rp_kernels.h -- launchers of the gfx950 kernels (rp_mfcc / rp_dtw / rp_scan / rp_resample / rp_frontend / rp_mlp .hip) used by the
// host-side mirror (rp_detector.cpp) and the C ABI (rp_capi.cpp).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rp {

constexpr int kFrame = 480;   // src/constants.rs:2 @16 kHz
constexpr int kShift = 160;   // src/constants.rs:8
constexpr int kBins = 240;    // src/mfcc/extractor.rs:28

// Device-resident constant tables of the MFCC pipeline for one mfcc_size K.
// mfcc_size 5 / 16 (K1 = 6 / 17 filters): can filter f be non-zero on a bin of the 16-bin group k2 (bins l + 16*k2,
// l = 0..15), or -- mirror -- on a bin 240 - (l + 16*k2)?  Triangle f is non-zero on bins (c[f], c[f+2]) exclusive;
// the centres are what new_mel_filter_bank (src/mfcc/extractor.rs:164-198) yields for these sizes (checked against
// the table on upload, Ctx::tables_for).
template <int K1T> struct MelCentres;
template <> struct MelCentres<6> { static constexpr int c[8] = {0, 9, 22, 41, 68, 106, 161, 240}; };
template <> struct MelCentres<14> { static constexpr int c[16] = {0, 4, 8, 14, 20, 28, 37, 47, 60, 74, 92, 112, 137, 166, 200, 240}; };
template <> struct MelCentres<17> {
    static constexpr int c[19] = {0, 3, 7, 11, 16, 21, 28, 35, 43, 53, 64, 77, 92, 109, 128, 150, 176, 206, 240};
};
template <int K1T> __host__ __device__ constexpr bool mel_touches(int f, int k2, bool mirror) {
    if (f < 0 || f >= K1T) return false;
    const int lo = MelCentres<K1T>::c[f] + 1, hi = MelCentres<K1T>::c[f + 2] - 1;  // non-zero bins lo..hi
    const int g_lo = mirror ? 240 - (16 * k2 + 15) : 16 * k2;                        // bins the group covers
    const int g_hi = mirror ? 240 - 16 * k2 : 16 * k2 + 15;
    return g_hi >= lo && g_lo <= hi;
}

// The sparse mel tables of the mfcc_size 5 / 16 kernels: per lane l (= bin residue mod 16) the weights of exactly the
// (filter, 16-bin group, direct / mirror) triples mel_touches<K1T> admits, in the order the kernel accumulates them
// (filter ascending, group ascending, direct before mirror) -- one contiguous row of kMelRowPitch<K1T> floats per lane,
// read with 16-byte LDS loads (a pitch of 36 / 44 floats keeps the 16 lanes of a frame on distinct banks).
template <int K1T> __host__ __device__ constexpr int mel_index(int f, int k2, bool mirror) {
    int n = 0;
    for (int ff = 0; ff < K1T; ++ff)
        for (int kk = 0; kk < 8; ++kk) {
            if (ff == f && kk == k2 && !mirror) return n;
            if (mel_touches<K1T>(ff, kk, false)) ++n;
            if (ff == f && kk == k2 && mirror) return n;
            if (mel_touches<K1T>(ff, kk, true)) ++n;
        }
    return n;  // f == K1T: the number of entries
}
template <int K1T> constexpr int kMelEntries = mel_index<K1T>(K1T, 0, false);
template <int K1T> constexpr int kMelRowPitch = ((kMelEntries<K1T> + 3) / 4) * 4 + (((kMelEntries<K1T> + 3) / 4) % 2 == 0 ? 4 : 0);  // 16-byte units: odd count
static_assert(kMelEntries<6> == 31 && kMelRowPitch<6> == 36 && kMelEntries<17> == 41 && kMelRowPitch<17> == 44, "mel table shape");

struct MfccTablesDev {
    int K1 = 0;               // K+1 filters / cepstral coefficients
    bool mel_sparse = false;  // K1 is 6 or 17 and the mel bank is non-zero only where mel_touches<K1> says (checked on upload)
    float *hamming = nullptr; // [480]
    float2 *tw240 = nullptr;  // [240] exp(-2*pi*i*k/240)
    float2 *tw480 = nullptr;  // [240] exp(-2*pi*i*k/480)
    float *fb = nullptr;      // [K1][240]
    float *melw = nullptr;    // mel_sparse: [16][kMelRowPitch<K1>] compact per-lane weights (mel_index order)
    float *dct = nullptr;     // [K1][K1] cos table, row k col n
};

// Templates of one length that one wave scores together (register kernel).
constexpr int kChunkMax = 8;
struct DtwChunk {
    int len;              // frames of every template in the chunk
    int count;            // real templates, 1..tc
    int tc;               // register tile the kernel is instantiated for: 2, 4 or 8
    int rows_off;         // float offset of the chunk's rows in TemplatesDev::dup
    int tid[kChunkMax];   // output column of each template; T means the averaged template
    int aimg_off;         // chunks of 3..8 templates at mfcc_size 5: offset (16-byte units) of the chunk's A image in TemplatesDev::aimg
    int aimg3_off;        // ... of its three-part bf16 image in TemplatesDev::aimg3 (kDtwMfma3RowBytes per row)
};

// dtw_mfma_kernel (rp_dtw_mfma.hip): A image = per template row [k half 2][template 8] x 8 f16 (the negated unit row, split in two f16
// parts, in the slot order of the MFMA's B operand) for len + 16 rows (the tail rows are zero); per wave two stream segments of frames.
constexpr int kDtwMfmaRowBytes = 256;
// The f32-grade form (rp_ctx arithmetic RP_ARITH_F32_MATRIX, the default): the negated unit row as THREE bf16 parts a0 + a1 + a2 (exact: 3 x 8
// significant bits = an f32's 24), two k-steps of 16 slots per row = [k-step 2][k half 2][template 8] x 8 bf16.
constexpr int kDtwMfma3RowBytes = 512;
__host__ __device__ inline int dtw_mfma_stage_floats(int max_len) { return ((32 + 2 * (max_len + 3)) * 5 + 3) & ~3; }
// dtw_mfma_wide_kernel (mfcc_size 13 / 16): components per lane half and k-steps of 16 f16 slots (rp_dtw_mfma_wide.hip)
__host__ __device__ constexpr int dtw_mfma_wide_chm(int K) { return (K + 1) / 2; }
__host__ __device__ constexpr int dtw_mfma_wide_ksteps(int K) {
    return (3 * (dtw_mfma_wide_chm(K) / 2) + (dtw_mfma_wide_chm(K) % 2 ? 2 : 0) + 3) / 4;
}
// bytes from one template row of the wide A image to the next: the k-steps' 256-byte blocks + 128 -- a ds_read_b128 takes 16 lanes at a
// time = two row slots x 128 bytes; a pitch that is a multiple of 256 bytes put both on the same 32 banks (SQ_LDS_BANK_CONFLICT = half of
// the kernel's LDS cycles, profiles/r05_wide_pmc.txt)
__host__ __device__ constexpr int dtw_mfma_wide_row_bytes(int K) { return dtw_mfma_wide_ksteps(K) * 256 + 128; }
// dtw_mfma_wide3_kernel (rp_dtw_mfma_wide3.hip; mfcc_size 13 / 16 in the three-part bf16 arithmetic): FOUR template slots per wave, six k-steps
// of 16 slots; A image = per template row [k-step 6][k half 2][template 4] x 8 bf16 = 768 bytes + 64 (the four row slots a ds_read_b128 serves
// fall on different banks)
constexpr int kDtwWide3KSteps = 6;
constexpr int kDtwWide3RowBytes = kDtwWide3KSteps * 128 + 64;
// The window side's operand of a lane half is ONE run of twelve registers read in overlapping four-register pieces at offsets 0, 2, 4, 4, 6, 8
// (rp_dtw_mfma_wide3.hip, w3_run_piece):
//   mfcc_size 16 (four component pairs a..d):          [P2a P2b P1a P1b P0a P0b P0c P0d P1c P1d P2c P2d]
//   mfcc_size 13 (three pairs a..c + one component s): [P2a P2b P1a P1b P0a P0b P0c S01 P1c S2C P2c  0 ]
// with Pk of a pair = (xk of its first component, xk of its second), S01 = (x0s, x1s), S2C = (x2s, the 1.0 of 1 - a.x in half 1).  Slot i of
// k-step ks of the A image holds the template parts that complete the products: W3Slot{component, part} for the slot's two bf16 (component
// within the lane half; -1: zero; -2: the constant's 1.0, half 1 only).  S01 is read three times -- against (a0s, a0s), (a1s, a1s), (a2s, 0) --
// so the odd component needs no register of its own per product; registers read more often than they have products meet zeros.
struct W3Slot { int c0, p0, c1, p1; };
__host__ __device__ constexpr W3Slot dtw_mfma_wide3_slot(int K, int ks, int i) {
    constexpr int pair16[6][4] = {{0, 1, 0, 1}, {0, 1, 0, 1}, {0, 1, 2, 3}, {0, 1, 2, 3}, {2, 3, 2, 3}, {2, 3, 2, 3}};
    constexpr int part16[6][4] = {{0, 0, 0, 0}, {1, 1, 0, 0}, {1, 1, 0, 0}, {2, 2, 1, 1}, {2, 2, 0, 0}, {1, 1, 0, 0}};
    constexpr W3Slot s13[6][4] = {
        {{0, 0, 1, 0}, {2, 0, 3, 0}, {0, 0, 1, 0}, {2, 0, 3, 0}},      // P2a P2b P1a P1b  x a0
        {{0, 1, 1, 1}, {2, 1, 3, 1}, {0, 0, 1, 0}, {2, 0, 3, 0}},      // P1a P1b x a1, P0a P0b x a0
        {{0, 1, 1, 1}, {2, 1, 3, 1}, {4, 0, 5, 0}, {6, 0, 6, 0}},      // P0a P0b x a1, P0c x a0, S01 x (a0s, a0s)
        {{0, 2, 1, 2}, {2, 2, 3, 2}, {4, 1, 5, 1}, {6, 1, 6, 1}},      // P0a P0b x a2, P0c x a1, S01 x (a1s, a1s)
        {{4, 2, 5, 2}, {6, 2, -1, 0}, {4, 0, 5, 0}, {6, 0, -2, 0}},    // P0c x a2, S01 x (a2s, 0), P1c x a0, S2C x (a0s, 1.0)
        {{4, 1, 5, 1}, {-1, 0, -1, 0}, {4, 0, 5, 0}, {-1, 0, -1, 0}}}; // P1c x a1, S2C x 0, P2c x a0, 0 x 0
    return K == 16 ? W3Slot{2 * pair16[ks][i], part16[ks][i], 2 * pair16[ks][i] + 1, part16[ks][i]} : s13[ks][i];
}
// Tiles a wave of the matrix-core DTW kernels takes by its own index before it turns to the chunk's atomic counter: every whole round
// of a launch of at most three rounds (live-stream calls, BASELINE config C2 -- the waves start together and would ask for their
// tickets together; the counter then hands out what is left), the first round of longer launches (the waves drift apart by themselves
// and the counter evens out what the CUs finish unevenly) and of lists whose length only the device knows.
inline unsigned mfma_static_rounds(size_t total_tiles, size_t chunk_waves, bool list) {
    const size_t r = chunk_waves ? total_tiles / chunk_waves : 0;
    if (const char *e = std::getenv("RP_MFMA_STATIC_ROUNDS")) return (unsigned)std::atoi(e);  // A/B runs: 0 = every tile from the counter
    return (list || r > 3 || r < 1) ? 1u : (unsigned)r;
}
inline size_t dtw_mfma_lds_bytes(int max_len, int waves, int row_bytes = kDtwMfmaRowBytes) {
    return (size_t)(max_len + 16) * (size_t)row_bytes + (size_t)waves * (size_t)dtw_mfma_stage_floats(max_len) * sizeof(float);
}

// The arithmetic a context asks of the DTW launchers (rp_ctx_new flags / rp_ctx_set_arithmetic, include/rustpotter_hip.h RP_ARITH_*): read per
// call through TemplatesDev::arith, so that one template set serves every mode.
//   kArithF32Matrix (default): matrix-core kernels only in their f32-grade form -- operands as three bf16 parts (exact), six of the nine partial
//     products of a multiplication (what is dropped is below 2^-22 of it, 2^-25.7 rms: an f32 multiply itself rounds by up to 2^-24) -- and the
//     f32 vector kernels wherever no such form exists;
//   kArithStrictF32: the f32 vector ("register") kernels only, every product an f32 FMA;
//   kArithFastSplit: the two-part f16 forms of rounds 3-5 (22-bit operands, one partial product dropped): dtw_mfma_kernel, dtw_mfma_group_kernel,
//     dtw_mfma_wide_kernel; `ragged` additionally admits dtw_ragged_kernel (same two-part arithmetic on per-stream offsets).
enum { kArithF32Matrix = 0, kArithStrictF32 = 1, kArithFastSplit = 2 };
struct DtwArith {
    int mode = kArithF32Matrix;
    int ragged = 0;
};

// Device-resident template set of one wakeword reference.
struct TemplatesDev {
    const DtwArith *arith = nullptr;   // the owning context's setting (host memory; null = defaults)
    int arith_mode() const { return arith ? arith->mode : (int)kArithF32Matrix; }
    bool arith_ragged() const { return arith && arith->mode == kArithFastSplit && arith->ragged; }
    int T = 0;        // sample templates
    int K = 0;
    int Lpad = 0;     // row pitch (frames) of `unit`
    int has_avg = 0;  // template index T is the averaged template
    int max_len = 0;  // max over sample templates
    int max_diff = 0; // max(0, longest template incl. avg - max_len): >0 forces the generic kernel
    int *lens = nullptr;   // [T+has_avg]
    float *unit = nullptr; // [T+has_avg][Lpad][K] rows scaled to unit L2 norm (zero rows stay zero)
    // register-kernel layout: chunks sorted by class (see class_first); dup holds per chunk [len][tc/2][K][2]: the coefficients of two templates
    // interleaved, so that one scalar 8-byte load feeds both halves of a packed f32 FMA.
    DtwChunk *chunks = nullptr;
    float *dup = nullptr;
    // chunk classes: 0 = two templates (tc 2), 1 = three or four (tc 4), 2 = five to eight (tc 8), 3 = ONE template (scored two
    // windows per lane by dtw_band2_kernel; the averaged template is the LAST chunk of class 3)
    int class_first[4] = {0, 0, 0, 0};
    int class_count[4] = {0, 0, 0, 0};
    // every class-2 chunk once more as two tc-4 halves (only when each of them holds 7 or 8 templates): a small batch whose
    // tc-8 waves would fill the chip 2.x times is scored by twice as many tc-4 waves, three resident per SIMD instead of two
    int split_first = 0, split_count = 0;
    // dtw_mfma_kernel: A images of the class-1 and class-2 chunks (mfcc_size 5 only), and the shortest template among them
    void *aimg = nullptr;
    void *aimg3 = nullptr;  // the same chunks' three-part bf16 images (kDtwMfma3RowBytes per row, DtwChunk::aimg3_off)
    int mfma_min_len = 0;   // shortest template among the class-2 chunks (8 template slots; needs >= 12 frames)
    int mfma_min_len4 = 0;  // ... among the class-1 chunks (4 template slots; needs >= 16 frames)
    // mfcc_size 13 / 16: the sample templates once more as chunks of up to 8 same-length templates for dtw_mfma_wide_kernel (only when
    // every length occurs at least three times: the matrix kernel always pays for eight template slots)
    int wide8_first = 0, wide8_count = 0;
    // ... and as chunks of up to 4 for dtw_mfma_wide3_kernel (three-part bf16 images in aimg3, DtwChunk::aimg3_off)
    int wide4_first = 0, wide4_count = 0, wide4_min_len = 0;
    int n_chunks_total = 0;   // entries of `chunks` that index tile counters (all but the ragged chunks at its end; the counters live in the CALL's workspace, DtwWork::sched)
    // The reference forms the cosine as dot_ab / sqrt(dot_a * dot_b) in f32 (comparator.rs:28-48): the PRODUCT of the squared norms
    // can underflow (-> "magnitude == 0" -> similarity 0) or overflow where neither factor does.  The kernels above are scale
    // invariant (unit-length rows and frames), so they only agree with it while both squared norms stay in a range where the product
    // is a normal f32: kDtwNormLo..kDtwNormHiRow for template rows (checked here, on upload), ..kDtwNormHiFrame for window frames
    // (checked by every kernel, per frame).  `raw` = the rows as given, for the reference-shaped cell of dtw_ref_kernel; ref_only:
    // some row is outside the range -- every window of this set is scored by dtw_ref_kernel.
    float *raw = nullptr;     // [T+has_avg][Lpad][K]
    int ref_only = 0;
    // dtw_mfma_group_kernel (rp_dtw_mfma_group.hip): runs of 4 consecutive class-2 chunks of one template length, grp_first[] (device) the
    // first chunk of each; rest_*: the class-2 chunks outside every group, as runs for dtw_mfma_kernel
    int *grp_first = nullptr;
    int grp_count = 0, grp4_max_len = 0;
    int rest_runs = 0, rest_first[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rest_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // dtw_ragged_kernel (rp_dtw_ragged.hip, mfcc_size 5): the sample templates the equal-length matrix kernel does not take (lengths that
    // occur once or twice) as chunks of up to 8 templates of ANY lengths, shortest first; rimg = per template its A image, 32 bytes per
    // row ([k half 2] x 8 f16: the negated unit row split in two f16 parts) + 16 zero rows, rag_off[t] its offset in 16-byte units
    int rag_first = 0, rag_count = 0;
    int rag_min_len = 0;      // shortest template among them (needs >= 16 frames)
    int rag_a_cap = 0;        // largest chunk's images in bytes
    int rag_rows_cap = 0;     // ... unit rows in floats (multiple of 4)
    void *rimg = nullptr;
    int *rag_off = nullptr;   // [T]
};
// squared-norm range in which the scale-invariant cell equals the reference's: 2^-60 <= |row|^2 <= 2^60, 2^-60 <= |frame|^2 <= 2^30
// (products stay within 2^-120 .. 2^90; a frame is tested with ONE v_max3_f32 of its squared norm and its reciprocal square root
// against 2^30).  Zero rows / frames are exact in both forms (similarity 0).
constexpr float kDtwNormLo = 8.673617379884035e-19f, kDtwNormHiRow = 1.152921504606847e18f, kDtwFixLimit = 1073741824.f;

// Per-CALL device workspace of the DTW launchers (owned by the context, Ctx::dtw_work): the tile counters of the matrix-core kernels
// ({next tile, workgroups done} per chunk, zero between launches: the last workgroup of a launch puts them back) and the list of
// (window, chunk) pairs whose frames left the range above (fix[0] = entries appended, fix[1] = workgroups of dtw_ref_kernel done,
// entries from fix[2] on as 64-bit words row << 24 | spec).  Calls on one context are serialised (one stream), so the words are
// never shared by two launches.
constexpr int kDtwSchedChunks = 2048;       // chunks of one template set the counters cover
constexpr uint32_t kDtwFixCap = 1u << 18;   // listed (window, chunk) pairs; beyond: every window of the call is rescored
constexpr uint32_t kFixSpecTemplate = 1u << 22, kFixSpecAll = 1u << 23, kFixSpecMask = (1u << 24) - 1;
struct DtwWork {
    uint32_t *sched = nullptr;   // [2 * kDtwSchedChunks]
    uint32_t *fix = nullptr;     // [2 + 2 * kDtwFixCap + 2]: the last two words count the pairs rescored since the context was made
    uint32_t *ran = nullptr;     // HOST word (may be null): the launchers OR in the kernel families they launched (kDtwRan*, rp_ctx_dtw_kernels)
    // dtw_ragged_kernel's per-call device blocks (Ctx::dtw_work_for; null: that kernel is not taken): rag_prep [rag_streams][8] floats
    // (offset and scale of every stream), rag_list [1 + rag_rows] words (the windows to score again with the register kernels; rows x ragged chunks)
    float *rag_prep = nullptr;
    uint32_t *rag_list = nullptr;
    size_t rag_streams = 0, rag_rows = 0;
};
// == RP_DTW_KERNEL_* (include/rustpotter_hip.h)
enum : uint32_t { kDtwRanMfma = 1u, kDtwRanMfmaWide = 2u, kDtwRanRagged = 4u, kDtwRanRegister = 8u, kDtwRanGeneric = 16u, kDtwRanSingle = 32u, kDtwRanRefAll = 64u, kDtwRanMfmaGroup = 128u,
                  // the product arithmetic of the matrix-core launches: three bf16 parts (f32-grade) / two f16 parts (22-bit)
                  kDtwRanBf16x3 = 256u, kDtwRanF16x2 = 512u };
inline void dtw_mark(const DtwWork &wk, uint32_t bit) { if (wk.ran) *wk.ran |= bit; }
__host__ __device__ inline unsigned long long *dtw_fix_stats(uint32_t *fix) { return reinterpret_cast<unsigned long long *>(fix + 2 + 2 * (size_t)kDtwFixCap); }
// (dtw_fix_append, the kernels' side of the list: rp_device.h)

// The matrix-core DTW kernel (rp_dtw_mfma.hip) for the chunks of class 2 (5..8 templates; slots = 8, band 3..5) and class 1 (3..4
// templates; slots = 4, band 5) at mfcc_size 5.  from_global: lanes read their frames from global memory (live-stream batches, LIST mode
// of the averaged-template gate) instead of an LDS stage (needs n_win >= 32).  list / count / dense_min / abandon_nc: as GateList in
// rp_dtw.hip.
// score_ref: the relative error of a score is (1 - score) x d(cost / (m + n)) / score_ref; the split products keep it within the parity
// gate (1e-5) down to kDtwMfmaMinScoreRef (tools/probe_score_ref.py, tests/test_gpu_round4.py); below, the f32 register kernels score
constexpr float kDtwMfmaMinScoreRef = 0.05f;
bool dtw_mfma_supported(const TemplatesDev &t, int band, size_t n_win, bool from_global, int slots, float score_ref);
// mfcc_size 13 / 16 at band 5: frames always from global memory (the caller's rows end with slack: launch_dtw's padded_rows)
bool dtw_mfma_wide_supported(const TemplatesDev &t, int band, float score_ref);
hipError_t launch_dtw_mfma_wide(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg, const uint32_t *list,
                                const uint32_t *count, uint32_t dense_min, float abandon_nc);
// the same shapes in the default arithmetic (three bf16 parts per operand, no score_ref floor): chunks of up to four templates
bool dtw_mfma_wide3_supported(const TemplatesDev &t, int band);
hipError_t launch_dtw_mfma_wide3(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                 size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg, const uint32_t *list,
                                 const uint32_t *count, uint32_t dense_min, float abandon_nc);
// ScoreMode::Max folded into the matrix-core DTW kernel when ONE chunk holds every sample template of the reference and no averaged
// template is scored in the call (BASELINE C2 / C3, a live-stream call with same-length templates): the lane pair of a window holds all
// its scores, so the kernel also writes agg[row] = max_t score and raises the stream's `hot` flag like agg_store (rp_dtw.hip) -- the
// aggregate pass (a launch, and a second read of every score) is skipped.  launch_dtw sets `done` when it took that route.
struct DtwFusedAgg {
    float *agg = nullptr;
    uint32_t *hot = nullptr;   // one flag per stream, zero before the launch (Ctx::hot_flags: the scan leaves them so); may be null
    float threshold = 0.f;
    bool done = false;
};
size_t dtw_mfma_group_lds_bytes(int L, int sh);
bool dtw_mfma_group_supported(const TemplatesDev &t, int band, size_t n_win, size_t S, float score_ref);
hipError_t launch_dtw_mfma_group(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch,
                                 size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores);
hipError_t launch_dtw_mfma(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, int slots, int chunk_base, int n_chunks, const float *mfcc, size_t S,
                           size_t frame_pitch, size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float *avg,
                           bool from_global, const uint32_t *list, const uint32_t *count, uint32_t dense_min, float abandon_nc,
                           const DtwFusedAgg *fuse = nullptr);

// The matrix-core DTW kernel for templates of unequal length (rp_dtw_ragged.hip): the ragged chunks of `t` over every window of the call
// (LDS-staged tiles of 512 windows: needs n_win >= 64; live-stream batches and the gate's list keep the register kernels).
constexpr int kDtwRaggedWaves = 8;    // waves per workgroup = 512 windows per tile
constexpr int kDtwRaggedSegs = 16;    // stream segments a tile may span
// the cell error is 2^-22 x |x_f - o| / |x_f - mu| (a few times dtw_mfma_kernel's): floor of score_ref for the 1e-5 parity gate
constexpr float kDtwRaggedMinScoreRef = 0.1f;
bool dtw_ragged_supported(const TemplatesDev &t, int band, size_t n_win, float score_ref);
size_t dtw_ragged_lds_bytes(const TemplatesDev &t, size_t n_win, int *frames_cap);
// list_rows: windows the kernel cannot score within the parity gate go to wk.rag_list (the caller runs the register kernels' list mode on it);
// else to wk.fix (dtw_ref_kernel)
hipError_t launch_dtw_ragged(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, int band, const float *mfcc, size_t S, size_t frame_pitch,
                             size_t first_win, size_t n_win, size_t out_win_pitch, float score_ref, float *scores, float abandon_nc, bool list_rows);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process that drives several GPUs
// (one rp_ctx per device) has to set it on each of them.  Sets it once per (current device, kernel), thread-safe.
hipError_t allow_dynamic_lds(const void *kernel, int bytes);
// compute units of the current device (cached per device)
int device_cu_count();

enum KernelId { kKernelMfcc = 0, kKernelDtw = 1, kKernelAggregate = 2, kKernelScan = 3, kKernelMlp = 4, kKernelResample = 5, kKernelCount = 6 };

// Sample-rate converter plan (rp_resampler.cpp): out[j] = sum_n x2[n] * g2t[j][n], x2 = previous | current input frame
struct ResamplerDev {
    int fs_in = 0, fi = 0, fo = 0, kpad = 0;  // input rate, input / output frame length, 2*fi rounded up to 16
    const float *g2t = nullptr;                // [fo][kpad]
    const float *fft48 = nullptr;              // 48 kHz only: twiddles + filter spectrum of resample48_fft_kernel
};

// table block of resample48_fft_kernel, offsets in float2 units (rp_resampler.cpp fills it)
constexpr int kR48OffTw240 = 0, kR48OffTw480 = 240, kR48OffTwc = 480, kR48OffW960c = 480 + 6 * 480, kR48OffHf = kR48OffW960c + 256,
              kR48TableLen = kR48OffHf + 480;

struct ScanConfig {
    float threshold, avg_threshold;
    int min_scores, eager, max_len, avg_enabled;
    int fpf = 3;  // MFCC frames per input frame of the stream's encoder: 3 (480 encoded samples) or 4 (640: 11.025 / 22.05 kHz input)
    int stream_base = 0;  // added to the stream index a detection reports (a shard's first global stream, rp_batch_detect_sharded)
};

// the wakewords of one detector in the batched scan (run_wakeword_detectors, src/detector.rs:433-447): per wakeword
// its aggregate scores [S][n_win], avg-template scores (nullptr: gate off) and its own thresholds
constexpr int kScanMaxWakewords = 8;
struct ScanWakewords {
    int n;
    const float *agg[kScanMaxWakewords], *avg[kScanMaxWakewords];
    float threshold[kScanMaxWakewords], avg_threshold[kScanMaxWakewords];
    const int32_t *label[kScanMaxWakewords];  // model wakewords: the winning label of every window (reported instead of j)
    uint32_t *hot = nullptr;                  // [S] from the aggregate pass (AggExtra): 0 = no window of the stream can fire; the scan puts the flags it reads back to 0
};

struct BatchDetection {  // == rp_batch_detection
    int32_t stream, frame, window, counter;
    float avg_score, score;
};

// mfcc2 (optional): a second, packed copy of the frames [S][n_frames][K]
hipError_t launch_mfcc(hipStream_t st, const MfccTablesDev &tb, const float *pcm, size_t S, size_t n_samples,
                       size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc,
                       float *mfcc2 = nullptr);

// pcm in one of the reference's sample formats (rp_sample_format: 0 i8, 1 i16, 2 i32, 3 f32), decoded in the kernel
hipError_t launch_mfcc_stream(hipStream_t st, const MfccTablesDev &tb, const void *pcm, int fmt, size_t S, size_t n_chunks,
                              size_t pcm_stride, const float *hist, size_t hist_pitch, float *hist_out, size_t out_frame_pitch,
                              float *mfcc);
hipError_t launch_mfcc_fmt(hipStream_t st, const MfccTablesDev &tb, const void *pcm, int fmt, size_t S, size_t n_samples,
                           size_t pcm_stride, size_t first_frame, size_t n_frames, size_t out_frame_pitch, float *mfcc);

// scores [S][n_win][T]; avg [S][n_win] or nullptr.  mfcc rows have `frame_pitch` frames per stream.
// abandon_nc (dtw_abandon_nc(threshold, score_ref), or +inf = off): DETECT-ONLY calls in ScoreMode::Max may stop a wave
// whose windows x templates all cost more than any score above `threshold` allows; those rows get score 0 (GateList in
// rp_dtw.hip).  Every window that can fire keeps exact scores, so the detections do not change.
float dtw_abandon_nc(float threshold, float score_ref);
hipError_t launch_dtw(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch,
                      size_t first_win, size_t n_win, size_t out_win_pitch, int band, float score_ref, int with_avg,
                      float *scores, float *avg, bool padded_rows = false, float abandon_nc = __builtin_inff(), DtwFusedAgg *fuse = nullptr);

// the same gate for template sets only dtw_generic_kernel serves (dtw_uses_generic), at wave granularity, and for the
// single-stream API (one launch for the averaged template, one for the sample templates when a window passed)
bool dtw_uses_generic(const TemplatesDev &t, int band, size_t S, size_t n_win);
hipError_t launch_dtw_generic_gated(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                                    size_t n_win, size_t out_win_pitch, int band, float score_ref, float avg_threshold, float *scores,
                                    float *avg);
hipError_t launch_dtw_single_part(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t frame_pitch, size_t first_win, size_t n_win,
                                  size_t out_win_pitch, int band, float score_ref, int t_first, int t_count, float *scores, float *avg);

// The averaged-template gate as a skip (wakeword_comp.rs:85-93): every window against the averaged template (-> avg),
// the rows with avg >= avg_threshold listed (list [S*n_win] / count: device workspaces), the sample templates on the
// listed rows only (-> scores [S][n_win][T]; other rows are not written).  mfcc needs 64*K floats of slack behind the last stream.
// hipErrorNotSupported when the template set has no register kernel for this (see dtw_gate_supported).
bool dtw_gate_supported(const TemplatesDev &t, int band, size_t rows);
hipError_t launch_dtw_gated(hipStream_t st, const DtwWork &wk, const TemplatesDev &t, const float *mfcc, size_t S, size_t frame_pitch, size_t first_win,
                            size_t n_win, int band, float score_ref, float avg_threshold, float *scores, float *avg, uint32_t *list,
                            uint32_t *count, bool few_windows = false, float abandon_nc = __builtin_inff());

// Largest template tile the register DTW kernel is built for (0: only the generic kernel applies).
int dtw_register_tile(int K, int band);

// optional extras of the aggregate pass (all null = plain aggregate): gate_avg / gate_threshold -- rows whose averaged-template
// score is below the threshold get aggregate 0 (they were never scored); hot / threshold / n_win -- hot[row / n_win] = 1 for
// streams with a window that can fire (the caller zeroes `hot` first)
struct AggExtra {
    const float *gate_avg = nullptr;
    float gate_threshold = 0.f;
    uint32_t *hot = nullptr;
    float threshold = 0.f;
    size_t n_win = 1;
};
hipError_t launch_aggregate(hipStream_t st, const float *scores, size_t n_rows, int T, int mode, float *agg, AggExtra x = AggExtra{});

// vad_value [S][n_frames] = mean |mfcc| per frame (launch_vad_value) or nullptr (no VAD);
// vad_mode_value = VADMode::get_value (2 / 2.5 / 3, src/config.rs:140-146)
hipError_t launch_vad_value(hipStream_t st, const float *mfcc, size_t n_frames_total, int K, float *out);
hipError_t launch_vad_value_rows(hipStream_t st, const float *mfcc, size_t S, size_t n, size_t pitch, int K, float *out);
hipError_t launch_scan_multi(hipStream_t st, const ScanWakewords &ww, const float *vad_value, float vad_mode_value, size_t S,
                             size_t n_frames, const ScanConfig &cfg, BatchDetection *det, int32_t *det_ww, int32_t *n_det, int max_det);
// hot (optional): the per-stream flags of the aggregate pass (AggExtra) -- streams whose flag is 0 report no detection unseen
hipError_t launch_scan(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                       size_t S, size_t n_frames, const ScanConfig &cfg, BatchDetection *det, int32_t *n_det, int max_det,
                       uint32_t *hot = nullptr);

// Decode + GainNormalizerFilter + BandPassFilter over whole streams.  ring [S][window_size], rms / gains
// [S][n_samples/480] are device workspaces; biquad coefficients as BandPassFilter::new computes them.
hipError_t launch_frontend(hipStream_t st, const void *pcm, int fmt, size_t S, size_t n_samples, size_t pcm_stride, int gain_on,
                           float rms_level_ref, float min_gain, float max_gain, int window_size, int band_pass, float a0,
                           float a1, float a2, float b1, float b2, float *ring, float *rms, float *gains, float *out,
                           size_t out_stride);

// Resampler front-end.  Stage: decode + first channel + history -> xs [S][(1+n_chunks)*fi] f32 (prev [S][fi] or
// nullptr = silence before the stream); resample: xs -> out [S][n_chunks*fo].
hipError_t launch_resample_stage(hipStream_t st, const void *pcm, int fmt, int channels, size_t S, size_t n_chunks, int fi,
                                 size_t pcm_stride, const float *prev, float *xs);
hipError_t launch_resample48(hipStream_t st, const float *tables, const void *pcm, int fmt, int channels, size_t pcm_stride,
                             int has_hist, const float *prev, float *prev_out, size_t S, size_t n_chunks, float *out,
                             size_t out_stride);
bool resample_reads_in_place(const ResamplerDev &rs, const void *pcm, int fmt, size_t pcm_stride, const float *out, size_t out_stride);
hipError_t launch_resample_in_place(hipStream_t st, const ResamplerDev &rs, const void *pcm, int fmt, int channels, size_t pcm_stride,
                                    const float *prev, float *prev_out, size_t S, size_t n_chunks, float *out, size_t out_stride);
hipError_t launch_resample(hipStream_t st, const ResamplerDev &rs, const float *xs, size_t S, size_t n_chunks, float *out,
                           size_t out_stride);

// streaming batches (state carried between calls; rp_stream.cpp)
hipError_t launch_stream_stage(hipStream_t st, const void *pcm, int fmt, int channels, size_t S, size_t n_new, size_t pcm_stride,
                               const float *old_hist, size_t old_off, float *hist, size_t hist_pitch);
hipError_t launch_carry_rows(hipStream_t st, const float *src, size_t S, size_t src_pitch, size_t src_off, size_t count, float *dst,
                             size_t dst_pitch);
hipError_t launch_stream_state_init(hipStream_t st, void *state, size_t S);
size_t stream_state_bytes();
hipError_t launch_stream_state_reset(hipStream_t st, void *state, size_t S, long long stream, long long resume);
hipError_t launch_scan_stream(hipStream_t st, const float *agg, const float *avg, const float *vad_value, float vad_mode_value,
                              size_t S, long long f0, int n_new, const ScanConfig &cfg, void *state, BatchDetection *det,
                              int32_t *n_det, int max_det);
// the same for a detector that holds several wakewords (references and / or models); det_ww / det_label (optional): the
// wakeword a detection belongs to and, when that wakeword is a model, its label index (else -1)
hipError_t launch_scan_stream_multi(hipStream_t st, const ScanWakewords &ww, const float *vad_value, float vad_mode_value, size_t S,
                                    long long f0, int n_new, const ScanConfig &cfg, void *state, BatchDetection *det, int32_t *det_ww,
                                    int32_t *det_label, int32_t *n_det, int max_det);

hipError_t launch_synth(hipStream_t st, uint64_t seed, uint64_t first_stream, size_t S, size_t n_samples,
                        size_t pcm_stride, float *pcm);

// x [B][dims[0]] -> out [B][dims[n_layers]]; W/Bv device pointers per layer.
// MfccNormalizer::normalize of windows [w, w+L) flattened row-major to x [n_win][L*K]
// (WakewordNN::run_detection, src/wakewords/nn/wakeword_nn.rs:139-149,268-273).
hipError_t launch_normalize_windows(hipStream_t st, const float *mfcc, size_t first_win, size_t n_win, int L, int K,
                                    float *x);

// Device-resident wakeword model (src/wakewords/nn/wakeword_nn.rs:305-389): layer 1 padded for
// the MFMA kernels, the small tail layers packed for the per-row epilogue.
struct MlpDev {
    int n_layers = 0;
    int dims[5] = {0, 0, 0, 0, 0};
    int nt = 0;            // 16-column tiles of layer 1 (1, 2, 5 or 9)
    int kpad = 0;          // dims[0] rounded up to 128
    float *w1f = nullptr;  // [16*nt][kpad] f32, zero padded
    void *w1h = nullptr;   // [16*nt][kpad] bf16, zero padded
    void *w1s = nullptr;   // [2][16*nt][kpad] f16: the two parts of the weights' f16 split (kMlpF16x2), zero padded
    void *w1t = nullptr;   // [3][16*nt][kpad] bf16: the three parts of the weights' exact bf16 split (kMlpBf16x3), zero padded
    void *wwin = nullptr;  // mfcc_size 16, layer 1 <= 160 wide: the same two parts in the order the lanes of mlp_windows_kernel (<= 32 wide) /
                           // mlp_windows_wide_kernel read them, [frame f][32-output tile q][part][k-half h][output j < 32][8 k = 16 f + 8 h ..] f16
    void *wwin3 = nullptr; // the same in three bf16 parts (kMlpBf16x3): [frame f][32-output tile q][part 3][k-half h][output j < 32][8 k]
    float *b1 = nullptr;   // [16*nt]
    float *tail = nullptr; // layers 2..n: W [out][in] then b [out], concatenated
    int tail_floats = 0;
};
// kMlpF16x2: layer 1 at the matrix cores' f16 rate -- inputs and weights as f16 two-way splits
// (x = x0 + x1, w = w0 + w1, 22 significant bits each; x0 w0 + x1 w0 + x0 w1, f32 accumulate: the dropped x1 w1 is 2^-22 of a product):
// RP_MLP_F32_FAST callers (through round 5 what RP_MLP_F32 meant)
// kMlpStrictF32: the f32 matrix instructions for every row (RP_MLP_F32_STRICT callers).  kMlpRedoF32 (internal): only
// the second pass of kMlpF16x2 -- the rows listed in `redo` again with the f32 matrix instructions (behind launch_mlp_stream).
// redo (Ctx::mlp_redo): [2 + B] words, redo[0] = rows listed, redo[1] = workgroups of the second pass done, both zero between calls;
// a row is listed when one of its features is beyond the f16 range (|x| > 65504, or not finite): the split cannot hold it, the f32
// instructions can -- rp_mlp_forward_batch never answers NaN for finite input.
// kMlpBf16x3: f32-grade layer 1 at the matrix cores' bf16 rate -- inputs and weights as THREE bf16 parts each (exact), six of the nine partial
// products, f32 accumulate: what RP_MLP_F32 callers get (round 6; kMlpF16x2, 22-bit operands, is RP_MLP_F32_FAST).  No range limit, no second pass.
enum { kMlpF32 = 0, kMlpBf16 = 1, kMlpF16x2 = 2, kMlpStrictF32 = 3, kMlpRedoF32 = 4, kMlpBf16x3 = 5 };
// A launcher that fails between the split pass and the pass over the listed rows must not leave rows of THIS call listed for the next one
// (the next call would append behind them and run its second pass on stale indices): the two counter words go back to zero behind
// whatever was queued, as dtw_abort does for the DTW words.  Returns the error it was given.
inline hipError_t mlp_redo_abort(hipStream_t st, uint32_t *redo, hipError_t e) {
    if (redo) (void)hipMemsetAsync(redo, 0, 2 * sizeof(uint32_t), st);
    (void)hipGetLastError();
    return e;
}
// Fused forward of all layers; layer 1 on the matrix cores (f32-input MFMA: bit-for-bit an fmaf
// chain; or bf16 inputs with f32 accumulation), tail layers + ReLU per row in f32.
hipError_t launch_mlp_mfma(hipStream_t st, const MlpDev &m, const float *x, size_t B, int precision, float *out, uint32_t *redo);
bool mlp_mfma_fits(const MlpDev &m);   // the fused kernel's LDS fits a CU (else: the per-layer kernel)
// The same forward for dense rows with the rows streamed through LDS by LDS-DMA in whole 128-byte lines (rp_mlp_stream.hip):
// layer-1 widths <= 32, row pitch a multiple of 64 bytes, x 16-byte aligned (mlp_stream_supported).  The plan holds the
// layer-1 weights in MFMA-fragment order and the line phases of the rows (Model::stream_plan).
struct MlpStreamPlan {
    const void *wimg = nullptr;                // [k-step m][...] k-step m = k in [32m, 32m+32), zero padded; see Model::stream_plan
    int q[2] = {0, 0};                         // even / odd rows: 16-byte chunks between a row's start and the line start below it
    int units = 0;                             // 64-k steps per row (2 lines of 128 bytes)
    int par = 0;                               // 1: even and odd rows differ in phase -- a workgroup takes rows of one parity
    int nbt = 0;                               // workgroup tiles: 128 rows (256-row span of one parity when par)
};
bool mlp_stream_supported(const MlpDev &m, const float *x, int precision = kMlpF32);
hipError_t launch_mlp_stream(hipStream_t st, const MlpDev &m, const MlpStreamPlan &p, const float *x, size_t B, int precision, float *out,
                             int n_cu, uint32_t *redo);
// The rows are windows of L = dims[0]/K frames read IN PLACE from mfcc [S][n_frames][K] (a window's flattened features
// are a contiguous slice of the frame array), row = s * n_win + w; the window mean (MfccNormalizer::normalize) is taken
// out after layer 1: W.(f - mu) = W.f - sum_k mu[k] * wsum[o][k], wsum[o][k] = sum_i W[o][i*K + k].  f32 MFMA.
// mean [S*n_win][K] from launch_window_means, wsum [16*nt][K].
hipError_t launch_window_means(hipStream_t st, const float *mfcc, size_t S, size_t n_frames, size_t n_win, int L, int K, float *mean);
// frame_pitch (0 = n_win + L - 1, whole streams): frames between the rows of two streams -- live-stream batches keep their
// windows in longer rows (window w of stream s starts at frame s * frame_pitch + w, counted from `mfcc`)
int mlp_windows_supported(const MlpDev &m, size_t n_win, int K, bool three_part = false);   // 1: mlp_windows_kernel, 2: mlp_windows_wide_kernel takes the call (0: mlp_mfma_kernel, rows read in place)
hipError_t launch_mlp_mfma_windows(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_frames, size_t n_win, int K,
                                   const float *mean, const float *wsum, float *out, uint32_t *redo, size_t frame_pitch = 0,
                                   int precision = kMlpF32);   // kMlpF32 (three bf16 parts, rows read in place) / kMlpF16x2 (RP_MLP_F32_FAST: the
                                                               // staged-frame kernels) / kMlpStrictF32

// WakewordModelTrain (src/wakewords/nn/wakeword_model_train.rs:204-209): act[l] / dz[l] are [B][dims[l+1]] device buffers
hipError_t launch_train_forward(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                                float *const *Bv, float *const *act);
hipError_t launch_train_step(hipStream_t st, const float *x, const int32_t *labels, size_t B, int n_layers, const int *dims,
                             float *const *W, float *const *Bv, float *const *act, float *const *dz, float lr, float *loss_rows);
// WakewordNN::run_detection (src/wakewords/nn/wakeword_nn.rs:39-159) over whole batches: windows of L frames cut from
// mfcc [S][frame_pitch][K] and mean-normalised into rows (first_row .. first_row+n_rows of the S*n_win windows), and the
// label / score logic on the logits: agg = score where the detection is valid (label != none, score >= threshold,
// avg_score >= avg_threshold) else -2, avg = avg_score, label = arg-max label.
hipError_t launch_normalize_windows_batch(hipStream_t st, const float *mfcc, size_t frame_pitch, size_t n_win, size_t first_row,
                                          size_t n_rows, int L, int K, float *x);
hipError_t launch_nn_score(hipStream_t st, const float *logits, size_t n_rows, int n_labels, int none_index, float score_ref10,
                           int calc_avg, float threshold, float avg_threshold, float *agg, float *avg, int32_t *label,
                           uint32_t *hot = nullptr, size_t rows_per_stream = 0);  // hot: a flag per stream (zeroed by the caller) raised by every window that passed
hipError_t launch_mlp(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                      float *const *Bv, float *scratch0, float *scratch1, float *out);

}  // namespace rp
