// dtw_mfma_probe2.hip -- second prototype of the banded DTW with the cosine costs on the matrix cores: f16 two-way splits.
//
// dtw_mfma_probe.hip (bf16, exact three-way splits, 30 + 1 slots = two k-steps, six MFMAs per column) came out at the shipped
// kernel's speed: every VALU instruction costs ~4 cycles whatever its kind, its ablations put a column at base 165 + frame
// 160 + recurrence 240 + MFMA 115 cycles.  This variant halves the matrix work and trims the frame work:
//   * x = x0 + x1, a = a0 + a1 with x0 = rtz_f16(x), x1 = rtz_f16(x - x0) (22 significant bits; v_cvt_pkrtz_f16_f32 converts
//     and packs two components at once), products x0 a0, x1 a0, x0 a1: 15 slots + the constant 1.0 x 1.0 = 16 = ONE k-step of
//     v_mfma_f32_32x32x16_f16: three MFMAs per column, A image 256 B per template row;
//   * lane (window n, half h) owns components (0, 1) or (3, 4) plus component 2: it reads, centres and splits only those; the
//     squared norm's two partial sums meet through v_permlane32_swap.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/scratch/dtw_mfma_probe2.hip -o /tmp/p2 && /tmp/p2 [L streams windows]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 5, W = 5, B = 2 * W, NS = 12;  // NS circular row slots (3 tiles x 4)
constexpr int kRowBytes = 256;                   // A image: [row][khalf 2][template 8] x 16 B
#define RP_INF __builtin_inff()
#define RP_LDSP(p) ((__attribute__((address_space(3))) void *)(p))
#ifndef NWAVES
#define NWAVES 8
#endif
#ifndef WGS_PER_CU
#define WGS_PER_CU 1
#endif
#ifndef RP_SCALAR_ADD
#define RP_SCALAR_ADD 0
#endif
#ifndef RP_ABL
#define RP_ABL 0
#endif

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// last band position q of column phase u whose MFMA row slot (u + q + 9) mod 12 lies in tile g (every tile is used by every column)
__host__ __device__ constexpr int last_use(int u, int g) {
    int last = -1;
    for (int q = 0; q < B; ++q)
        if (((u + q + NS - W + 2) % NS) / 4 == g) last = q;
    return last;
}
__device__ __forceinline__ unsigned pkrtz(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ float lo_f32(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p)[0]; }
__device__ __forceinline__ float hi_f32(unsigned p) { return (float)__builtin_bit_cast(fp16x2, p)[1]; }

// mfcc [S][F][K]; aimg [(L + 12)][2][8] x 16 B; scores [S * n_win][8]
__global__ __launch_bounds__(64 * NWAVES, WGS_PER_CU) void dtw_mfma_kernel(const float *__restrict__ mfcc, int F, int n_win, size_t n_streams,
                                                                           int L, const u32x4 *__restrict__ aimg, float score_ref,
                                                                           float *__restrict__ scores, size_t total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int a_bytes = (L + NS) * kRowBytes;
    const int xs_floats = ((34 + L) * K + 63) & ~63;  // per buffer; two buffers per wave (the next tile's frames arrive by LDS-DMA)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < a_bytes / 16; i += 64 * NWAVES) adst[i] = aimg[i];
    }
    __syncthreads();
    float *xs_wave = reinterpret_cast<float *>(smem + a_bytes) + wave * 2 * xs_floats;
    const int n = lane & 31, h = lane >> 5;
    // A operand: this lane supplies row m = lane & 31 of a tile = (slot 4g + jj, template 4h' + r'), k half = lane >> 5
    const int jj = (lane & 31) >> 3, tA = ((lane >> 2) & 1) * 4 + (lane & 3);
    const unsigned a_lane = (unsigned)(h * 128 + tA * 16);
    unsigned dl[4];  // byte offset back to the row this lane's slot holds when the newest row sits in slot e of its tile
#pragma unroll
    for (int e = 0; e < 4; ++e) dl[e] = (unsigned)(((e - jj + NS) % NS) * kRowBytes);
    const unsigned sel_one = h ? 0x07060100u : 0x03020100u;  // slot 7: x1 of component 2 (half 0) / the constant 1.0 (half 1)
    const int tiles_per_stream = n_win / 32;

    // frames of a tile: (33 + L) x K contiguous floats of its stream, 256 B per wave-instruction straight into LDS
    auto stage = [&](size_t tile, float *buf) {
        const size_t s = tile / tiles_per_stream;
        const int w0 = (int)(tile - s * tiles_per_stream) * 32;
        const float *src = mfcc + (s * F + w0) * K + lane;
        for (int j = 0; j < xs_floats; j += 64) __builtin_amdgcn_global_load_lds(src + j, RP_LDSP(buf + j), 4, 0, 0);
    };
    const size_t tile_step = (size_t)gridDim.x * NWAVES;
    size_t tile = (size_t)blockIdx.x * NWAVES + wave;
    int cur = 0;
    if (tile < total_tiles) stage(tile, xs_wave);
    for (; tile < total_tiles; tile += tile_step, cur ^= 1) {
        const size_t s = tile / tiles_per_stream;
        const int w0 = (int)(tile - s * tiles_per_stream) * 32;
        float *xs = xs_wave + cur * xs_floats;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
        if (tile + tile_step < total_tiles) stage(tile + tile_step, xs_wave + (cur ^ 1) * xs_floats);
        const float *xa = xs + n * K + (h ? 3 : 0);  // this half's two components; component 2 at xs + n * K + 2
        const float *x2 = xs + n * K + 2;
        float mua = 0.f, mub = 0.f, mu2 = 0.f;
#pragma unroll 10
        for (int i = 0; i < L; ++i) { mua += xa[i * K]; mub += xa[i * K + 1]; mu2 += x2[i * K]; }
        mua = mua / (float)L; mub = mub / (float)L; mu2 = mu2 / (float)L;

        // Q[p][q] = D[(c - 1) - W + 1 + q][c - 1] of the template pair p; column 0: D[0][0] = 0 at q = W - 1; Q[p][B] stays +inf
        v2f Q[2][B + 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[p][q] = (v2f){RP_INF, RP_INF};
            Q[p][W - 1] = (v2f){0.f, 0.f};
        }
        u32x4 Areg[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int slot = 4 * g + jj;
            int r = W - ((W - slot + NS) % NS);  // 1-based template row in this slot for the state "newest row = W"
            r = r < 1 ? 1 : r;
            Areg[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)(r - 1) * kRowBytes);
        }
        v16f acc[3];  // costs of the current column; a tile is refilled for the next column as soon as its last cell is done
        u32x4 bop[2];  // B operand of column cc in bop[cc & 1]: built two columns ahead, in pieces between the cells

// The frame work of column cc, cut into ten pieces P0..P9 that are placed between the cells of the recurrence (each cell ends with
// two packed adds whose results the next cell's v_min3 needs: the pieces fill those wait states instead of s_nop).
#define RP_P0(cc) fa_ = xa[((cc) - 1) * K]; fb_ = xa[((cc) - 1) * K + 1]; f2_ = x2[((cc) - 1) * K];
#define RP_P1(cc) da_ = fa_ - mua; db_ = fb_ - mub; d2_ = f2_ - mu2;
#define RP_P2(cc) own_ = fmaf(da_, da_, db_ * db_);
#define RP_P3(cc) { const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(own_), __float_as_uint(own_), false, false); \
                    bb_ = fmaf(d2_, d2_, __uint_as_float(sw_[0]) + __uint_as_float(sw_[1])); }
#define RP_P4(cc) inv_ = bb_ > 0.f ? __builtin_amdgcn_rsqf(bb_) : 0.f;
#define RP_P5(cc) ua_ = da_ * inv_; ub_ = db_ * inv_; u2_ = d2_ * inv_;
#define RP_P6(cc, par) bop[par].x = pkrtz(ua_, ub_); bop[par].z = bop[par].x;
#define RP_P7(cc, par) bop[par].y = pkrtz(ua_ - lo_f32(bop[par].x), ub_ - hi_f32(bop[par].x));
#define RP_P8(cc) t_ = pkrtz(u2_, 0.f);
#define RP_P9(cc, par) bop[par].w = __builtin_amdgcn_perm(0x3c000000u, pkrtz(u2_, u2_ - lo_f32(t_)), sel_one);
#define RP_PREP_ALL(cc, par) RP_P0(cc) RP_P1(cc) RP_P2(cc) RP_P3(cc) RP_P4(cc) RP_P5(cc) RP_P6(cc, par) RP_P7(cc, par) RP_P8(cc) RP_P9(cc, par)
// the A tile that receives template row cc + W (cc = 1 + uu mod 12)
#define RP_AREF(cc, uu, GUARD)                                                                                                \
    {                                                                                                                         \
        const int sn = ((uu) + 1 + W) % NS, g = sn / 4, e = sn % 4;                                                           \
        int off = ((cc) + W - 1) * kRowBytes - (int)dl[e];                                                                    \
        if (GUARD) off = off < 0 ? 0 : off;                                                                                   \
        Areg[g] = *reinterpret_cast<const u32x4 *>(smem + a_lane + (unsigned)off);                                            \
    }
#define RP_MFMA(g, par)                                                                                                       \
    do {                                                                                                                      \
        const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Areg[g]), __builtin_bit_cast(f16x8, bop[par]), zero16, 0, 0, 0); \
    } while (0)

// column c (c = 1 + u mod 12): rows r_q = c - W + 1 + q, q = 0..2W-1, sit in MFMA row slot (u + q + 9) mod 12.  Each tile's MFMA
// for column c + 1 goes out right after the last cell that reads the tile; the frame of column c + 2 is prepared between the cells.
#define RP_STEP(GUARD)                                                                                                        \
    do {                                                                                                                      \
        RP_AREF(c + 1, (u + 1) % NS, GUARD)                                                                                   \
        v2f up[2] = {(v2f){RP_INF, RP_INF}, (v2f){RP_INF, RP_INF}};                                                           \
        _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                                       \
            const int sl = (u + q + NS - W + 2) % NS;                                                                         \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) { /* two independent chains, interleaved */                         \
                const v2f cost = (v2f){acc[sl / 4][4 * (sl % 4) + 2 * p], acc[sl / 4][4 * (sl % 4) + 2 * p + 1]};             \
                v2f m, v;                                                                                                     \
                m.x = fminf(fminf(up[p].x, Q[p][q + 1].x), Q[p][q].x);                                                        \
                m.y = fminf(fminf(up[p].y, Q[p][q + 1].y), Q[p][q].y);                                                        \
                if (RP_SCALAR_ADD) { v.x = cost.x + m.x; v.y = cost.y + m.y; } else v = cost + m;                             \
                if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                              \
                Q[p][q] = v;                                                                                                  \
                up[p] = v;                                                                                                    \
            }                                                                                                                 \
            if (q == 0) { RP_P0(c + 2) } if (q == 1) { RP_P1(c + 2) } if (q == 2) { RP_P2(c + 2) } if (q == 3) { RP_P3(c + 2) } \
            if (q == 4) { RP_P4(c + 2) } if (q == 5) { RP_P5(c + 2) } if (q == 6) { RP_P6(c + 2, (u + 1) & 1) } if (q == 7) { RP_P7(c + 2, (u + 1) & 1) } \
            if (q == 8) { RP_P8(c + 2) } if (q == 9) { RP_P9(c + 2, (u + 1) & 1) }                                                         \
            _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                                     \
                if (last_use(u, g) == q) RP_MFMA(g, u & 1);                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        }                                                                                                                     \
    } while (0)

        float fa_, fb_, f2_, da_, db_, d2_, own_, bb_, inv_, ua_, ub_, u2_;
        unsigned t_;
        RP_AREF(1, 0, true)
        RP_PREP_ALL(1, 1)
        RP_MFMA(0, 1); RP_MFMA(1, 1); RP_MFMA(2, 1);
        RP_PREP_ALL(2, 0)
        __builtin_amdgcn_sched_barrier(0);
        int c0 = 1;
        {   // first block: cells of rows < 1 stay +inf (L >= 12)
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(true); }
        }
        for (c0 = 1 + NS; c0 + NS - 1 <= L; c0 += NS) {  // full blocks: u is compile time, so every slot, tile and band index is a fixed register
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int c = c0 + u; RP_STEP(false); }
        }
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) {  // the last L mod 12 columns
            const int c = c0 + u;
            if (c <= L) RP_STEP(false);
        }
#undef RP_STEP
#undef RP_MFMA
#undef RP_AREF

        // D[m - 1][n] with m == n == L: band position q = (L - 1) - (L - W + 1) = W - 2
        const v2f res[2] = {Q[0][W - 2], Q[1][W - 2]};
        const size_t row = s * n_win + w0 + n;
        const float denom = (float)(L + L);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float c0_ = res[p].x / denom, c1_ = res[p].y / denom;
            scores[row * 8 + 4 * h + 2 * p] = 1.f / (1.f + expf((c0_ - score_ref) / score_ref));
            scores[row * 8 + 4 * h + 2 * p + 1] = 1.f / (1.f + expf((c1_ - score_ref) / score_ref));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
static uint16_t f16_rtz_bits(float v) {  // |v| < 65504
    uint32_t b; memcpy(&b, &v, 4);
    const uint32_t sign = (b >> 16) & 0x8000u, mant = b & 0x7fffffu;
    const int e = (int)((b >> 23) & 0xff) - 127;
    if (((b >> 23) & 0xff) == 0 || e < -24) return (uint16_t)sign;
    if (e < -14) return (uint16_t)(sign | ((0x800000u | mant) >> (-e - 1)));
    return (uint16_t)(sign | ((uint32_t)(e + 15) << 10) | (mant >> 13));
}
static float f16_bits_to_f32(uint16_t hb) {
    const int e = (hb >> 10) & 0x1f; const uint32_t m = hb & 0x3ffu; float v;
    if (e == 0) v = ldexpf((float)m, -24); else v = ldexpf((float)(0x400u | m), e - 25);
    return (hb & 0x8000u) ? -v : v;
}
static void split2(float a, uint16_t p[2]) { p[0] = f16_rtz_bits(a); p[1] = f16_rtz_bits(a - f16_bits_to_f32(p[0])); }

static float cpu_dtw(const float *a, int m, const float *b, int n) {  // unit rows both sides; oracle/rp_oracle.c orc_dtw_banded
    std::vector<float> D((size_t)(m + 1) * (n + 1), INFINITY);
    D[0] = 0.f;
    for (int r = 1; r <= m; ++r) {
        int start = r - W > 1 ? r - W : 1, end = n + 1 < r + W ? n + 1 : r + W;
        for (int c = start; c < end; ++c) {
            float dot = 0.f;
            for (int k = 0; k < K; ++k) dot = fmaf(a[(r - 1) * K + k], b[(c - 1) * K + k], dot);
            float mn = fminf(fminf(D[(size_t)(r - 1) * (n + 1) + c], D[(size_t)r * (n + 1) + c - 1]), D[(size_t)(r - 1) * (n + 1) + c - 1]);
            D[(size_t)r * (n + 1) + c] = (1.f - dot) + mn;
        }
    }
    return D[(size_t)(m - 1) * (n + 1) + n];
}

int main(int argc, char **argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 100;
    const size_t S = argc > 2 ? (size_t)atol(argv[2]) : 8192;
    const int n_win = argc > 3 ? atoi(argv[3]) : 288;
    const int F = n_win + L - 1, T = 8;
    const float score_ref = 0.22f;
    srand(7);
    std::vector<float> mf(S * F * K), tm((size_t)T * L * K);
    for (auto &v : mf) v = 4.f * ((float)rand() / RAND_MAX - 0.5f);
    for (size_t i = 0; i < S * (size_t)F; ++i) mf[i * K] += 3.f;  // a mean to subtract
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < L; ++r) {
            float v[K], nn = 0.f;
            for (int k = 0; k < K; ++k) { v[k] = (float)rand() / RAND_MAX - 0.5f; nn += v[k] * v[k]; }
            for (int k = 0; k < K; ++k) tm[((size_t)t * L + r) * K + k] = v[k] / sqrtf(nn);
        }
    if (L > 3) for (int k = 0; k < K; ++k) tm[(size_t)3 * K + k] = 0.f;  // a zero row: cost 1
    // A image: [row][k half][template] x 8 f16 = the negated unit row, split, in the slot order of the B operand
    std::vector<uint16_t> img((size_t)(L + NS) * kRowBytes / 2, 0);
    for (int r = 0; r < L; ++r)
        for (int t = 0; t < T; ++t) {
            uint16_t p[K][2];
            for (int k = 0; k < K; ++k) split2(-tm[((size_t)t * L + r) * K + k], p[k]);
            for (int kh = 0; kh < 2; ++kh) {
                const int ca = kh ? 3 : 0, cb = ca + 1;
                uint16_t sl[8] = {p[ca][0], p[cb][0], p[ca][0], p[cb][0], p[ca][1], p[cb][1], 0, 0};
                if (kh == 0) { sl[6] = p[2][0]; sl[7] = p[2][0]; } else { sl[6] = p[2][1]; sl[7] = 0x3c00; }
                memcpy(&img[((size_t)r * kRowBytes + kh * 128 + t * 16) / 2], sl, 16);
            }
        }
    float *d_mf, *d_sc; u32x4 *d_img;
    hipMalloc(&d_mf, mf.size() * 4 + 4096); hipMalloc(&d_sc, S * n_win * 8 * 4); hipMalloc(&d_img, img.size() * 2);
    hipMemcpy(d_mf, mf.data(), mf.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemset(d_sc, 0xff, S * n_win * 8 * 4);
    const size_t total_tiles = S * (n_win / 32);
    const size_t lds = (size_t)(L + NS) * kRowBytes + (size_t)NWAVES * 2 * ((((34 + L) * K + 63) & ~63) * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int n_cu = 256; { hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount; }
    size_t blocks = (size_t)n_cu * WGS_PER_CU;
    if (blocks * NWAVES > total_tiles) blocks = (total_tiles + NWAVES - 1) / NWAVES;
    printf("L %d, %zu streams x %d windows x %d templates, %zu tiles, %zu workgroups of %d waves, %zu B of LDS\n", L, S, n_win, T, total_tiles, blocks, NWAVES, lds);
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(dtw_mfma_kernel, dim3((unsigned)blocks), dim3(64 * NWAVES), lds, 0, d_mf, F, n_win, S, L, d_img, score_ref, d_sc, total_tiles);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        // shipped dtw_band_kernel at C3: 19.5 ms for 19.46 M windows x 8 templates of 100 rows
        printf("rep %d: %.3f ms (%s)  = %.1f M windows/s x 8 templates; the shipped kernel's C3 rate gives %.3f ms for this many window-rows\n", rep, ms,
               hipGetErrorString(e), S * n_win / ms * 1e-3, 19.5 * (double)(S * n_win) / 19.46e6 * (double)L / 100.0);
    }
    std::vector<float> sc(S * n_win * 8);
    hipMemcpy(sc.data(), d_sc, sc.size() * 4, hipMemcpyDeviceToHost);
    // check a sample of windows against the CPU
    double worst = 0; int bad = 0, checked = 0;
    for (int it = 0; it < 400; ++it) {
        const size_t s = (size_t)rand() % S; const int w = it < 40 ? (it % 2 ? n_win - 1 - it / 2 : it / 2) : rand() % n_win;
        std::vector<float> x((size_t)L * K);
        float mu[K] = {0, 0, 0, 0, 0};
        for (int i = 0; i < L; ++i) for (int k = 0; k < K; ++k) mu[k] += mf[(s * F + w + i) * K + k];
        for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;
        for (int i = 0; i < L; ++i) {
            float d[K], bb = 0.f;
            for (int k = 0; k < K; ++k) { d[k] = mf[(s * F + w + i) * K + k] - mu[k]; bb = fmaf(d[k], d[k], bb); }
            const float inv = bb > 0.f ? 1.f / sqrtf(bb) : 0.f;
            for (int k = 0; k < K; ++k) x[(size_t)i * K + k] = d[k] * inv;
        }
        for (int t = 0; t < T; ++t) {
            const float cost = cpu_dtw(&tm[(size_t)t * L * K], L, x.data(), L);
            const float ref = 1.f / (1.f + expf((cost / (float)(2 * L) - score_ref) / score_ref));
            const float got = sc[(s * n_win + w) * 8 + t];
            const double err = fabs((double)got - ref) / fmax(fabs((double)ref), 1e-30);
            if (!(err <= 1e-5)) { if (bad < 10) printf("  MISMATCH stream %zu window %d template %d: got %.9g ref %.9g (cost %.6f)\n", s, w, t, got, ref, cost); ++bad; }
            if (err > worst) worst = err;
            ++checked;
        }
    }
    printf("checked %d scores against the CPU: worst relative error %.3g, %d beyond 1e-5\n", checked, worst, bad);
    return bad ? 1 : 0;
}
