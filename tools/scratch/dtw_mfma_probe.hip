// dtw_mfma_probe.hip -- prototype of the banded DTW with the cosine costs on the matrix cores.
//
// Shipped kernel: one lane per window, 5 v_pk_fma_f32 per band cell of a template pair = 62 % of its issue cycles, at the f32
// FMA peak of the vector pipe.  Here the costs of a whole band column come out of v_mfma_f32_32x32x16_bf16:
//   * a wave owns 32 windows x 8 templates; lane l = (window l & 31, half l >> 5) runs the recurrence of templates 4h..4h+3
//     (two packed pairs) of its window -- the MFMA's C/D layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
//     puts exactly those costs into that lane's registers, as consecutive register pairs;
//   * column-major sweep: step c takes window frame c against the 2W template rows of its band; M = 8 templates x 12 circular
//     row slots = 3 tiles of 32 rows, N = 32 windows, K = 32 bf16 slots = the 30 products of an exact three-way bf16 split
//     (x = x0 + x1 + x2, a = a0 + a1 + a2 by truncation, every partial product with i + j <= 2: error <= 2^-23 |a||x|),
//     accumulated in f32 on top of the inline constant 1.0: the instruction leaves 1 - a.x;
//   * the B operand (window frame, unit length) is built per step by the lanes (normalise, split, v_perm pack), the A operand
//     (negated unit template rows, split on the host) sits in LDS and one tile of it is refreshed per step.
// VALU per step and lane: ~40 (frame) + 60 (2 pairs x 10 cells x (2 v_min3 + v_pk_add)); 6 MFMAs run beside them.
//
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/dtw_mfma_probe.hip -o tools/scratch/dtw_mfma_probe && tools/scratch/dtw_mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 5, W = 5, B = 2 * W, NS = 12;  // NS circular row slots (3 tiles x 4)
constexpr int kRowBytes = 512;                   // A image: [row][kstep 2][khalf 2][template 8] x 16 B
#define RP_INF __builtin_inff()
#ifdef RP_RAW_RSQ
#define RP_RSQ(x) __builtin_amdgcn_rsqf(x)
#else
#define RP_RSQ(x) rsqrtf(x)
#endif

#ifndef RP_ABL
#define RP_ABL 0
#endif
#ifndef NWAVES
#define NWAVES 6
#endif
#ifndef WGS_PER_CU
#define WGS_PER_CU 2
#endif

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ unsigned hi16_pair(float odd, float even) {  // {bf16 trunc(even), bf16 trunc(odd)} = slots (e, e+1)
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }

// mfcc [S][F][K]; aimg [(L + 12)][2][2][8] x 16 B; scores [S * n_win][8]
__global__ __launch_bounds__(64 * NWAVES, WGS_PER_CU) void dtw_mfma_kernel(const float *__restrict__ mfcc, int F, int n_win, size_t n_streams,
                                                                           int L, const u32x4 *__restrict__ aimg, float score_ref,
                                                                           float *__restrict__ scores, size_t total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int a_bytes = (L + NS) * kRowBytes;
    const int xs_floats = ((32 + L) * K + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        u32x4 *adst = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < a_bytes / 16; i += 64 * NWAVES) adst[i] = aimg[i];
    }
    __syncthreads();
    float *xs = reinterpret_cast<float *>(smem + a_bytes) + wave * xs_floats;
    const int n = lane & 31, h = lane >> 5;
    // A operand: this lane supplies row m = lane & 31 of a tile = (slot 4g + jj, template 4h' + r'), k half = lane >> 5
    const int jj = (lane & 31) >> 3, tA = ((lane >> 2) & 1) * 4 + (lane & 3);
    const unsigned a_lane = (unsigned)(h * 128 + tA * 16);
    unsigned dl[4];  // byte offset back to the row this lane's slot holds when the newest row sits in slot e of its tile
#pragma unroll
    for (int e = 0; e < 4; ++e) dl[e] = (unsigned)(((e - jj + NS) % NS) * kRowBytes);
    const unsigned sel_one = h ? 0x03020706u : 0x07060706u;  // slots (4, 5) of k-step 1: (y2.p0, y2.p0) / (y2.p0, 1.0) -- the '1 -' of the cost
    const int tiles_per_stream = n_win / 32;

    for (size_t tile = (size_t)blockIdx.x * NWAVES + wave; tile < total_tiles; tile += (size_t)gridDim.x * NWAVES) {
        const size_t s = tile / tiles_per_stream;
        const int w0 = (int)(tile - s * tiles_per_stream) * 32;
        {
            const float *src = mfcc + (s * F + w0) * K;
            const int cnt = (32 + L) * K, room = (F - w0) * K;
            for (int i = lane; i < cnt; i += 64) xs[i] = i < room ? src[i] : 0.f;
        }
        wave_lds_sync();
        const float *xl = xs + n * K;
        float mu[K];
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] = 0.f;
        for (int i = 0; i < L; ++i) {
#pragma unroll
            for (int k = 0; k < K; ++k) mu[k] += xl[i * K + k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) mu[k] = mu[k] / (float)L;

        // Q[p][q] = D[(c - 1) - W + 1 + q][c - 1] of the template pair p (band position, as P[] of dtw_band_kernel with rows and
        // columns swapped); column 0: D[0][0] = 0 sits at q = W - 1.  Q[p][B] stays +inf (the cell below the band).
        v2f Q[2][B + 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = 0; q <= B; ++q) Q[p][q] = (v2f){RP_INF, RP_INF};
            Q[p][W - 1] = (v2f){0.f, 0.f};
        }
        // A tiles for the state "newest row R = W" (rows 1..W in slots 1..W; the other slots are never read unguarded)
        u32x4 Areg[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int slot = 4 * g + jj;
            int r = W - ((W - slot + NS) % NS);  // 1-based template row in this slot
            r = r < 1 ? 1 : r;
            const unsigned char *ap = smem + a_lane + (unsigned)(r - 1) * kRowBytes;
            Areg[g][0] = *reinterpret_cast<const u32x4 *>(ap);
            Areg[g][1] = *reinterpret_cast<const u32x4 *>(ap + 256);
        }
        v16f acc[2][3];  // costs of column c in acc[c & 1]: the MFMAs of column c + 1 run under the recurrence of column c

// the frame of column cc: normalise (as dtw_band_kernel), pick this half's components, split, pack; refresh the A tile that
// receives template row cc + W; six MFMAs into acc[par]
#define RP_ISSUE(cc, uu, par, GUARD)                                                                                          \
    do {                                                                                                                      \
        float d_[K], bb_ = 0.f;                                                                                               \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                                                       \
            d_[k] = (RP_ABL & 4) ? mu[k] : xl[((cc) - 1) * K + k] - mu[k];                                                    \
            bb_ = fmaf(d_[k], d_[k], bb_);                                                                                    \
        }                                                                                                                     \
        const float inv_ = (RP_ABL & 4) ? mu[0] : (bb_ > 0.f ? RP_RSQ(bb_) : 0.f);                                            \
        const float u0 = (h ? d_[3] : d_[0]) * inv_, u1 = (h ? d_[4] : d_[1]) * inv_, u2 = d_[2] * inv_;                      \
        const float u0r1 = u0 - trunc_bf16(u0), u0r2 = u0r1 - trunc_bf16(u0r1);                                               \
        const float u1r1 = u1 - trunc_bf16(u1), u1r2 = u1r1 - trunc_bf16(u1r1);                                               \
        const float u2r1 = u2 - trunc_bf16(u2), u2r2 = u2r1 - trunc_bf16(u2r1);                                               \
        u32x4 b0, b1;                                                                                                         \
        b0.x = hi16_pair(u0, u0);   b0.y = hi16_pair(u0r1, u0);   b0.z = hi16_pair(u0r2, u0r1); b0.w = hi16_pair(u1, u1);     \
        b1.x = hi16_pair(u1r1, u1); b1.y = hi16_pair(u1r2, u1r1); b1.z = __builtin_amdgcn_perm(__float_as_uint(u2), 0x3f800000u, sel_one); b1.w = hi16_pair(u2r2, u2r1); \
        {                                                                                                                     \
            const int sn = ((uu) + 1 + W) % NS, g = sn / 4, e = sn % 4;                                                       \
            int off = ((cc) + W - 1) * kRowBytes - (int)dl[e];                                                                \
            if (GUARD) off = off < 0 ? 0 : off;                                                                               \
            const unsigned char *ap = smem + a_lane + (unsigned)off;                                                          \
            Areg[g][0] = *reinterpret_cast<const u32x4 *>(ap);                                                                \
            Areg[g][1] = *reinterpret_cast<const u32x4 *>(ap + 256);                                                          \
        }                                                                                                                     \
        if (RP_ABL & 1) { /* no MFMA: costs are whatever the registers hold (wrong results) */                                \
            const unsigned x_ = b0.x ^ b0.y ^ b0.z ^ b0.w ^ b1.x ^ b1.y ^ b1.z ^ b1.w;                                        \
            _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[par][g][2 * g] = __uint_as_float((x_ ^ Areg[g][0].x ^ Areg[g][1].y) & 0x3fffffffu); \
        } else if (RP_ABL & 8) { /* k-step 0 only */                                                                           \
            _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                                   \
                const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          \
                acc[par][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g][0]), __builtin_bit_cast(bf16x8, b0 ^ b1), zero16, 0, 0, 0); \
            }                                                                                                                 \
        } else {                                                                                                              \
        _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                                       \
            const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              \
            acc[par][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g][0]), __builtin_bit_cast(bf16x8, b0), zero16, 0, 0, 0); \
        }                                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                                         \
            acc[par][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Areg[g][1]), __builtin_bit_cast(bf16x8, b1), acc[par][g], 0, 0, 0); \
        }                                                                                                                     \
    } while (0)

// columns c0 .. c0 + 11 (c0 = 1 mod 12): u is compile time, so every slot, tile and band index is a fixed register
#define RP_COL(GUARD)                                                                                                         \
    _Pragma("unroll") for (int u = 0; u < NS; ++u) {                                                                          \
        const int c = c0 + u;                                                                                                 \
        if (c <= L) {                                                                                                         \
            RP_ISSUE(c + 1, (u + 1) % NS, (u + 1) & 1, GUARD);                                                                \
            /* rows r_q = c - W + 1 + q, q = 0..2W-1, in MFMA row slot (u + q - 3) mod 12 since c = 1 + u (mod 12) */          \
            v2f up[2] = {(v2f){RP_INF, RP_INF}, (v2f){RP_INF, RP_INF}};                                                       \
            if (RP_ABL & 2) { _Pragma("unroll") for (int g = 0; g < 3; ++g) { Q[0][W - 2].x += acc[u & 1][g][0]; Q[1][W - 2].y += acc[u & 1][g][9]; } } \
            else _Pragma("unroll") for (int q = 0; q < B; ++q) {                                                              \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) { /* two independent chains, interleaved */                     \
                    const int sl = (u + q + NS - W + 2) % NS;                                                                 \
                    const v2f cost = (v2f){acc[u & 1][sl / 4][4 * (sl % 4) + 2 * p], acc[u & 1][sl / 4][4 * (sl % 4) + 2 * p + 1]}; \
                    v2f m;                                                                                                    \
                    m.x = fminf(fminf(up[p].x, Q[p][q + 1].x), Q[p][q].x);                                                    \
                    m.y = fminf(fminf(up[p].y, Q[p][q + 1].y), Q[p][q].y);                                                    \
                    v2f v = cost + m;                                                                                         \
                    if (GUARD) v = (c - W + 1 + q >= 1) ? v : (v2f){RP_INF, RP_INF};                                          \
                    Q[p][q] = v;                                                                                              \
                    up[p] = v;                                                                                                \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
    }

        RP_ISSUE(1, 0, 0, true);
        {
            const int c0 = 1;
            RP_COL(true)
        }
        for (int c0 = 1 + NS; c0 <= L; c0 += NS) { RP_COL(false) }
#undef RP_COL
#undef RP_ISSUE

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
        wave_lds_sync();  // the next tile restages xs
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
static uint16_t bf16_trunc_bits(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }
static float bf16_trunc(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
static void split3(float a, uint16_t p[3]) {
    const float p0 = bf16_trunc(a), r1 = a - p0, p1 = bf16_trunc(r1), r2 = r1 - p1;
    p[0] = bf16_trunc_bits(p0); p[1] = bf16_trunc_bits(p1); p[2] = bf16_trunc_bits(r2);
}

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
    // A image
    std::vector<uint16_t> img((size_t)(L + NS) * kRowBytes / 2, 0);
    for (int r = 0; r < L; ++r)
        for (int t = 0; t < T; ++t) {
            uint16_t p[K][3];
            for (int k = 0; k < K; ++k) split3(-tm[((size_t)t * L + r) * K + k], p[k]);
            for (int kh = 0; kh < 2; ++kh) {
                const int c0 = kh ? 3 : 0, c1 = kh ? 4 : 1;
                uint16_t s0[8] = {p[c0][0], p[c0][1], p[c0][2], p[c0][0], p[c0][1], p[c0][0], p[c1][0], p[c1][1]};
                uint16_t s1[8] = {p[c1][2], p[c1][0], p[c1][1], p[c1][0], 0, 0, 0, 0};
                if (kh == 0) { s1[4] = p[2][0]; s1[5] = p[2][1]; s1[6] = p[2][0]; s1[7] = p[2][0]; }
                else { s1[4] = p[2][2]; s1[5] = 0x3f80; s1[6] = p[2][1]; s1[7] = 0; }  // slot 5: 1.0 x 1.0, the constant of 1 - a.x
                memcpy(&img[((size_t)r * kRowBytes + 0 * 256 + kh * 128 + t * 16) / 2], s0, 16);
                memcpy(&img[((size_t)r * kRowBytes + 1 * 256 + kh * 128 + t * 16) / 2], s1, 16);
            }
        }
    float *d_mf, *d_sc; u32x4 *d_img;
    hipMalloc(&d_mf, mf.size() * 4 + 4096); hipMalloc(&d_sc, S * n_win * 8 * 4); hipMalloc(&d_img, img.size() * 2);
    hipMemcpy(d_mf, mf.data(), mf.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipMemset(d_sc, 0xff, S * n_win * 8 * 4);
    const size_t total_tiles = S * (n_win / 32);
    const size_t lds = (size_t)(L + NS) * kRowBytes + (size_t)NWAVES * ((((32 + L) * K + 3) & ~3) * 4);
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
