// rp_mlp.hip -- the wakeword model: forward (src/wakewords/nn/wakeword_nn.rs:101-163,305-389; mlp_layer_kernel,
// mlp_mfma_kernel on the matrix cores, normalize_windows_kernel) and training (wakeword_model_train.rs:204-209).
// DESIGN.md §4.4.
#include "rp_device.h"

namespace rp {

// -------------------------------------------------------------------------- MLP
// Linear (x.W^T + b, W [out][in]) + optional ReLU, f32, k-ordered accumulation like
// candle's CPU gemm restated in the oracle.  One wave per (row, 64 outputs) tile with
// the input row staged in LDS.  (Round-1 correctness path; profiles/HISTORY.md lists the MFMA
// bf16 path for BASELINE config 5 as next.)
__global__ __launch_bounds__(64) void mlp_layer_kernel(const float *__restrict__ x, size_t B, int in, int on,
                                                       const float *__restrict__ Wt, const float *__restrict__ bias,
                                                       int relu, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xr = reinterpret_cast<float *>(smem);
    const size_t b = blockIdx.x;
    const int o = blockIdx.y * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < in; i += 64) xr[i] = x[b * in + i];
    __syncthreads();
    if (o >= on) return;
    const float *w = Wt + (size_t)o * in;
    float s = 0.f;
    for (int i = 0; i < in; ++i) s += xr[i] * w[i];
    s += bias[o];
    if (relu && s < 0.f) s = 0.f;
    out[b * on + o] = s;
}

__global__ __launch_bounds__(64) void normalize_windows_kernel(const float *__restrict__ mfcc, size_t first_win,
                                                                size_t n_win, int L, int K, float *__restrict__ x) {
    const size_t w = blockIdx.x;
    const float *src = mfcc + (first_win + w) * K;
    float *dst = x + w * (size_t)L * K;
    for (int k = threadIdx.x; k < K; k += 64) {
        float sum = 0.f;
        for (int i = 0; i < L; ++i) sum += src[(size_t)i * K + k];
        for (int i = 0; i < L; ++i) dst[(size_t)i * K + k] = src[(size_t)i * K + k] - sum / (float)L;
    }
}

hipError_t launch_normalize_windows(hipStream_t st, const float *mfcc, size_t first_win, size_t n_win, int L, int K,
                                    float *x) {
    if (n_win == 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_windows_kernel, dim3((unsigned)n_win), dim3(64), 0, st, mfcc, first_win, n_win, L, K, x);
    return hipGetLastError();
}

// the same for windows of many streams: row = s * n_win + w
__global__ __launch_bounds__(64) void normalize_windows_batch_kernel(const float *__restrict__ mfcc, size_t frame_pitch, size_t n_win,
                                                                      size_t first_row, int L, int K, float *__restrict__ x) {
    const size_t row = first_row + blockIdx.x;
    const size_t s = row / n_win, w = row - s * n_win;
    const float *src = mfcc + (s * frame_pitch + w) * K;
    float *dst = x + (size_t)blockIdx.x * L * K;
    for (int k = threadIdx.x; k < K; k += 64) {
        float sum = 0.f;
        for (int i = 0; i < L; ++i) sum += src[(size_t)i * K + k];
        for (int i = 0; i < L; ++i) dst[(size_t)i * K + k] = src[(size_t)i * K + k] - sum / (float)L;
    }
}

hipError_t launch_normalize_windows_batch(hipStream_t st, const float *mfcc, size_t frame_pitch, size_t n_win, size_t first_row,
                                          size_t n_rows, int L, int K, float *x) {
    if (n_rows == 0) return hipSuccess;
    if (n_rows > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(normalize_windows_batch_kernel, dim3((unsigned)n_rows), dim3(64), 0, st, mfcc, frame_pitch, n_win, first_row, L, K, x);
    return hipGetLastError();
}

// calc_inverse_similarity, wakeword_nn.rs:161-163
__device__ __forceinline__ float nn_inverse_similarity(float n1, float n2, float reference) {
    return 1.f - (1.f / (1.f + expf(((n1 - n2) - reference) / reference)));
}

// get_label (:47-60: max_by(total_cmp) keeps the LAST maximum), run_detection_by_label (:61-99: second_prob is the
// smallest other logit), validate_scores (:113-123)
__global__ __launch_bounds__(256) void nn_score_kernel(const float *__restrict__ logits, size_t n_rows, int nl, int none_index,
                                                       float ref, int calc_avg, float threshold, float avg_threshold,
                                                       float *__restrict__ agg, float *__restrict__ avg, int32_t *__restrict__ label,
                                                       uint32_t *__restrict__ hot, size_t rows_per_stream) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const float *lg = logits + r * nl;
    int bi = 0;
    for (int i = 1; i < nl; ++i) if (!(lg[i] < lg[bi])) bi = i;
    float score = -2.f, av = 0.f;
    if (bi != none_index) {
        const float label_prob = lg[bi], none_prob = none_index >= 0 ? lg[none_index] : 0.f;
        float second = 0.f;
        if (calc_avg) {
            bool any = false;
            for (int i = 0; i < nl; ++i) {
                if (lg[i] == label_prob) continue;
                if (!any || !(lg[i] > second)) { second = lg[i]; any = true; }
            }
            if (!any) second = 0.f;
            av = nn_inverse_similarity(label_prob, second, ref);
        }
        const float sc = nn_inverse_similarity(label_prob, none_prob, ref);
        if (sc >= threshold && av >= avg_threshold) score = sc;
    }
    agg[r] = score; avg[r] = av; label[r] = bi;
    // a stream with no window that passed cannot fire (detector.rs:411-429): the scan skips it (every writer stores the same value)
    if (hot && score != -2.f) hot[r / rows_per_stream] = 1u;
}

hipError_t launch_nn_score(hipStream_t st, const float *logits, size_t n_rows, int n_labels, int none_index, float score_ref10,
                           int calc_avg, float threshold, float avg_threshold, float *agg, float *avg, int32_t *label, uint32_t *hot,
                           size_t rows_per_stream) {
    if (n_rows == 0) return hipSuccess;
    if (hot && rows_per_stream == 0) return hipErrorInvalidValue;
    const size_t blocks = (n_rows + 255) / 256;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nn_score_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, n_rows, n_labels, none_index, score_ref10, calc_avg,
                       threshold, avg_threshold, agg, avg, label, hot, rows_per_stream);
    return hipGetLastError();
}

hipError_t launch_mlp(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                      float *const *Bv, float *scratch0, float *scratch1, float *out) {
    if (B == 0) return hipSuccess;
    const float *cur = x;
    float *bufs[2] = {scratch0, scratch1};
    for (int l = 0; l < n_layers; ++l) {
        float *dst = (l + 1 == n_layers) ? out : bufs[l & 1];
        dim3 grid((unsigned)B, (unsigned)((dims[l + 1] + 63) / 64));
        size_t lds = (size_t)dims[l] * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(mlp_layer_kernel, grid, dim3(64), lds, st, cur, B, dims[l], dims[l + 1], W[l], Bv[l],
                           l + 1 < n_layers ? 1 : 0, dst);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        cur = dst;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------ MLP training
// WakewordModelTrain's loop (src/wakewords/nn/wakeword_model_train.rs:204-209): full-batch forward,
// log_softmax + nll (mean over the batch), backward, plain SGD.  The matrices are tiny (tens of recordings x a few
// thousand features): one thread per result element, reductions along the batch / the layer width in a loop.
__global__ __launch_bounds__(64) void train_forward_kernel(const float *__restrict__ x, size_t B, int in, int on,
                                                           const float *__restrict__ W, const float *__restrict__ bias, int relu,
                                                           float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xr = reinterpret_cast<float *>(smem);
    const size_t b = blockIdx.x;
    const int o = blockIdx.y * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < in; i += 64) xr[i] = x[b * in + i];
    __syncthreads();
    if (o >= on) return;
    const float *w = W + (size_t)o * in;
    float s = 0.f;
    for (int i = 0; i < in; ++i) s += xr[i] * w[i];
    s += bias[o];
    if (relu && s < 0.f) s = 0.f;
    out[b * on + o] = s;
}

// per row: log_softmax (x - max - ln(sum exp(x - max))), loss_row = -log_sm[label], dlogits = (softmax - onehot) / B
__global__ __launch_bounds__(64) void train_softmax_grad_kernel(const float *__restrict__ logits, const int32_t *__restrict__ labels,
                                                                size_t B, int C, float *__restrict__ dz, float *__restrict__ loss_rows) {
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float *x = logits + b * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float lse = logf(se);
    const int lab = labels[b];
    for (int c = 0; c < C; ++c) {
        const float lsm = (x[c] - mx) - lse;
        if (c == lab) loss_rows[b] = -lsm;
        dz[b * C + c] = (expf(lsm) - (c == lab ? 1.f : 0.f)) / (float)B;
    }
}

// dZprev[b][i] = A_prev[b][i] > 0 ? sum_o dZ[b][o] * W[o][i] : 0     (ReLU backward through the layer's input)
__global__ __launch_bounds__(256) void train_backprop_kernel(const float *__restrict__ dz, const float *__restrict__ W,
                                                             const float *__restrict__ a_prev, size_t B, int in, int on,
                                                             float *__restrict__ dz_prev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= in) return;
    float s = 0.f;
    for (int o = 0; o < on; ++o) s += dz[b * on + o] * W[(size_t)o * in + i];
    dz_prev[b * in + i] = a_prev[b * in + i] > 0.f ? s : 0.f;
}

// SGD step of one layer: W[o][i] -= lr * sum_b dZ[b][o] * A_in[b][i];  bias[o] -= lr * sum_b dZ[b][o]
__global__ __launch_bounds__(256) void train_update_kernel(const float *__restrict__ dz, const float *__restrict__ a_in, size_t B,
                                                           int in, int on, float lr, float *__restrict__ W, float *__restrict__ bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (i > in) return;  // i == in: the bias column
    float g = 0.f;
    if (i < in) {
        for (size_t b = 0; b < B; ++b) g += dz[b * on + o] * a_in[b * in + i];
        W[(size_t)o * in + i] = W[(size_t)o * in + i] - g * lr;
    } else {
        for (size_t b = 0; b < B; ++b) g += dz[b * on + o];
        bias[o] = bias[o] - g * lr;
    }
}

hipError_t launch_train_forward(hipStream_t st, const float *x, size_t B, int n_layers, const int *dims, float *const *W,
                                float *const *Bv, float *const *act) {
    if (B == 0) return hipSuccess;
    const float *cur = x;
    for (int l = 0; l < n_layers; ++l) {
        dim3 grid((unsigned)B, (unsigned)((dims[l + 1] + 63) / 64));
        const size_t lds = (size_t)dims[l] * sizeof(float);
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(train_forward_kernel, grid, dim3(64), lds, st, cur, B, dims[l], dims[l + 1], W[l], Bv[l],
                           l + 1 < n_layers ? 1 : 0, act[l]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        cur = act[l];
    }
    return hipSuccess;
}

// act[l] = output of layer l (post-ReLU for hidden layers, logits for the last); dz[l] same shapes
hipError_t launch_train_step(hipStream_t st, const float *x, const int32_t *labels, size_t B, int n_layers, const int *dims,
                             float *const *W, float *const *Bv, float *const *act, float *const *dz, float lr, float *loss_rows) {
    if (B == 0) return hipSuccess;
    hipError_t e = launch_train_forward(st, x, B, n_layers, dims, W, Bv, act);
    if (e != hipSuccess) return e;
    const int C = dims[n_layers];
    hipLaunchKernelGGL(train_softmax_grad_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, act[n_layers - 1], labels, B, C,
                       dz[n_layers - 1], loss_rows);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    for (int l = n_layers - 1; l >= 0; --l) {
        const int in = dims[l], on = dims[l + 1];
        const float *a_in = l == 0 ? x : act[l - 1];
        if (l > 0) {  // through W_l as it was in the forward pass, before its own update
            hipLaunchKernelGGL(train_backprop_kernel, dim3((unsigned)((in + 255) / 256), (unsigned)B), dim3(256), 0, st, dz[l], W[l],
                               act[l - 1], B, in, on, dz[l - 1]);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(train_update_kernel, dim3((unsigned)((in + 1 + 255) / 256), (unsigned)on), dim3(256), 0, st, dz[l], a_in, B, in,
                           on, lr, W[l], Bv[l]);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------ MLP on MFMA
// One workgroup = 8 waves = 128 rows; one wave = one 16-row tile x all layer-1 outputs (NT
// 16-column tiles).  The layer-1 weights are walked in k-groups of 128: the group's [16*NT][128]
// slice is staged once per workgroup in LDS and shared by the 8 waves (reading it per wave from L2
// cost more than the HBM stream itself: 0.39 -> 0.18 ms when removed), rows stream from HBM in
// the MFMA A-operand layout (16 rows x 64 B per instruction; 5.5 TB/s measured on its own).
//  f32 variant:  v_mfma_f32_16x16x4_f32, exact f32 (each output is a k-ordered fmaf chain).  A lane
//                loads 16 bytes of its row per 16-k block and feeds component j to MFMA step j; the
//                weight lane does the same, so both sides agree on the (permuted) k order.
//  bf16 variant: v_mfma_f32_16x16x32_bf16, inputs rounded to bf16 (RNE) in registers, f32 accumulate.
// The tail layers (<= 130 x 32 weights) run per row from LDS.  HBM-bound by construction: 4*in bytes
// per row against 2*in*N1 flops (12 480 B/row, ceiling 0.64 G rows/s at 8 TB/s).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int kMlpWaves = 8;
constexpr int kMlpRowsPerWave = 16;
constexpr int kMlpKG = 128;  // k-group staged per step

__host__ __device__ constexpr int mlp_wpitch_f32() { return kMlpKG + 4; }   // floats per staged weight row
__host__ __device__ constexpr int mlp_wpitch_bf16() { return kMlpKG + 8; }  // bf16 per staged weight row

template <int NT, int PREC>
__global__ __launch_bounds__(64 * kMlpWaves, 3) void mlp_mfma_kernel(
    const float *__restrict__ x, size_t B, int in, int kpad, const float *__restrict__ w1f,
    const __bf16 *__restrict__ w1h, const float *__restrict__ b1, const float *__restrict__ tail, int tail_floats,
    int n_layers, int d1, int d2, int d3, int d4, int h2w, int wbuf_floats, float *__restrict__ out, size_t row_stride,
    size_t rows_per_stream, size_t stream_skip, const float *__restrict__ mean, const float *__restrict__ wsum, int K, uint32_t *redo,
    int redo_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int N1P = 16 * NT;
    // redo (rp_kernels.h, MlpRedo): kMlpF16x2 lists the rows that hold a feature beyond the f16 range (redo[0] = rows listed, from
    // redo[2] on their indices); redo_list != 0: this launch computes exactly those rows (the f32 matrix instructions) and the last
    // workgroup to leave puts redo[0] and redo[1] back to zero
    const uint32_t *rlist = nullptr;
    if (redo_list) {
        const size_t n = redo[0];
        if (n == 0) return;   // nothing listed (every call on ordinary features): no workgroup touches the counters
        if ((size_t)blockIdx.x * kMlpWaves * kMlpRowsPerWave >= n) {   // the whole workgroup: nothing (more) listed
            if (threadIdx.x == 0 && atomicAdd(redo + 1, 1u) == gridDim.x - 1) { redo[0] = 0; redo[1] = 0; }
            return;
        }
        B = n;
        rlist = redo + 2;
    }
    auto row_of = [&](size_t i) -> size_t { return rlist ? (size_t)rlist[i] : i; };   // position in this launch -> row of x / out
    float *tl = reinterpret_cast<float *>(smem);                       // tail weights
    float *wbuf = tl + ((tail_floats + 3) & ~3);                       // staged weight group; later h1 [waves][16][N1P+1]
    float *h2_all = wbuf + wbuf_floats;                                // [waves][16][h2w]
    for (int i = threadIdx.x; i < tail_floats; i += blockDim.x) tl[i] = tail[i];

    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int li = l & 15, lk = l >> 4;
    const size_t row0 = ((size_t)blockIdx.x * kMlpWaves + wave) * kMlpRowsPerWave;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t r = row0 + li;
    if (r >= B) r = B - 1;  // rows past the end recompute the last row; their results are dropped
    r = row_of(r);
    // row_stride != 0: rows are overlapping windows read in place from the frame array (launch_mlp_mfma_windows)
    const float *xr = row_stride ? x + r * row_stride + (r / rows_per_stream) * stream_skip : x + r * in;
    float rng = 0.f;   // kMlpF16x2: largest |feature| this lane has seen

    // staged weight groups are double buffered when they fit (NT <= 2): the next group's global
    // loads are issued before this group's MFMAs and land in the other buffer, one barrier per group
    constexpr bool DB = NT <= 2;
    constexpr int PF = mlp_wpitch_f32(), PH = mlp_wpitch_bf16();
    // 16-byte pieces of a staged weight group: f32 four k each, bf16 eight, kMlpF16x2 eight in each of its two planes
    constexpr int NPL = PREC == kMlpBf16x3 ? 3 : 2;   // planes of a split form's weights: w1h = [NPL][N1P][kpad] (MlpDev::w1t / w1s)
    constexpr int NPIECE = PREC == kMlpF32 ? N1P * (kMlpKG / 4) : PREC == kMlpBf16 ? N1P * (kMlpKG / 8) : NPL * N1P * (kMlpKG / 8);
    constexpr int NV = (NPIECE + 64 * kMlpWaves - 1) / (64 * kMlpWaves);
    const int half = DB ? wbuf_floats / 2 : 0;
    // every thread moves NV pieces; when the slice is a whole number of pieces per thread the bounds test is dropped
    // (a conditional store into wreg makes the compiler keep the array in scratch memory)
    constexpr bool FULL = NPIECE % (64 * kMlpWaves) == 0;
    // one 16-byte piece = 4 f32 or 8 bf16; plain vector values (HIP's float4 struct is copied with memcpy, which keeps
    // the staging array in scratch memory)
    auto wload = [&](int g, f32x4 (&wreg)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kMlpWaves;
            if (PREC == kMlpF32) {
                const int o = i / (kMlpKG / 4), c = i - o * (kMlpKG / 4);
                if (FULL || o < N1P) wreg[v] = *reinterpret_cast<const f32x4 *>(w1f + (size_t)o * kpad + g * kMlpKG + 4 * c);
            } else if (PREC == kMlpBf16) {
                const int o = i / (kMlpKG / 8), c = i - o * (kMlpKG / 8);
                if (FULL || o < N1P) wreg[v] = *reinterpret_cast<const f32x4 *>(w1h + (size_t)o * kpad + g * kMlpKG + 8 * c);
            } else {   // kMlpF16x2: plane p of w1h = [2][N1P][kpad] f16 (MlpDev::w1s)
                const int po = i / (kMlpKG / 8), c = i - po * (kMlpKG / 8);
                if (FULL || po < NPL * N1P) wreg[v] = *reinterpret_cast<const f32x4 *>(w1h + (size_t)po * kpad + g * kMlpKG + 8 * c);
            }
        }
    };
    auto wstore = [&](float *dstbuf, const f32x4 (&wreg)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = threadIdx.x + v * 64 * kMlpWaves;
            if (PREC == kMlpF32) {
                const int o = i / (kMlpKG / 4), c = i - o * (kMlpKG / 4);
                if (FULL || o < N1P) *reinterpret_cast<f32x4 *>(dstbuf + o * PF + 4 * c) = wreg[v];
            } else if (PREC == kMlpBf16) {
                const int o = i / (kMlpKG / 8), c = i - o * (kMlpKG / 8);
                if (FULL || o < N1P) *reinterpret_cast<f32x4 *>(reinterpret_cast<__bf16 *>(dstbuf) + o * PH + 8 * c) = wreg[v];
            } else {   // both planes, rows po = plane * N1P + o at the bf16 pitch
                const int po = i / (kMlpKG / 8), c = i - po * (kMlpKG / 8);
                if (FULL || po < NPL * N1P) *reinterpret_cast<f32x4 *>(reinterpret_cast<__bf16 *>(dstbuf) + po * PH + 8 * c) = wreg[v];
            }
        }
    };
    const int ngrp = kpad / kMlpKG;
    // the rows of k-group g+1 are requested before the MFMAs of group g (registers double buffered like the staged
    // weights), so the HBM stream never drains at the per-group barrier
    constexpr int NA = PREC == kMlpF32 ? kMlpKG / 16 : 2 * (kMlpKG / 32);
    float4 areg[2][NA];
    // Windows (mean != nullptr): the row's own window mean leaves each feature as it is loaded -- MfccNormalizer::normalize itself.  For
    // mfcc_size 16 a lane's pieces always hold the same coefficients (k0 mod 16 = 4 lk, or 8 (lk & 1) + 4 (u & 1)): registers; for the
    // other sizes (multiples of 4) the piece's place in the frame, kc = k0 mod K, is carried from group to group and the four means come
    // from the row's mean vector (a cached 16-byte load).  Until round 4 the mean was taken out after layer 1
    // (W.(f - mu) = W.f - sum_k mu[k] wsum[k]): real MFCCs sit on offsets many times their spread, and a sum that carries the offsets loses
    // their size in f32 rounding (5e-6 of a score on the reference's own recording).
    const bool norm = mean != nullptr, norm16 = norm && K == 16;
    float4 mu_a = make_float4(0.f, 0.f, 0.f, 0.f), mu_b = mu_a;
    const float *mrow = norm ? mean + r * (size_t)K : nullptr;
    int kc[NA];
    const int kstep = (norm && !norm16) ? kMlpKG % K : 0;
#pragma unroll
    for (int u = 0; u < NA; ++u) kc[u] = 0;
    if (norm16) {
        const float4 *mu = reinterpret_cast<const float4 *>(mrow);
        if (PREC == kMlpF32) mu_a = mu[lk];
        else { mu_a = mu[2 * (lk & 1)]; mu_b = mu[2 * (lk & 1) + 1]; }
    } else if (norm) {
#pragma unroll
        for (int u = 0; u < NA; ++u) kc[u] = (PREC == kMlpF32 ? 16 * u + 4 * lk : 32 * (u >> 1) + 8 * lk + 4 * (u & 1)) % K;
    }
    auto aload = [&](int g, float4 (&dst)[NA]) __attribute__((always_inline)) {   // called for g = 0, 1, 2, .. in this order
        const int kg = g * kMlpKG;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            // f32: piece u = k-step u (16 wide), lane part 4*lk;  bf16: pieces 2u / 2u+1 = low / high half of the lane's 8
            const int k0 = PREC == kMlpF32 ? kg + 16 * u + 4 * lk : kg + 32 * (u >> 1) + 8 * lk + 4 * (u & 1);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + 3 < in) {
                v = *reinterpret_cast<const float4 *>(xr + k0);
                if (norm) {
                    const float4 m = norm16 ? ((PREC == kMlpF32 || !(u & 1)) ? mu_a : mu_b) : *reinterpret_cast<const float4 *>(mrow + kc[u]);
                    v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
                }
            }
            dst[u] = v;
            if (norm && !norm16) { kc[u] += kstep; if (kc[u] >= K) kc[u] -= K; }
        }
    };
    {
        f32x4 w0[NV];
        wload(0, w0);
        aload(0, areg[0]);
        wstore(wbuf, w0);
    }
    __syncthreads();  // group 0 staged (and the tail weights landed)
    auto group = [&](int g, float4 (&a)[NA], float4 (&anext)[NA]) __attribute__((always_inline)) {
        const float *cur = wbuf + ((DB && (g & 1)) ? half : 0);
        // The next group's requests go out after the first k-step of this one: that step waits for this group's rows
        // (requested one group ago) while nothing newer is outstanding, so the wait never covers the new requests.
        f32x4 wnext[NV];
        auto prefetch = [&]() __attribute__((always_inline)) { if (g + 1 < ngrp) { wload(g + 1, wnext); aload(g + 1, anext); } };
        if (PREC == kMlpF32) {
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                float4 b[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const float4 *>(cur + (16 * n + li) * PF + 16 * u + 4 * lk);
                // the accumulators take turns so that consecutive MFMAs do not wait on each other's result
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[n].x, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[n].y, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[n].z, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[n].w, acc[n], 0, 0, 0);
                if (u == 0) prefetch();
            }
        } else if (PREC == kMlpF16x2) {
            // f16 two-way splits: x0 = rtz_f16(x) (as f32: 13 mantissa bits cleared), x1 = rtz_f16(x - x0); x0 w0 + x1 w0 + x0 w1
            // on the f16 matrix instruction, f32 accumulate.  A row with a feature beyond the f16 range is listed and computed again by the
            // f32 matrix instructions (one v_max_f32 |x| per feature finds it).
            const __bf16 *wb = reinterpret_cast<const __bf16 *>(cur);
#pragma unroll
            for (int u = 0; u < NA / 2; ++u) {
                const float4 lo = a[2 * u], hi = a[2 * u + 1];
                const float xs[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                unsigned h0[4], h1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = xs[2 * e], q = xs[2 * e + 1];
                    rng = fmaxf(fmaxf(rng, fabsf(p)), fabsf(q));
                    h0[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p, q));
                    h1[e] = pk_f16_second(p - __uint_as_float(__float_as_uint(p) & 0xffffe000u), q - __uint_as_float(__float_as_uint(q) & 0xffffe000u));
                }
                const f16x8 av0 = __builtin_bit_cast(f16x8, (u32x4v){h0[0], h0[1], h0[2], h0[3]});
                const f16x8 av1 = __builtin_bit_cast(f16x8, (u32x4v){h1[0], h1[1], h1[2], h1[3]});
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f16x8 b0 = *reinterpret_cast<const f16x8 *>(wb + (16 * n + li) * PH + 32 * u + 8 * lk);
                    const f16x8 b1v = *reinterpret_cast<const f16x8 *>(wb + (N1P + 16 * n + li) * PH + 32 * u + 8 * lk);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av0, b0, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av1, b0, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av0, b1v, acc[n], 0, 0, 0);
                }
                if (u == 0) prefetch();
            }
        } else if (PREC == kMlpBf16x3) {
            // three bf16 parts per operand (exact), six of the nine partial products, f32 accumulate: f32-grade (RP_MLP_F32; rp_mlp_stream.hip
            // has the same arithmetic).  No range limit: nothing is listed.
            const __bf16 *wb = reinterpret_cast<const __bf16 *>(cur);
#pragma unroll
            for (int u = 0; u < NA / 2; ++u) {
                const float4 lo = a[2 * u], hi = a[2 * u + 1];
                const float xs[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                unsigned h0[4], h1[4], h2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = xs[2 * e], q = xs[2 * e + 1];
                    const float rp = p - __uint_as_float(__float_as_uint(p) & 0xffff0000u), rq = q - __uint_as_float(__float_as_uint(q) & 0xffff0000u);
                    h0[e] = __builtin_amdgcn_perm(__float_as_uint(q), __float_as_uint(p), 0x07060302u);
                    h1[e] = __builtin_amdgcn_perm(__float_as_uint(rq), __float_as_uint(rp), 0x07060302u);
                    h2[e] = __builtin_amdgcn_perm(__float_as_uint(rq - __uint_as_float(__float_as_uint(rq) & 0xffff0000u)),
                                                  __float_as_uint(rp - __uint_as_float(__float_as_uint(rp) & 0xffff0000u)), 0x07060302u);
                }
                const bf16x8 av0 = __builtin_bit_cast(bf16x8, (u32x4v){h0[0], h0[1], h0[2], h0[3]});
                const bf16x8 av1 = __builtin_bit_cast(bf16x8, (u32x4v){h1[0], h1[1], h1[2], h1[3]});
                const bf16x8 av2 = __builtin_bit_cast(bf16x8, (u32x4v){h2[0], h2[1], h2[2], h2[3]});
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(wb + (16 * n + li) * PH + 32 * u + 8 * lk);
                    const bf16x8 b1v = *reinterpret_cast<const bf16x8 *>(wb + (N1P + 16 * n + li) * PH + 32 * u + 8 * lk);
                    const bf16x8 b2v = *reinterpret_cast<const bf16x8 *>(wb + (2 * N1P + 16 * n + li) * PH + 32 * u + 8 * lk);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av0, b2v, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av1, b1v, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av2, b0, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av0, b1v, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av1, b0, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av0, b0, acc[n], 0, 0, 0);
                }
                if (u == 0) prefetch();
            }
        } else {
            const __bf16 *wb = reinterpret_cast<const __bf16 *>(cur);
#pragma unroll
            for (int u = 0; u < NA / 2; ++u) {
                const float4 lo = a[2 * u], hi = a[2 * u + 1];
                bf16x8 av;
                av[0] = (__bf16)lo.x; av[1] = (__bf16)lo.y; av[2] = (__bf16)lo.z; av[3] = (__bf16)lo.w;
                av[4] = (__bf16)hi.x; av[5] = (__bf16)hi.y; av[6] = (__bf16)hi.z; av[7] = (__bf16)hi.w;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wb + (16 * n + li) * PH + 32 * u + 8 * lk);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acc[n], 0, 0, 0);
                }
                if (u == 0) prefetch();
            }
        }
        if (g + 1 < ngrp) {
            if (!DB) __syncthreads();  // single buffer: everyone must be done reading before the overwrite
            wstore(wbuf + ((DB && !(g & 1)) ? half : 0), wnext);
        }
        __syncthreads();
    };
    for (int g = 0; g < ngrp; g += 2) {
        group(g, areg[0], areg[1]);
        if (g + 1 < ngrp) group(g + 1, areg[1], areg[0]);
    }
    // every wave is past the last barrier, i.e. done with the staged weights: the buffer becomes h1
    // ---- layer-1 bias (+ReLU) -> LDS, C/D layout: col = lane&15, row = (lane>>4)*4 + reg
    float *h1 = wbuf + wave * kMlpRowsPerWave * (N1P + 1);
    const bool relu1 = n_layers > 1;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[n][e];
            v += b1[16 * n + li];
            if (relu1 && v < 0.f) v = 0.f;
            h1[(4 * lk + e) * (N1P + 1) + 16 * n + li] = v;
        }
    wave_lds_sync();
    // ---- tail layers: lane = (row l&15, output phase l>>4), outputs strided by 4 over the phases
    {
        const int rr = l & 15, ph = l >> 4;
        const bool row_ok = row0 + rr < B;
        const size_t orow = row_ok ? row_of(row0 + rr) : 0;
        if (PREC == kMlpF16x2) {   // the four lanes (row rr, k part 0..3) of a row agree on its range; one of them lists it
            rng = fmaxf(rng, __shfl_xor(rng, 16));
            rng = fmaxf(rng, __shfl_xor(rng, 32));
            if (ph == 0 && row_ok && !(rng <= 65504.f)) mlp_redo_append(redo, (uint32_t)orow, B);
        }
        const float *hin = h1 + rr * (N1P + 1);
        float *h2 = h2_all + (wave * kMlpRowsPerWave + rr) * h2w;
        const int dd[5] = {in, d1, d2, d3, d4};
        const float *wp = tl;
        int cur_in = d1;
        float *dst = out + orow * (size_t)dd[n_layers];
        if (n_layers == 1 && row_ok)
            for (int o = ph; o < d1; o += 4) dst[o] = hin[o];
        for (int layer = 1; layer < n_layers; ++layer) {
            const int on = dd[layer + 1];
            const bool last = layer + 1 == n_layers;
            for (int o = ph; o < on; o += 4) {
                const float *wr = wp + (size_t)o * cur_in;
                float s0 = 0.f, s1 = 0.f;
                int i = 0;
                for (; i + 1 < cur_in; i += 2) { s0 = fmaf(hin[i], wr[i], s0); s1 = fmaf(hin[i + 1], wr[i + 1], s1); }
                if (i < cur_in) s0 = fmaf(hin[i], wr[i], s0);
                float sacc = (s0 + s1) + wp[(size_t)on * cur_in + o];
                if (!last && sacc < 0.f) sacc = 0.f;
                if (last) { if (row_ok) dst[o] = sacc; } else h2[o] = sacc;
            }
            wave_lds_sync();  // the hidden layer is complete before anyone reads it
            wp += (size_t)on * cur_in + on;
            cur_in = on;
            hin = h2;  // n_layers <= 3: at most one hidden tail layer
        }
    }
    if (redo_list) {
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(redo + 1, 1u) == gridDim.x - 1) { redo[0] = 0; redo[1] = 0; }
    }
}

template <int NT>
static hipError_t launch_mlp_nt(hipStream_t st, const MlpDev &m, const float *x, size_t B, int precision, float *out, uint32_t *redo,
                                size_t row_stride = 0, size_t rows_per_stream = 1, size_t stream_skip = 0, const float *mean = nullptr,
                                const float *wsum = nullptr, int K = 0) {
    const size_t rows_per_block = (size_t)kMlpWaves * kMlpRowsPerWave;
    const size_t blocks = (B + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    int h2w = 1;
    for (int l2 = 2; l2 < m.n_layers; ++l2) h2w = m.dims[l2] + 1 > h2w ? m.dims[l2] + 1 : h2w;
    h2w |= 1;
    static_assert(mlp_wpitch_bf16() >= mlp_wpitch_f32(), "staged weight group");
    // the caller's precision (rp_kernels.h): kMlpF32 = RP_MLP_F32 = f32-grade products on the bf16 matrix instruction (three exact parts per
    // operand, kMlpBf16x3); kMlpF16x2 = RP_MLP_F32_FAST (two f16 parts, 22-bit, + the f32 pass on listed rows); kMlpStrictF32 = the f32 matrix
    // instructions for every row
    if (precision == kMlpStrictF32) precision = kMlpF32;
    else if (precision == kMlpF32 && m.w1t) precision = kMlpBf16x3;
    else if (precision == kMlpF16x2 && !m.w1s) precision = kMlpF32;
    // staged weight group(s): 16 NT rows x the bf16 pitch, in floats, for the two f16 planes of kMlpF16x2 (f32 and bf16 groups are smaller);
    // the three bf16 planes of kMlpBf16x3 take half again as much -- a model whose three-part groups do not fit the CU's LDS beside its tail
    // layers (the widest ones) runs the f32 matrix instructions instead: exact either way
    size_t wbuf = 0, lds = 0;
    for (;;) {
        wbuf = (size_t)(precision == kMlpBf16x3 ? 24 : 16) * NT * mlp_wpitch_bf16() * (NT <= 2 ? 2 : 1);
        const size_t h1 = (size_t)kMlpWaves * kMlpRowsPerWave * (16 * NT + 1);
        if (h1 > wbuf) wbuf = h1;
        wbuf = (wbuf + 3) & ~(size_t)3;
        lds = ((size_t)((m.tail_floats + 3) & ~3) + wbuf + (size_t)kMlpWaves * kMlpRowsPerWave * h2w) * sizeof(float);
        if (lds <= 160 * 1024) break;
        if (precision != kMlpBf16x3) return hipErrorInvalidValue;
        precision = kMlpF32;
    }
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpBf16>), 160 * 1024); e != hipSuccess) return e;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpF32>), 160 * 1024); e != hipSuccess) return e;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpF16x2>), 160 * 1024); e != hipSuccess) return e;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_mfma_kernel<NT, kMlpBf16x3>), 160 * 1024); e != hipSuccess) return e;
    if (B > 0xffffffffULL) return hipErrorInvalidValue;
    uint32_t *const no_redo = nullptr;
    if (precision == kMlpF16x2 || precision == kMlpRedoF32) {
        if (!redo) return hipErrorInvalidValue;
        if (precision == kMlpF16x2) {
            hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpF16x2>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                               m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1s), m.b1, m.tail, m.tail_floats, m.n_layers,
                               m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out, row_stride, rows_per_stream, stream_skip, mean, wsum, K, redo, 0);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return mlp_redo_abort(st, redo, e);
        }
        // the rows the split form listed (a feature beyond the f16 range), again with the f32 matrix instructions: sized for every row,
        // workgroups past the list's end leave at once
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpF32>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1h), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out, row_stride, rows_per_stream, stream_skip, mean, wsum, K, redo, 1);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return mlp_redo_abort(st, redo, e);
        return hipSuccess;
    } else if (precision == kMlpBf16x3)
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpBf16x3>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1t), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out, row_stride, rows_per_stream, stream_skip, mean, wsum, K, no_redo, 0);
    else if (precision == kMlpBf16)
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpBf16>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1h), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out, row_stride, rows_per_stream, stream_skip, mean, wsum, K, no_redo, 0);
    else
        hipLaunchKernelGGL((mlp_mfma_kernel<NT, kMlpF32>), dim3((unsigned)blocks), dim3(64 * kMlpWaves), lds, st, x, B, m.dims[0],
                           m.kpad, m.w1f, static_cast<const __bf16 *>(m.w1h), m.b1, m.tail, m.tail_floats, m.n_layers,
                           m.dims[1], m.dims[2], m.dims[3], m.dims[4], h2w, (int)wbuf, out, row_stride, rows_per_stream, stream_skip, mean, wsum, K, no_redo, 0);
    return hipGetLastError();
}

// LDS of mlp_mfma_kernel for this model (tail weights + staged layer-1 group(s) / h1 + the hidden tail layer): wide hidden
// layers do not fit the CU's 160 KB and take the per-layer kernel instead (Model::create)
bool mlp_mfma_fits(const MlpDev &m) {
    int h2w = 1;
    for (int l2 = 2; l2 < m.n_layers; ++l2) h2w = m.dims[l2] + 1 > h2w ? m.dims[l2] + 1 : h2w;
    h2w |= 1;
    size_t wbuf = (size_t)16 * m.nt * mlp_wpitch_bf16() * (m.nt <= 2 ? 2 : 1);
    const size_t h1 = (size_t)kMlpWaves * kMlpRowsPerWave * (16 * m.nt + 1);
    if (h1 > wbuf) wbuf = h1;
    wbuf = (wbuf + 3) & ~(size_t)3;
    const size_t lds = ((size_t)((m.tail_floats + 3) & ~3) + wbuf + (size_t)kMlpWaves * kMlpRowsPerWave * h2w) * sizeof(float);
    return lds <= 160 * 1024;
}

hipError_t launch_mlp_mfma(hipStream_t st, const MlpDev &m, const float *x, size_t B, int precision, float *out, uint32_t *redo) {
    if (B == 0) return hipSuccess;
    switch (m.nt) {
    case 1: return launch_mlp_nt<1>(st, m, x, B, precision, out, redo);
    case 2: return launch_mlp_nt<2>(st, m, x, B, precision, out, redo);
    case 5: return launch_mlp_nt<5>(st, m, x, B, precision, out, redo);
    case 9: return launch_mlp_nt<9>(st, m, x, B, precision, out, redo);
    }
    return hipErrorInvalidValue;
}

// per window the column means over its L frames, summed in frame order like MfccNormalizer::normalize
// (src/mfcc/normalizer.rs:3-31); one lane per (window, coefficient), 64 consecutive windows of a stream per block
__global__ __launch_bounds__(256) void window_means_kernel(const float *__restrict__ mfcc, size_t n_frames, size_t n_win, unsigned tiles,
                                                           int L, int K, float *__restrict__ mean) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *fs = reinterpret_cast<float *>(smem);  // [64 + L - 1][K]
    const size_t s = blockIdx.x / tiles;
    const size_t w0 = (size_t)(blockIdx.x - s * tiles) * 64;
    const size_t nw = n_win - w0 < 64 ? n_win - w0 : 64;
    const float *src = mfcc + (s * n_frames + w0) * K;
    const int nfr = (int)nw + L - 1;
    // (eight loads in flight per wait in both loops; the kernel is bound by the LDS reads of the sums -- 64 x K x L per block, each
    // window summed in frame order as MfccNormalizer::normalize does -- 0.22 ms for 4 096 streams x 202 windows either way)
#pragma unroll 8
    for (int i = threadIdx.x; i < nfr * K; i += 256) fs[i] = src[i];
    __syncthreads();
    // four coefficients per lane (K % 4 == 0, launch_window_means): 16-byte LDS reads move twice the bytes per LDS cycle of 4-byte ones
    // and a lane's four sums are independent chains (round 4: 1.81 -> 1.02 ms for 32 768 streams x 202 windows)
    const int K4 = K / 4;
    for (int i = threadIdx.x; i < (int)nw * K4; i += 256) {
        const int w = i / K4, q = i - w * K4;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        const float *col = fs + w * K + 4 * q;
#pragma unroll 8
        for (int f = 0; f < L; ++f) sum += *reinterpret_cast<const f32x4 *>(col + f * K);   // sequential sums, as MfccNormalizer::normalize
        const float n = (float)L;
        *reinterpret_cast<f32x4 *>(mean + (s * n_win + w0 + w) * K + 4 * q) = f32x4{sum.x / n, sum.y / n, sum.z / n, sum.w / n};
    }
}

hipError_t launch_window_means(hipStream_t st, const float *mfcc, size_t S, size_t n_frames, size_t n_win, int L, int K, float *mean) {
    if (S == 0 || n_win == 0) return hipSuccess;
    const size_t tiles = (n_win + 63) / 64, blocks = tiles * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = (size_t)(64 + L - 1) * K * sizeof(float);
    if (lds > 64 * 1024 || K % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(window_means_kernel, dim3((unsigned)blocks), dim3(256), lds, st, mfcc, n_frames, n_win, (unsigned)tiles, L, K, mean);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Layer 1 over ALL windows of a stream from one staging of its frames (round 4).  mlp_mfma_kernel in window mode reads every window's
// 3 120 features through the vector cache and splits each of them into its two f16 parts again -- a frame belongs to L = 195 windows,
// so the same split is done 195 times: 57 % of the SIMD cycles of that kernel are vector instructions, 24 % matrix ones (PMC, 32 768
// streams: 7.6 ms), and every 64 rows stage the 400 KB of split weights again.  Here a workgroup owns NT <= 7 tiles of 32 consecutive
// windows of ONE stream: their frames (32 NT + L - 1) are split once into LDS, part p / k-half h / frame in 16-byte slots (a wave's 32
// rows read 32 consecutive slots: conflict-free), and window row w at k-step f (= frame f of the window: mfcc_size 16 is one
// v_mfma_f32_32x32x16_f16 step) is simply slot w + f.  The four waves split the FRAMES of the window, not the rows: wave q runs
// frames [q L/4, (q+1) L/4) for all NT tiles, so a weight fragment is fetched once per workgroup -- straight from the lane-ordered image
// (MlpDev::wwin) into registers, four frames ahead -- and used NT times; there is no barrier inside the loop (the first version staged
// weight groups through LDS for 2 tiles per wave: 39 barriers per workgroup, 52 % matrix-pipe occupancy, 4.9 ms).  The four partial
// sums of a tile meet in LDS in a fixed order ((q0 + q1) + (q2 + q3)).  Same products as kMlpF16x2 (x0 w0 + x1 w0 + x0 w1), same mean
// correction, bias, ReLU and tail layers as mlp_mfma_kernel; rows holding a frame beyond the f16 range are listed for the f32 pass.
// Shapes: mfcc_size 16, layer 1 <= 32 wide, tail layers <= 32 wide, n_win >= 32.
// BITS: the frames are staged minus the mean of the workgroup's middle window (below), so a window's logits depend on how its stream is cut
// into workgroups -- which is a function of the stream's window count alone: the same stream gives the same bits alone, in any batch and
// on every repetition (tests/test_gpu_model_pin.py), but NOT the bits of mlp_mfma_kernel, which live-stream calls with fewer than 32 new
// windows per stream take (the row's own mean): those agree within the logit gate, 2e-6 on scores.
constexpr int kWinWaves = 4, kWinTile = 32, kWinMaxTiles = 7, kWinAhead = 3;
typedef float f32x16w __attribute__((ext_vector_type(16)));
// P3 (round 6, RP_MLP_F32): the frames are staged as THREE bf16 parts (exact) and a frame step is six v_mfma_f32_32x32x16_bf16 per tile
// (x_i w_j, i + j <= 2, smallest first) against the three-part weight image MlpDev::wwin3 -- f32-grade products, no f16 range, nothing listed;
// three planes of frames and seven accumulator tiles take two waves per SIMD.  !P3: two f16 parts (RP_MLP_F32_FAST), as described above.
template <int NT, bool P3>
__global__ __launch_bounds__(64 * kWinWaves, P3 ? 2 : 3) void mlp_windows_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_win, int L, unsigned blocks_per_stream, const u32x4v *__restrict__ wimg, int slots,
    int n1p, const float *__restrict__ b1, const float *__restrict__ mean, const float *__restrict__ wsum, const float *__restrict__ tail,
    int tail_floats, int n_layers, int d1, int d2, int d3, float *__restrict__ out, uint32_t *redo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tl = reinterpret_cast<float *>(smem);                                   // tail weights
    unsigned *flag = reinterpret_cast<unsigned *>(tl + ((tail_floats + 3) & ~3));   // [slots] frame holds a value beyond the f16 range
    u32x4v *A = reinterpret_cast<u32x4v *>(flag + slots);                           // [2 parts][2 k-halves][slots]; later the partial sums
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, lr = l & 31, lh = l >> 5;
    const size_t s = blockIdx.x / blocks_per_stream;
    const size_t w0 = (size_t)(blockIdx.x - s * blocks_per_stream) * (NT * kWinTile);   // first window of this workgroup
    const size_t rows_here = n_win - w0 < (size_t)(NT * kWinTile) ? n_win - w0 : (size_t)(NT * kWinTile);
    const int n_real = (int)rows_here + L - 1;                                          // frames w0 .. w0 + n_real - 1 exist
    for (int i = tid; i < tail_floats; i += 64 * kWinWaves) tl[i] = tail[i];
    // ---- frames -> the two f16 parts, once.  The window mean is taken out after layer 1 (W.(f - mu) = W.f - sum_k mu[k] wsum[k]); MFCC
    // coefficients sit on offsets many times their spread, and summing W.f with the offsets inside costs the sum their size in f32
    // rounding.  So the frames are staged minus c = the mean of the workgroup's middle window -- every window here overlaps or
    // adjoins it, so mu - c is small -- and the correction uses mu - c: W.(f - mu) = W.(f - c) - sum_k (mu - c)[k] wsum[k].
    const float *src = mfcc + (s * frame_pitch + w0) * 16;
    const float4 *cmid = reinterpret_cast<const float4 *>(mean + (s * n_win + w0 + rows_here / 2) * 16);
    {
        // a thread's items are (frame i / 2, k-half i & 1 = tid & 1) for i = tid, tid + 256, ..: every load goes out before the first split
        constexpr int MAXI = (2 * (kWinMaxTiles * kWinTile + 256) + 64 * kWinWaves - 1) / (64 * kWinWaves);
        const int h = tid & 1;
        const float4 cl = cmid[2 * h], ch = cmid[2 * h + 1];
        float4 lo[MAXI], hi[MAXI];
#pragma unroll
        for (int v = 0; v < MAXI; ++v) {
            const int i = tid + v * 64 * kWinWaves, fr = i >> 1;
            if (i < 2 * slots && fr < n_real) {
                lo[v] = *reinterpret_cast<const float4 *>(src + (size_t)fr * 16 + 8 * h);
                hi[v] = *reinterpret_cast<const float4 *>(src + (size_t)fr * 16 + 8 * h + 4);
            } else { lo[v] = cl; hi[v] = ch; }   // past the real frames: zero after the offset leaves
        }
#pragma unroll
        for (int v = 0; v < MAXI; ++v) {
            const int i = tid + v * 64 * kWinWaves, fr = i >> 1;
            if (i < 2 * slots) {
                const float xs[8] = {lo[v].x - cl.x, lo[v].y - cl.y, lo[v].z - cl.z, lo[v].w - cl.w, hi[v].x - ch.x, hi[v].y - ch.y, hi[v].z - ch.z, hi[v].w - ch.w};
                u32x4v p0, p1, p2;
                float rng = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = xs[2 * e], b = xs[2 * e + 1];
                    if (P3) {   // a = a0 + a1 + a2 exactly: a0 = a & 0xffff0000 (a bf16), r = a - a0, a1 = r & 0xffff0000, a2 = r - a1; a packed register = two upper halves
                        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
                        p0[e] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
                        p1[e] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
                        p2[e] = __builtin_amdgcn_perm(__float_as_uint(rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u)),
                                                      __float_as_uint(ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u)), 0x07060302u);
                    } else {
                        rng = fmaxf(fmaxf(rng, fabsf(a)), fabsf(b));
                        p0[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
                        p1[e] = pk_f16_second(a - __uint_as_float(__float_as_uint(a) & 0xffffe000u), b - __uint_as_float(__float_as_uint(b) & 0xffffe000u));
                    }
                }
                A[(0 * 2 + h) * slots + fr] = p0;
                A[(1 * 2 + h) * slots + fr] = p1;
                if (P3) A[(2 * 2 + h) * slots + fr] = p2;
                if (h == 0) flag[fr] = 0u;   // the two halves of a frame belong to neighbouring lanes of one wave: its LDS stores keep their order
                if (!(rng <= 65504.f)) flag[fr] = 1u;
            }
        }
    }
    __syncthreads();
    // ---- this wave's quarter of the frames, all NT tiles
    const int Lq = (L + kWinWaves - 1) / kWinWaves, fb = wave * Lq, fe = fb + Lq < L ? fb + Lq : L;
    f32x16w acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    constexpr int NPART = P3 ? 3 : 2;
    u32x4v wq0[kWinAhead], wq1[kWinAhead], wq2[P3 ? kWinAhead : 1];   // the weight fragments of the next kWinAhead frames: [frame][part][k-half][output] 16 bytes
    auto wfetch = [&](int f, int j) __attribute__((always_inline)) {
        const int fc = f < fe ? f : fe - 1;
        wq0[j] = wimg[(((size_t)fc * NPART + 0) * 2 + lh) * 32 + lr];
        wq1[j] = wimg[(((size_t)fc * NPART + 1) * 2 + lh) * 32 + lr];
        if (P3) wq2[j] = wimg[(((size_t)fc * NPART + 2) * 2 + lh) * 32 + lr];
    };
    if (fb < fe) {
#pragma unroll
        for (int j = 0; j < kWinAhead; ++j) wfetch(fb + j, j);
        const u32x4v *A0 = A + (0 * 2 + lh) * slots + lr, *A1 = A + (1 * 2 + lh) * slots + lr;
        const u32x4v *A2 = A + ((P3 ? 2 : 0) * 2 + lh) * slots + lr;
        (void)A2;
        for (int f0 = fb; f0 < fe; f0 += kWinAhead) {
#pragma unroll
            for (int j = 0; j < kWinAhead; ++j) {
                const int f = f0 + j;
                if (f < fe) {   // wave-uniform
                    if (P3) {   // six products per tile, smallest first; tiles in pairs so that consecutive matrix instructions do not wait on each other
                        const bf16x8 c0 = __builtin_bit_cast(bf16x8, wq0[j]), c1 = __builtin_bit_cast(bf16x8, wq1[j]), c2 = __builtin_bit_cast(bf16x8, wq2[j]);
                        wfetch(f + kWinAhead, j);
#pragma unroll
                        for (int t = 0; t < NT; t += 2) {
                            const bool two = t + 1 < NT;
                            const bf16x8 x0 = __builtin_bit_cast(bf16x8, A0[t * kWinTile + f]), x1 = __builtin_bit_cast(bf16x8, A1[t * kWinTile + f]),
                                         x2 = __builtin_bit_cast(bf16x8, A2[t * kWinTile + f]);
                            bf16x8 y0 = x0, y1 = x1, y2 = x2;
                            if (two) {
                                y0 = __builtin_bit_cast(bf16x8, A0[(t + 1) * kWinTile + f]); y1 = __builtin_bit_cast(bf16x8, A1[(t + 1) * kWinTile + f]);
                                y2 = __builtin_bit_cast(bf16x8, A2[(t + 1) * kWinTile + f]);
                            }
#define RP_WIN6(XA, XB, WB)                                                                                            \
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XA, WB, acc[t], 0, 0, 0);                 \
                            if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XB, WB, acc[t + 1], 0, 0, 0);
                            RP_WIN6(x0, y0, c2) RP_WIN6(x1, y1, c1) RP_WIN6(x2, y2, c0) RP_WIN6(x0, y0, c1) RP_WIN6(x1, y1, c0) RP_WIN6(x0, y0, c0)
#undef RP_WIN6
                        }
                        continue;
                    }
                    const f16x8 b0 = __builtin_bit_cast(f16x8, wq0[j]), b1v = __builtin_bit_cast(f16x8, wq1[j]);
                    wfetch(f + kWinAhead, j);
                    // tiles in pairs: the two parts of two tiles live at a time (all NT at once cost 8 NT registers and the third wave
                    // per SIMD), and the three products of a tile are one other matrix instruction apart
#pragma unroll
                    for (int t = 0; t < NT; t += 2) {
                        const bool two = t + 1 < NT;
                        const f16x8 p0 = __builtin_bit_cast(f16x8, A0[t * kWinTile + f]), p1 = __builtin_bit_cast(f16x8, A1[t * kWinTile + f]);
                        f16x8 q0 = p0, q1 = p1;
                        if (two) { q0 = __builtin_bit_cast(f16x8, A0[(t + 1) * kWinTile + f]); q1 = __builtin_bit_cast(f16x8, A1[(t + 1) * kWinTile + f]); }
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b0, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, b0, acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b0, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, b0, acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b1v, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, b1v, acc[t + 1], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the frame planes: the region becomes the partial sums
    // ---- ((q3 + q2) + q1) + q0 through one set of sums R[tile][e / 4][lane] (16 bytes each): a wave adds what is there to its own
    // and puts the result back; one set, and h1 of a tile in that tile's own 4 KB, keep the workgroup at 48 KB = three per CU
    f32x4 *R = reinterpret_cast<f32x4 *>(A);
    auto put = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) R[(t * 4 + e4) * 64 + l] = f32x4{acc[t][4 * e4], acc[t][4 * e4 + 1], acc[t][4 * e4 + 2], acc[t][4 * e4 + 3]};
    };
    auto add = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 v = R[(t * 4 + e4) * 64 + l];
                acc[t][4 * e4] = v.x + acc[t][4 * e4]; acc[t][4 * e4 + 1] = v.y + acc[t][4 * e4 + 1];
                acc[t][4 * e4 + 2] = v.z + acc[t][4 * e4 + 2]; acc[t][4 * e4 + 3] = v.w + acc[t][4 * e4 + 3];
            }
    };
    if (wave == 3) put();
    __syncthreads();
    if (wave == 2) { add(); put(); }
    __syncthreads();
    if (wave == 1) { add(); put(); }
    __syncthreads();
    if (wave == 0) { add(); put(); }
    __syncthreads();
    // ---- bias, mean correction, ReLU, tail layers: tile t by wave t % 4.  h1 [32][32] takes the place of the tile's sums, h2 [32][32] of
    // the wave sits behind the sums; column i of row r lives at (i + r) % 32: a lane walking its own row and the 32 lanes writing one
    // row both touch 32 different banks
    float *h2 = reinterpret_cast<float *>(R + NT * 4 * 64) + wave * 32 * 32;
    const bool relu1 = n_layers > 1;
    float4 ws4[4], c4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ws4[k] = lr < n1p ? *reinterpret_cast<const float4 *>(wsum + (size_t)lr * 16 + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
        c4[k] = cmid[k];
    }
    const float bias1 = lr < n1p ? b1[lr] : 0.f;
    for (int t = wave; t < NT; t += kWinWaves) {
        const size_t wrow0 = (size_t)t * kWinTile;   // first row of the tile inside the workgroup
        if (wrow0 >= rows_here) break;                // wave-uniform
        float *h1 = reinterpret_cast<float *>(R + t * 4 * 64);
        f32x4 sums[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) sums[e4] = R[(t * 4 + e4) * 64 + l];
        wave_lds_sync();
        // C/D layout of the 32x32 tile: lane (column lr, half lh) holds rows 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const int row = 8 * e4 + 4 * lh + ee;
                size_t rr = wrow0 + row;
                if (rr >= rows_here) rr = rows_here - 1;
                const float4 *mu = reinterpret_cast<const float4 *>(mean + (s * n_win + w0 + rr) * 16);
                float corr = 0.f;   // - sum_k (mu[row][k] - c[k]) * wsum[o][k]
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 m4 = mu[k], w4 = ws4[k], c = c4[k];
                    corr = fmaf(m4.x - c.x, w4.x, corr); corr = fmaf(m4.y - c.y, w4.y, corr); corr = fmaf(m4.z - c.z, w4.z, corr); corr = fmaf(m4.w - c.w, w4.w, corr);
                }
                float v = sums[e4][ee] - corr;
                v += bias1;
                if (relu1 && v < 0.f) v = 0.f;
                h1[row * 32 + ((lr + row) & 31)] = v;
            }
        }
        wave_lds_sync();
        // tail layers: lane = (row lr, output phase lh), outputs strided by 2 over the phases; sums in mlp_mfma_kernel's order
        {
            const bool row_ok = wrow0 + lr < rows_here;
            const size_t orow = s * n_win + w0 + wrow0 + lr;
            if (!P3 && lh == 0 && row_ok) {   // a window holding a frame beyond the f16 range: listed for the f32 pass
                unsigned far = 0u;
                for (int f = 0; f < L; ++f) far |= flag[wrow0 + lr + f];
                if (far) mlp_redo_append(redo, (uint32_t)orow, (size_t)(gridDim.x / blocks_per_stream) * n_win);
            }
            const int dd[4] = {d1, d2, d3, 0};
            const float *hin = h1 + lr * 32;
            const float *wp = tl;
            int cur_in = d1;
            float *dst = out + orow * (size_t)dd[n_layers - 1];
            if (n_layers == 1 && row_ok)
                for (int o = lh; o < d1; o += 2) dst[o] = hin[(o + lr) & 31];
            for (int layer = 1; layer < n_layers; ++layer) {
                const int on = dd[layer];
                const bool last = layer + 1 == n_layers;
                for (int o = lh; o < on; o += 2) {
                    const float *wr = wp + (size_t)o * cur_in;
                    float s0 = 0.f, s1 = 0.f;
                    int i = 0;
                    for (; i + 1 < cur_in; i += 2) { s0 = fmaf(hin[(i + lr) & 31], wr[i], s0); s1 = fmaf(hin[(i + 1 + lr) & 31], wr[i + 1], s1); }
                    if (i < cur_in) s0 = fmaf(hin[(i + lr) & 31], wr[i], s0);
                    float sacc = (s0 + s1) + wp[(size_t)on * cur_in + o];
                    if (!last && sacc < 0.f) sacc = 0.f;
                    if (last) { if (row_ok) dst[o] = sacc; } else h2[lr * 32 + ((o + lr) & 31)] = sacc;
                }
                wave_lds_sync();
                wp += (size_t)on * cur_in + on;
                cur_in = on;
                hin = h2 + lr * 32;
            }
        }
        wave_lds_sync();
    }
}

// The same for layer 1 wider than 32 (the Medium and Large model types: 65 / 130 outputs for 195 frames): NQ = 2 .. 5 waves, wave q owns
// the 32-output tile q for all NT row tiles and ALL frames of the window (its own slice of the weight image, four frames ahead, used NT
// times) -- no partial sums to add up; the tiles then meet in LDS row tile by row tile ([32][32 NQ + 1] floats) for the tail layers, which
// every lane of the workgroup shares (row, output phase).  mlp_mfma_kernel in window mode, which these shapes had until now: 5.6 ms
// (Medium) / 13.5 ms (Large) per 8 192 streams.
template <int NT, int NQ, bool P3>
__global__ __launch_bounds__(64 * NQ, 2) void mlp_windows_wide_kernel(
    const float *__restrict__ mfcc, size_t frame_pitch, size_t n_win, int L, unsigned blocks_per_stream, const u32x4v *__restrict__ wimg, int slots,
    int n1p, const float *__restrict__ b1, const float *__restrict__ mean, const float *__restrict__ wsum, const float *__restrict__ tail,
    int tail_floats, int n_layers, int d1, int d2, int d3, float *__restrict__ out, uint32_t *redo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTHR = 64 * NQ, P1 = 32 * NQ + 1;
    float *tl = reinterpret_cast<float *>(smem);                                   // tail weights
    unsigned *flag = reinterpret_cast<unsigned *>(tl + ((tail_floats + 3) & ~3));   // [slots] frame holds a value beyond the f16 range
    u32x4v *A = reinterpret_cast<u32x4v *>(flag + slots);                           // [2 parts][2 k-halves][slots]; later h1 / h2 of a row tile
    const int tid = threadIdx.x, q = tid >> 6, l = tid & 63, lr = l & 31, lh = l >> 5;
    const size_t s = blockIdx.x / blocks_per_stream;
    const size_t w0 = (size_t)(blockIdx.x - s * blocks_per_stream) * (NT * kWinTile);
    const size_t rows_here = n_win - w0 < (size_t)(NT * kWinTile) ? n_win - w0 : (size_t)(NT * kWinTile);
    const int n_real = (int)rows_here + L - 1;
    for (int i = tid; i < tail_floats; i += NTHR) tl[i] = tail[i];
    // ---- frames minus the middle window's mean -> the two f16 parts, once (as mlp_windows_kernel)
    const float *src = mfcc + (s * frame_pitch + w0) * 16;
    const float4 *cmid = reinterpret_cast<const float4 *>(mean + (s * n_win + w0 + rows_here / 2) * 16);
    {
        constexpr int MAXI = (2 * (kWinMaxTiles * kWinTile + 256) + NTHR - 1) / NTHR;
        const int h = tid & 1;
        const float4 cl = cmid[2 * h], ch = cmid[2 * h + 1];
        float4 lo[MAXI], hi[MAXI];
#pragma unroll
        for (int v = 0; v < MAXI; ++v) {
            const int i = tid + v * NTHR, fr = i >> 1;
            if (i < 2 * slots && fr < n_real) {
                lo[v] = *reinterpret_cast<const float4 *>(src + (size_t)fr * 16 + 8 * h);
                hi[v] = *reinterpret_cast<const float4 *>(src + (size_t)fr * 16 + 8 * h + 4);
            } else { lo[v] = cl; hi[v] = ch; }
        }
#pragma unroll
        for (int v = 0; v < MAXI; ++v) {
            const int i = tid + v * NTHR, fr = i >> 1;
            if (i < 2 * slots) {
                const float xs[8] = {lo[v].x - cl.x, lo[v].y - cl.y, lo[v].z - cl.z, lo[v].w - cl.w, hi[v].x - ch.x, hi[v].y - ch.y, hi[v].z - ch.z, hi[v].w - ch.w};
                u32x4v p0, p1, p2;
                float rng = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = xs[2 * e], b = xs[2 * e + 1];
                    if (P3) {   // a = a0 + a1 + a2 exactly: a0 = a & 0xffff0000 (a bf16), r = a - a0, a1 = r & 0xffff0000, a2 = r - a1; a packed register = two upper halves
                        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
                        p0[e] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
                        p1[e] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
                        p2[e] = __builtin_amdgcn_perm(__float_as_uint(rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u)),
                                                      __float_as_uint(ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u)), 0x07060302u);
                    } else {
                        rng = fmaxf(fmaxf(rng, fabsf(a)), fabsf(b));
                        p0[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
                        p1[e] = pk_f16_second(a - __uint_as_float(__float_as_uint(a) & 0xffffe000u), b - __uint_as_float(__float_as_uint(b) & 0xffffe000u));
                    }
                }
                A[(0 * 2 + h) * slots + fr] = p0;
                A[(1 * 2 + h) * slots + fr] = p1;
                if (P3) A[(2 * 2 + h) * slots + fr] = p2;
                if (h == 0) flag[fr] = 0u;
                if (!(rng <= 65504.f)) flag[fr] = 1u;
            }
        }
    }
    __syncthreads();
    // ---- every frame, this wave's 32 outputs, all NT tiles
    f32x16w acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    constexpr int NPART = P3 ? 3 : 2;
    u32x4v wq0[kWinAhead], wq1[kWinAhead], wq2[P3 ? kWinAhead : 1];
    auto wfetch = [&](int f, int j) __attribute__((always_inline)) {
        const int fc = f < L ? f : L - 1;
        wq0[j] = wimg[((((size_t)fc * NQ + q) * NPART + 0) * 2 + lh) * 32 + lr];
        wq1[j] = wimg[((((size_t)fc * NQ + q) * NPART + 1) * 2 + lh) * 32 + lr];
        if (P3) wq2[j] = wimg[((((size_t)fc * NQ + q) * NPART + 2) * 2 + lh) * 32 + lr];
    };
    {
#pragma unroll
        for (int j = 0; j < kWinAhead; ++j) wfetch(j, j);
        const u32x4v *A0 = A + (0 * 2 + lh) * slots + lr, *A1 = A + (1 * 2 + lh) * slots + lr;
        const u32x4v *A2 = A + ((P3 ? 2 : 0) * 2 + lh) * slots + lr;
        (void)A2;
        for (int f0 = 0; f0 < L; f0 += kWinAhead) {
#pragma unroll
            for (int j = 0; j < kWinAhead; ++j) {
                const int f = f0 + j;
                if (f < L) {   // wave-uniform
                    if (P3) {
                        const bf16x8 c0 = __builtin_bit_cast(bf16x8, wq0[j]), c1 = __builtin_bit_cast(bf16x8, wq1[j]), c2 = __builtin_bit_cast(bf16x8, wq2[j]);
                        wfetch(f + kWinAhead, j);
#pragma unroll
                        for (int t = 0; t < NT; t += 2) {
                            const bool two = t + 1 < NT;
                            const bf16x8 x0 = __builtin_bit_cast(bf16x8, A0[t * kWinTile + f]), x1 = __builtin_bit_cast(bf16x8, A1[t * kWinTile + f]),
                                         x2 = __builtin_bit_cast(bf16x8, A2[t * kWinTile + f]);
                            bf16x8 y0 = x0, y1 = x1, y2 = x2;
                            if (two) {
                                y0 = __builtin_bit_cast(bf16x8, A0[(t + 1) * kWinTile + f]); y1 = __builtin_bit_cast(bf16x8, A1[(t + 1) * kWinTile + f]);
                                y2 = __builtin_bit_cast(bf16x8, A2[(t + 1) * kWinTile + f]);
                            }
#define RP_WIN6(XA, XB, WB)                                                                                            \
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XA, WB, acc[t], 0, 0, 0);                 \
                            if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(XB, WB, acc[t + 1], 0, 0, 0);
                            RP_WIN6(x0, y0, c2) RP_WIN6(x1, y1, c1) RP_WIN6(x2, y2, c0) RP_WIN6(x0, y0, c1) RP_WIN6(x1, y1, c0) RP_WIN6(x0, y0, c0)
#undef RP_WIN6
                        }
                        continue;
                    }
                    const f16x8 b0 = __builtin_bit_cast(f16x8, wq0[j]), b1v = __builtin_bit_cast(f16x8, wq1[j]);
                    wfetch(f + kWinAhead, j);
#pragma unroll
                    for (int t = 0; t < NT; t += 2) {
                        const bool two = t + 1 < NT;
                        const f16x8 p0 = __builtin_bit_cast(f16x8, A0[t * kWinTile + f]), p1 = __builtin_bit_cast(f16x8, A1[t * kWinTile + f]);
                        f16x8 q0 = p0, q1 = p1;
                        if (two) { q0 = __builtin_bit_cast(f16x8, A0[(t + 1) * kWinTile + f]); q1 = __builtin_bit_cast(f16x8, A1[(t + 1) * kWinTile + f]); }
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b0, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, b0, acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b0, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, b0, acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, b1v, acc[t], 0, 0, 0);
                        if (two) acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, b1v, acc[t + 1], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the frame planes: the region becomes h1 [32][P1] and h2 [32][33] of one row tile at a time
    float *h1 = reinterpret_cast<float *>(A), *h2 = h1 + 32 * P1;
    const bool relu1 = n_layers > 1;
    const int col = 32 * q + lr;
    float4 ws4[4], c4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ws4[k] = col < n1p ? *reinterpret_cast<const float4 *>(wsum + (size_t)col * 16 + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
        c4[k] = cmid[k];
    }
    const float bias1 = col < n1p ? b1[col] : 0.f;
    const int prow = tid & 31, ph = tid >> 5;   // tail layers: lane = (row, output phase), 2 NQ phases
    constexpr int NPH = 2 * NQ;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const size_t wrow0 = (size_t)t * kWinTile;
        if (wrow0 >= rows_here) break;   // workgroup-uniform
        // C/D layout of the 32x32 tile: lane (column lr, half lh) holds rows 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * lh + (e & 3);
            size_t rr = wrow0 + row;
            if (rr >= rows_here) rr = rows_here - 1;
            const float4 *mu = reinterpret_cast<const float4 *>(mean + (s * n_win + w0 + rr) * 16);
            float corr = 0.f;   // - sum_k (mu[row][k] - c[k]) * wsum[o][k]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 m4 = mu[k], w4 = ws4[k], c = c4[k];
                corr = fmaf(m4.x - c.x, w4.x, corr); corr = fmaf(m4.y - c.y, w4.y, corr); corr = fmaf(m4.z - c.z, w4.z, corr); corr = fmaf(m4.w - c.w, w4.w, corr);
            }
            float v = acc[t][e] - corr;
            v += bias1;
            if (relu1 && v < 0.f) v = 0.f;
            h1[row * P1 + col] = v;
        }
        __syncthreads();
        const bool row_ok = wrow0 + prow < rows_here;
        const size_t orow = s * n_win + w0 + wrow0 + prow;
        if (!P3 && ph == 0 && row_ok) {   // a window holding a frame beyond the f16 range: listed for the f32 pass
            unsigned far = 0u;
            for (int f = 0; f < L; ++f) far |= flag[wrow0 + prow + f];
            if (far) mlp_redo_append(redo, (uint32_t)orow, (size_t)(gridDim.x / blocks_per_stream) * n_win);
        }
        const int dd[4] = {d1, d2, d3, 0};
        float *dst = out + orow * (size_t)dd[n_layers - 1];
        const float *hin = h1 + prow * P1;
        if (n_layers == 1) {
            if (row_ok) for (int o = ph; o < d1; o += NPH) dst[o] = hin[o];
        } else {
            const float *wp = tl;
            int cur_in = d1;
            for (int layer = 1; layer < n_layers; ++layer) {
                const int on = dd[layer];
                const bool last = layer + 1 == n_layers;
                for (int o = ph; o < on; o += NPH) {   // sums in mlp_mfma_kernel's order
                    const float *wr = wp + (size_t)o * cur_in;
                    float s0 = 0.f, s1 = 0.f;
                    int i = 0;
                    for (; i + 1 < cur_in; i += 2) { s0 = fmaf(hin[i], wr[i], s0); s1 = fmaf(hin[i + 1], wr[i + 1], s1); }
                    if (i < cur_in) s0 = fmaf(hin[i], wr[i], s0);
                    float sacc = (s0 + s1) + wp[(size_t)on * cur_in + o];
                    if (!last && sacc < 0.f) sacc = 0.f;
                    if (last) { if (row_ok) dst[o] = sacc; } else h2[prow * 33 + o] = sacc;
                }
                __syncthreads();
                wp += (size_t)on * cur_in + on;
                cur_in = on;
                hin = h2 + prow * 33;
            }
        }
        __syncthreads();   // h1 / h2 are free for the next row tile
    }
}

int mlp_windows_supported(const MlpDev &m, size_t n_win, int K, bool three_part) {
    if (!(three_part ? m.wwin3 : m.wwin) || K != 16 || m.dims[0] % 16 != 0 || m.dims[1] > 160 || n_win < 32) return 0;
    const int L = m.dims[0] / 16;
    if (L > 256 || m.tail_floats > 6144) return 0;
    for (int l2 = 2; l2 <= m.n_layers; ++l2) if (m.dims[l2] > 32) return 0;
    const char *env = std::getenv("RP_MLP_WINDOWS");
    if (env && env[0] == '0') return 0;
    return m.dims[1] <= 32 ? 1 : 2;
}

template <int NT, int NQ, bool P3>
static hipError_t launch_mlp_windows_wide_nq(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_win, const float *mean,
                                             const float *wsum, float *out, uint32_t *redo, size_t pitch) {
    const int L = m.dims[0] / 16;
    const size_t bps = (n_win + NT * kWinTile - 1) / (NT * kWinTile), blocks = bps * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const int slots = (NT * kWinTile + L + 3) & ~3;
    const size_t region = std::max((size_t)(P3 ? 6 : 4) * slots * 16, (size_t)32 * (32 * NQ + 1) * 4 + (size_t)32 * 33 * 4);
    const size_t lds = (size_t)((m.tail_floats + 3) & ~3) * 4 + (size_t)slots * 4 + region;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_windows_wide_kernel<NT, NQ, P3>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((mlp_windows_wide_kernel<NT, NQ, P3>), dim3((unsigned)blocks), dim3(64 * NQ), lds, st, mfcc, pitch, n_win, L, (unsigned)bps,
                       static_cast<const u32x4v *>(P3 ? m.wwin3 : m.wwin), slots, 16 * m.nt, m.b1, mean, wsum, m.tail, m.tail_floats, m.n_layers, m.dims[1], m.dims[2],
                       m.dims[3], out, redo);
    return hipGetLastError();
}

template <int NT, bool P3>
static hipError_t launch_mlp_windows_wide_nt(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_win, const float *mean,
                                             const float *wsum, float *out, uint32_t *redo, size_t pitch) {
    switch ((m.dims[1] + 31) / 32) {
    case 2: return launch_mlp_windows_wide_nq<NT, 2, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 3: return launch_mlp_windows_wide_nq<NT, 3, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 4: return launch_mlp_windows_wide_nq<NT, 4, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 5: return launch_mlp_windows_wide_nq<NT, 5, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    }
    return hipErrorInvalidValue;
}

template <bool P3>
static hipError_t launch_mlp_windows_wide(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_win, const float *mean,
                                          const float *wsum, float *out, uint32_t *redo, size_t pitch) {
    // row tiles per workgroup: 2 or 4 (measured per 8 192 streams x 202 windows, Medium / Large: 7 tiles 3.87 / 8.62 ms at two waves per SIMD,
    // 4 tiles 3.36 / 7.90 at four, 2 tiles 3.84 / 11.2 -- the weights once per 64 rows)
    const size_t tiles = (n_win + kWinTile - 1) / kWinTile, wgs = (tiles + 3) / 4, per = (tiles + wgs - 1) / wgs;
    if (per <= 2) return launch_mlp_windows_wide_nt<2, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    return launch_mlp_windows_wide_nt<4, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
}

template <int NT, bool P3>
static hipError_t launch_mlp_windows_nt(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_win, const float *mean,
                                        const float *wsum, float *out, uint32_t *redo, size_t pitch) {
    const int L = m.dims[0] / 16;
    const size_t bps = (n_win + NT * kWinTile - 1) / (NT * kWinTile), blocks = bps * S;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const int slots = (NT * kWinTile + L + 3) & ~3;
    // frame planes (two or three parts x two k-halves); later the sums of the tiles (h1 of a tile in its place) + h2 of the four waves
    const size_t region = std::max((size_t)(P3 ? 6 : 4) * slots * 16, (size_t)NT * 4096 + (size_t)kWinWaves * 4096);
    const size_t lds = (size_t)((m.tail_floats + 3) & ~3) * 4 + (size_t)slots * 4 + region;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(mlp_windows_kernel<NT, P3>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((mlp_windows_kernel<NT, P3>), dim3((unsigned)blocks), dim3(64 * kWinWaves), lds, st, mfcc, pitch, n_win, L, (unsigned)bps,
                       static_cast<const u32x4v *>(P3 ? m.wwin3 : m.wwin), slots, 16 * m.nt, m.b1, mean, wsum, m.tail, m.tail_floats, m.n_layers, m.dims[1], m.dims[2],
                       m.dims[3], out, redo);
    return hipGetLastError();
}

template <bool P3>
static hipError_t launch_mlp_windows(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_frames, size_t n_win, const float *mean,
                                     const float *wsum, float *out, uint32_t *redo, size_t pitch) {
    (void)n_frames;
    // tiles per workgroup: as few workgroups per stream as 7 tiles each allow, the tiles spread evenly over them
    static const int cap_env = std::getenv("RP_MLP_WIN_TILES") ? std::atoi(std::getenv("RP_MLP_WIN_TILES")) : kWinMaxTiles;   // experiments
    const size_t cap = cap_env >= 1 && cap_env <= kWinMaxTiles ? (size_t)cap_env : (size_t)kWinMaxTiles;
    const size_t tiles = (n_win + kWinTile - 1) / kWinTile, wgs = (tiles + cap - 1) / cap, per = (tiles + wgs - 1) / wgs;
    switch (per) {
    case 1: return launch_mlp_windows_nt<1, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 2: return launch_mlp_windows_nt<2, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 3: return launch_mlp_windows_nt<3, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 4: return launch_mlp_windows_nt<4, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 5: return launch_mlp_windows_nt<5, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 6: return launch_mlp_windows_nt<6, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    case 7: return launch_mlp_windows_nt<7, P3>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mlp_mfma_windows(hipStream_t st, const MlpDev &m, const float *mfcc, size_t S, size_t n_frames, size_t n_win, int K,
                                   const float *mean, const float *wsum, float *out, uint32_t *redo, size_t frame_pitch, int precision) {
    const size_t B = S * n_win;
    if (B == 0) return hipSuccess;
    if (K < 1 || K % 4 != 0 || m.dims[0] % K != 0) return hipErrorInvalidValue;  // 16-byte aligned window rows
    const size_t L = (size_t)m.dims[0] / K;
    // window w of stream s starts at frame s * pitch + w: rows advance by K floats, a new stream skips the rest of its row
    const size_t pitch = frame_pitch ? frame_pitch : n_win + L - 1;
    if (pitch < n_win) return hipErrorInvalidValue;
    (void)n_frames;
    const size_t skip = (pitch - n_win) * K;
    // the staged-frame kernels (mlp_windows_kernel / mlp_windows_wide_kernel): three bf16 parts for RP_MLP_F32 (exact operands, nothing to
    // list), two f16 parts for RP_MLP_F32_FAST (+ the listed rows again with the f32 instructions)
    if (precision == kMlpF32 && redo) {
        if (const int form = mlp_windows_supported(m, n_win, K, true)) {
            if (S * n_win > 0xffffffffULL) return hipErrorInvalidValue;
            return form == 1 ? launch_mlp_windows<true>(st, m, mfcc, S, n_frames, n_win, mean, wsum, out, redo, pitch)
                             : launch_mlp_windows_wide<true>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch);
        }
    }
    if (const int form = (precision == kMlpF16x2 && redo) ? mlp_windows_supported(m, n_win, K, false) : 0) {
        // whole streams (or long runs of windows): the frames staged once per workgroup; then the listed rows with the f32 instructions
        if (S * n_win > 0xffffffffULL) return hipErrorInvalidValue;
        if (hipError_t e = form == 1 ? launch_mlp_windows<false>(st, m, mfcc, S, n_frames, n_win, mean, wsum, out, redo, pitch)
                                     : launch_mlp_windows_wide<false>(st, m, mfcc, S, n_win, mean, wsum, out, redo, pitch); e != hipSuccess)
            return mlp_redo_abort(st, redo, e);
        hipError_t e2 = hipErrorInvalidValue;   // (launch_mlp_nt puts the words back itself when ITS launch fails)
        switch (m.nt) {
        case 1: e2 = launch_mlp_nt<1>(st, m, mfcc, B, kMlpRedoF32, out, redo, (size_t)K, n_win, skip, mean, wsum, K); break;
        case 2: e2 = launch_mlp_nt<2>(st, m, mfcc, B, kMlpRedoF32, out, redo, (size_t)K, n_win, skip, mean, wsum, K); break;
        case 5: e2 = launch_mlp_nt<5>(st, m, mfcc, B, kMlpRedoF32, out, redo, (size_t)K, n_win, skip, mean, wsum, K); break;
        case 9: e2 = launch_mlp_nt<9>(st, m, mfcc, B, kMlpRedoF32, out, redo, (size_t)K, n_win, skip, mean, wsum, K); break;
        }
        return e2 == hipSuccess ? e2 : mlp_redo_abort(st, redo, e2);
    }
    const int prec = precision;
    switch (m.nt) {
    case 1: return launch_mlp_nt<1>(st, m, mfcc, B, prec, out, redo, (size_t)K, n_win, skip, mean, wsum, K);
    case 2: return launch_mlp_nt<2>(st, m, mfcc, B, prec, out, redo, (size_t)K, n_win, skip, mean, wsum, K);
    case 5: return launch_mlp_nt<5>(st, m, mfcc, B, prec, out, redo, (size_t)K, n_win, skip, mean, wsum, K);
    case 9: return launch_mlp_nt<9>(st, m, mfcc, B, prec, out, redo, (size_t)K, n_win, skip, mean, wsum, K);
    }
    return hipErrorInvalidValue;
}

}  // namespace rp
