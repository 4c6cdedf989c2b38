/* sharded_batch.c -- the batched operators from plain C, streams sharded over the GPUs of a node:
 * one context + one replica of the wakeword per device, rp_batch_detect_sharded runs one host thread per shard and
 * gathers every stream's detections into one host block.  Synthetic input (the benchmark's generator), so that it runs
 * anywhere; with one GPU the shards share it.
 *
 *   gcc -std=c99 -Iinclude examples/sharded_batch.c -Lrustpotter_amd -lrustpotter_hip -lm -Wl,-rpath,$PWD/rustpotter_amd -o sharded_batch
 *   ./sharded_batch [n_shards] [streams_per_shard]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rustpotter_hip.h"

#define SEED 0x5EED000000000001ull
enum { K = 5, T = 4, L = 60, N = 480 * 100 };

#define CHECK(call) do { if ((call) < 0) { fprintf(stderr, "%s: %s\n", #call, rp_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n_shards = argc > 1 ? atoi(argv[1]) : 2;
    const size_t per = argc > 2 ? (size_t)atol(argv[2]) : 256;
    if (n_shards < 1 || n_shards > 16) return 2;
    rp_ctx *ctx[16];
    rp_templates *tm[16];
    const rp_templates *ctm[16];
    float *pcm[16];
    const void *cpcm[16];
    size_t S[16], total = 0;

    /* templates: MFCC of T synthetic utterances, whole-matrix mean normalisation (src/mfcc/wav_file_extractor.rs:59-67), cut to
     * L frames; one of them is also planted into stream 0 of every shard so that something is found */
    const size_t nt = 480 * ((L + 3 + 2) / 3), ntf = rp_mfcc_num_frames(nt);
    float *utt = (float *)malloc(T * nt * sizeof(float)), *feat = (float *)malloc(T * ntf * K * sizeof(float)), *tfeat = (float *)malloc(T * L * K * sizeof(float));
    int lens[T];
    for (int g = 0; g < n_shards; ++g) {
        CHECK(rp_ctx_new(0 /* device ordinal: g on a multi-GPU node */, RP_CTX_HOST_POINTERS, &ctx[g]));
        if (g == 0) {
            for (int t = 0; t < T; ++t) CHECK(rp_synth_pcm_batch(ctx[0], SEED + 1 + t, 0, 1, nt, nt, utt + t * nt));
            CHECK(rp_mfcc_batch(ctx[0], utt, T, nt, nt, K, feat));
            for (int t = 0; t < T; ++t) {
                float mean[K] = {0};
                for (size_t f = 0; f < ntf; ++f) for (int k = 0; k < K; ++k) mean[k] += feat[(t * ntf + f) * K + k];
                for (int k = 0; k < K; ++k) mean[k] /= (float)ntf;
                for (int f = 0; f < L; ++f) for (int k = 0; k < K; ++k) tfeat[(t * L + f) * K + k] = feat[(t * ntf + f) * K + k] - mean[k];
                lens[t] = L;
            }
        }
        CHECK(rp_templates_new(ctx[g], T, K, lens, tfeat, 0, NULL, &tm[g]));
        ctm[g] = tm[g];
        S[g] = per + (size_t)g; /* ragged shards */
        pcm[g] = (float *)malloc(S[g] * N * sizeof(float));
        CHECK(rp_synth_pcm_batch(ctx[g], SEED, total, S[g], N, N, pcm[g]));
        memcpy(pcm[g] + 480 * 20, utt, nt * sizeof(float)); /* stream 0 of the shard hears utterance 0 */
        cpcm[g] = pcm[g];
        total += S[g];
    }
    rp_config cfg;
    rp_config_default(&cfg);
    cfg.detector.avg_threshold = 0.f;
    const int max_det = 4;
    rp_batch_detection *det = (rp_batch_detection *)calloc(total * max_det, sizeof(*det));
    int32_t *n_det = (int32_t *)calloc(total, sizeof(*n_det));
    CHECK(rp_batch_detect_sharded(ctx, ctm, n_shards, cpcm, RP_SAMPLE_F32, S, N, N, &cfg.detector, det, n_det, max_det));
    size_t found = 0, first = 0;
    for (int g = 0; g < n_shards; ++g) {
        for (size_t s = first; s < first + S[g]; ++s)
            for (int j = 0; j < n_det[s] && j < max_det; ++j) {
                const rp_batch_detection *x = &det[s * max_det + j];
                printf("shard %d stream %d (global %zu): frame %d window %d counter %d score %.6f\n", g, (int)(s - first), s, x->frame, x->window, x->counter, x->score);
                if (x->stream != (int32_t)s) { fprintf(stderr, "stream id mismatch\n"); return 1; }
                ++found;
            }
        first += S[g];
    }
    printf("%zu streams in %d shards, %zu detection(s)\n", total, n_shards, found);
    printf("gather: %s\n", rp_sharded_gather_info());
    /* the same streams once more as ONE host array through the pipelined host entry point (blocks of 100 streams, the next block's
     * copy under this block's kernels): the same detections */
    {
        float *all = (float *)malloc(total * N * sizeof(float));
        size_t at = 0;
        for (int g = 0; g < n_shards; ++g) { memcpy(all + at * N, pcm[g], S[g] * N * sizeof(float)); at += S[g]; }
        rp_batch_detection *det2 = (rp_batch_detection *)calloc(total * max_det, sizeof(*det2));
        int32_t *n_det2 = (int32_t *)calloc(total, sizeof(*n_det2));
        double seconds = 0.0;
        CHECK(rp_batch_detect_ingest(ctx[0], all, RP_SAMPLE_F32, total, N, N, tm[0], &cfg.detector, det2, n_det2, max_det, 100, &seconds));
        if (memcmp(n_det, n_det2, total * sizeof(*n_det)) != 0 || memcmp(det, det2, total * max_det * sizeof(*det)) != 0) {
            fprintf(stderr, "rp_batch_detect_ingest differs from rp_batch_detect_sharded\n");
            return 1;
        }
        printf("rp_batch_detect_ingest: the same %zu detection(s) from host memory in %.1f ms (build %s)\n", found, seconds * 1e3, rp_build_info());
        free(all); free(det2); free(n_det2);
    }
    for (int g = 0; g < n_shards; ++g) { rp_templates_free(tm[g]); rp_ctx_free(ctx[g]); free(pcm[g]); }
    free(det); free(n_det); free(utt); free(feat); free(tfeat);
    return found >= (size_t)n_shards ? 0 : 1; /* the planted utterance is found in every shard */
}
