/* detect_wav.c -- the single-stream drop-in API from plain C: what a host binding (Rust FFI, JNI, cgo ...) does per
 * 30 ms chunk.  Mirrors the reference's own usage (README.md "Basic usage", src/detector.rs:95-254):
 *   Rustpotter::new(&config) -> add_wakeword_from_file(key, path) -> process_bytes(chunk) per frame.
 *
 *   gcc -std=c99 -Iinclude examples/detect_wav.c -Lrustpotter_amd -lrustpotter_hip -Wl,-rpath,$PWD/rustpotter_amd -o detect_wav
 *   ./detect_wav tests/golden/oye_casa_g.rpw tests/golden/oye_casa_g_1.wav
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rustpotter_hip.h"

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <wakeword.rpw> <16 kHz mono i16 .wav> [threshold]\n", argv[0]); return 2; }
    rp_config cfg;
    rp_config_default(&cfg);
    cfg.fmt.sample_format = RP_SAMPLE_I16;
    if (argc > 3) cfg.detector.threshold = (float)atof(argv[3]);
    rp_detector *d = NULL;
    if (rp_new(&cfg, &d) < 0) { fprintf(stderr, "rp_new: %s\n", rp_last_error()); return 1; }
    if (rp_add_wakeword_from_file(d, "wakeword", argv[1]) < 0) { fprintf(stderr, "add_wakeword: %s\n", rp_last_error()); rp_free(d); return 1; }

    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); rp_free(d); return 1; }
    fseek(f, 44, SEEK_SET); /* canonical 44-byte RIFF header, like the reference's tests (tests/detector.rs:372-399) */
    const size_t chunk = rp_get_bytes_per_frame(d);
    unsigned char *buf = (unsigned char *)malloc(chunk);
    long n_chunks = 0, n_det = 0;
    /* one second of silence in front of and behind the recording, so that a template-length window exists around the utterance */
    for (int part = 0; part < 3; ++part) {
        for (int i = 0; part == 1 || i < 34; ++i) {
            if (part == 1) { if (fread(buf, 1, chunk, f) < chunk) break; }
            else memset(buf, 0, chunk);
            rp_detection det;
            const int r = rp_process_bytes(d, buf, chunk, &det);
            if (r < 0) { fprintf(stderr, "process_bytes: %s\n", rp_last_error()); free(buf); fclose(f); rp_free(d); return 1; }
            if (r == 1) {
                printf("chunk %ld: detection \"%s\" score %.7f avg_score %.7f counter %zu\n", n_chunks, det.name, det.score, det.avg_score, det.counter);
                for (size_t k = 0; k < det.n_scores; ++k) printf("    %s = %.7f\n", det.score_names[k], det.scores[k]);
                ++n_det;
            }
            ++n_chunks;
        }
    }
    printf("%ld chunks of %zu bytes, %ld detection(s)\n", n_chunks, chunk, n_det);
    free(buf);
    fclose(f);
    rp_free(d);
    return 0;
}
