/* tools/ac3mi_loop.c — what ONE stream gets through the drop-in calls: the inner loops of the ACM driver, plain C against
 * include/ac3mi_dropin.h, timed.
 *
 *   decode  a52_syncinfo -> a52_frame -> 6 x (a52_block -> MapTab converter on a52_samples())   src/AC3ACM.cpp:1498-1581
 *   encode  AC3_encode_frame(dst, pcm, chmap) per 1536 samples per channel                        src/AC3ACM.cpp:1762
 *
 *   ac3mi_loop <frames.ac3> [repeats]     the file: whole 5.1 AC-3 frames back to back (one stream)
 * prints one JSON line: frames per second of the decode loop, of the encode loop (on the decoded PCM) and of both in turn
 * (frame by frame, as a transcoding driver would call them).  Every call is one launch sequence for one frame with its
 * PCIe copies - the latency-bound end of the engine; batches are what ac3mi.h is for.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ac3mi_dropin.h"

/* the reference declares the encoder with C++ linkage (src/ac3enc/ac3enc.h:6-7); a C host takes the library's C aliases */
#define AC3_encode_init ac3mi_AC3_encode_init
#define AC3_encode_frame ac3mi_AC3_encode_frame

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static int nchans(int flags)
{
    static const int n[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};
    return n[flags & A52_CHANNEL_MASK] + ((flags & A52_LFE) ? 1 : 0);
}

/* one frame through the decode loop: 1536 x nch interleaved s16 in WAVE order; returns nch, 0 on error */
static int decode_frame(a52_state_t *st, uint8_t *buf, int16_t *pcm)
{
    int flags, sr, br, b;
    level_t level = 1;
    if (!a52_syncinfo(buf, &flags, &sr, &br)) return 0;
    flags |= A52_ADJUST_LEVEL;
    if (a52_frame(st, buf, &flags, &level, 384)) return 0;
    const int n = nchans(flags);
    ConvertProc cv = MapTab[IsMMX() ? 1 : 0][n - 1][n - 1];
    if (!cv) return 0;
    for (b = 0; b < 6; b++) {
        if (a52_block(st)) return 0;
        cv(a52_samples(st), pcm + b * 256 * n, flags);
    }
    return n;
}

int main(int argc, char **argv)
{
    static unsigned char chmap6[8] = {0, 2, 1, 4, 5, 3, 0, 0};       /* src/AC3ACM.cpp:1631-1662 */
    if (argc < 2) { fprintf(stderr, "usage: ac3mi_loop frames.ac3 [repeats]\n"); return 2; }
    const int repeats = argc > 2 ? atoi(argv[2]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *data = malloc(size + 64);
    if (!data || fread(data, 1, size, f) != (size_t)size) { fprintf(stderr, "read failed\n"); return 2; }
    memset(data + size, 0, 64);
    fclose(f);
    int flags, sr, br;
    const int fb = a52_syncinfo(data, &flags, &sr, &br);
    if (!fb || size % fb) { fprintf(stderr, "not a whole number of equal AC-3 frames\n"); return 3; }
    const int nfr = (int)(size / fb);
    a52_state_t *st = a52_init(0);
    if (!st) { fprintf(stderr, "a52_init failed (no GPU?)\n"); return 4; }
    int16_t *pcm = malloc((size_t)nfr * 1536 * 6 * sizeof(int16_t));
    unsigned char *out = malloc(4096);
    int i, r, nch = 0;
    /* warm-up: first launches, workspaces */
    for (i = 0; i < nfr && i < 4; i++) nch = decode_frame(st, data + (size_t)i * fb, pcm + (size_t)i * 1536 * 6);
    if (!nch) { fprintf(stderr, "decode failed\n"); return 5; }
    const int efb = AC3_encode_init(sr, br, nch);
    if (!efb) { fprintf(stderr, "AC3_encode_init rejected %d Hz %d bps %d ch\n", sr, br, nch); return 6; }
    for (i = 0; i < 4 && i < nfr; i++) AC3_encode_frame(out, pcm + (size_t)i * 1536 * nch, chmap6);

    double t0 = now();
    for (r = 0; r < repeats; r++)
        for (i = 0; i < nfr; i++)
            if (!decode_frame(st, data + (size_t)i * fb, pcm + (size_t)i * 1536 * nch)) { fprintf(stderr, "decode failed\n"); return 5; }
    const double t_dec = now() - t0;
    t0 = now();
    for (r = 0; r < repeats; r++)
        for (i = 0; i < nfr; i++)
            if (AC3_encode_frame(out, pcm + (size_t)i * 1536 * nch, chmap6) != efb) { fprintf(stderr, "encode failed\n"); return 7; }
    const double t_enc = now() - t0;
    t0 = now();
    for (r = 0; r < repeats; r++)
        for (i = 0; i < nfr; i++) {
            if (!decode_frame(st, data + (size_t)i * fb, pcm)) return 5;
            if (AC3_encode_frame(out, pcm, chmap6) != efb) return 7;
        }
    const double t_both = now() - t0;
    const double n = (double)nfr * repeats;
    printf("{\"frames\": %d, \"repeats\": %d, \"channels\": %d, \"decode_frames_per_s\": %.1f, \"encode_frames_per_s\": %.1f, "
           "\"decode_plus_encode_frames_per_s\": %.1f, \"realtime_x\": %.2f, "
           "\"note\": \"one stream, one frame per call through a52_syncinfo / a52_frame / 6 x a52_block + MapTab and AC3_encode_frame "
           "(plain-C host, tools/ac3mi_loop.c): launch sequence and PCIe copies per frame\"}\n",
           nfr, repeats, nch, n / t_dec, n / t_enc, n / t_both, n / t_both * 0.032);
    a52_free(st);
    free(pcm); free(out); free(data);
    return 0;
}
