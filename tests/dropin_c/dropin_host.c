/* tests/dropin_c/dropin_host.c — a plain-C host written against the REFERENCE's call surface
 * (a52_init / a52_syncinfo / a52_frame / a52_dynrng / a52_block / a52_samples / a52_free, the
 * ConvertProc table, AC3_encode_init / AC3_encode_frame) and linked against libac3mi.so.
 * It is the inner loop of stream_convert_ac3 (src/AC3ACM.cpp:1498-1581) and of a52dec
 * (a52dec-0.7.5-cvs/src/a52dec.c:270-305) and of stream_convert_pcm (src/AC3ACM.cpp:1762),
 * with files instead of ACM buffers.
 *
 *   dropin_host dec <in.ac3> <out.f32> <out.s16> <flags> <level> <bias> <dynrng: 0 stream's own, 1 off, 2 callback>
 *   dropin_host enc <in.s16> <out.ac3> <freq> <bitrate> <channels>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ac3mi_dropin.h"

static int nchans(int flags)
{
    static const int n[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};
    return n[flags & A52_CHANNEL_MASK] + ((flags & A52_LFE) ? 1 : 0);
}

/* a dynamic-range callback as a52dec's -c / compression hooks are written (a52dec.c:291): exact in float */
static int callback_calls;
static level_t halve_range(level_t range, void *data)
{
    callback_calls++;
    return range * *(float *)data;
}

static int decode(int argc, char **argv)
{
    static float half = 0.5f;
    FILE *in = fopen(argv[2], "rb"), *of = fopen(argv[3], "wb"), *os = fopen(argv[4], "wb");
    int req = atoi(argv[5]), dynoff = atoi(argv[8]);
    float level0 = (float)atof(argv[6]), bias = (float)atof(argv[7]);
    static uint8_t buf[3840 + 16];
    a52_state_t *st = a52_init(0);
    int errors = 0, frames = 0;
    (void)argc;
    if (!in || !of || !os || !st) { fprintf(stderr, "setup failed\n"); return 2; }
    while (fread(buf, 1, 8, in) == 8) {
        int flags, sr, br, len = a52_syncinfo(buf, &flags, &sr, &br);
        level_t level = level0;
        int b;
        if (!len) { fprintf(stderr, "lost sync\n"); return 3; }
        if (fread(buf + 8, 1, len - 8, in) != (size_t)(len - 8)) break;
        flags = req;
        if (a52_frame(st, buf, &flags, &level, bias)) { errors++; continue; }
        if (dynoff == 1) a52_dynrng(st, NULL, NULL);
        if (dynoff == 2) a52_dynrng(st, halve_range, &half);
        for (b = 0; b < 6; b++) {
            int n = nchans(flags);
            if (a52_block(st)) { errors++; break; }
            fwrite(a52_samples(st), sizeof(float), 256 * n, of);
            if (bias == 384.0f) {
                static int16_t pcm[256 * 6];
                ConvertProc cv = MapTab[IsMMX()][n - 1][n - 1];
                if (cv) { cv(a52_samples(st), pcm, flags); fwrite(pcm, 2, 256 * n, os); }
            }
        }
        frames++;
    }
    a52_free(st);
    fclose(in); fclose(of); fclose(os);
    printf("frames %d errors %d callback calls %d\n", frames, errors, callback_calls);
    return errors ? 1 : 0;
}

static int encode(char **argv)
{
    FILE *in = fopen(argv[2], "rb"), *out = fopen(argv[3], "wb");
    int freq = atoi(argv[4]), bitrate = atoi(argv[5]), ch = atoi(argv[6]);
    unsigned char chmap6[8] = {0, 2, 1, 4, 5, 3, 0, 0}, ident[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    static short pcm[1536 * 6];
    static unsigned char frame[4096];
    int fb = ac3mi_AC3_encode_init(freq, bitrate, ch), frames = 0;
    if (!in || !out || fb <= 0) { fprintf(stderr, "setup failed\n"); return 2; }
    while (fread(pcm, 2, 1536 * ch, in) == (size_t)(1536 * ch)) {
        int n = ac3mi_AC3_encode_frame(frame, pcm, ch == 6 ? chmap6 : ident);
        if (n != fb) return 3;
        fwrite(frame, 1, n, out);
        frames++;
    }
    fclose(in); fclose(out);
    printf("frames %d bytes %d\n", frames, fb);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 9 && !strcmp(argv[1], "dec")) return decode(argc, argv);
    if (argc >= 7 && !strcmp(argv[1], "enc")) return encode(argv);
    fprintf(stderr, "usage: see source\n");
    return 64;
}
