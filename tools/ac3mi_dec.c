/* tools/ac3mi_dec.c — a52dec-like file decoder on libac3mi.so.
 *
 * Same command line, stream handling and output formats as the reference tool
 *   a52dec-0.7.5-cvs/src/a52dec.c:130-238 (usage, handle_args), :240-310 (a52_decode_data), :600-640 (es_loop)
 *   a52dec-0.7.5-cvs/libao/audio_out_wav.c:60-213 (wav / wavdolby / wav6), audio_out_float.c, audio_out_null.c,
 *   a52dec-0.7.5-cvs/libao/convert2s16.c:33-41, 199-306 (convert2s16_wav)
 * so that whole files can be compared against the upstream tool, plus its three demultiplexers:
 *   a52dec.c:312-540 (demux: MPEG-1/2 program streams, private stream 1 sub-stream 0x80+track; PES-only mode),
 *   a52dec.c:554-592 (ts_loop: 188-byte transport packets of one PID carrying that PES).  Written against include/ac3mi_dropin.h (liba52's API): the program is
 * plain C and knows nothing about the GPU.
 *
 *   ac3mi_dec [-o wav|wavdolby|wav6|float|null|null4|null6] [-s [<track>]] [-t <pid>] [-T] [-r] [-a] [-g <gain dB>]
 *             [-c] [<file>]  > out
 */
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ac3mi_dropin.h"

#define A52_CHANNEL 0
#define A52_MONO 1
#define A52_STEREO 2
#define A52_3F 3
#define A52_2F1R 4
#define A52_3F1R 5
#define A52_2F2R 6
#define A52_3F2R 7
#define A52_CHANNEL1 8
#define A52_CHANNEL2 9
#define A52_DOLBY 10
#define A52_CHANNEL_MASK 15
#define A52_LFE 16
#define A52_ADJUST_LEVEL 32

enum { OUT_WAV, OUT_FLOAT, OUT_NULL };

static int out_kind = OUT_WAV;
static int out_flags = A52_STEREO;      /* requested configuration; -1 = whatever the stream carries (wav6) */
static int disable_dynrng, disable_adjust;
static double gain = 1;
static FILE *in_file;

/* ---- wav driver state (audio_out_wav.c:35-42) ---- */
static int wav_sample_rate, wav_set_params = 1, wav_size;
static uint32_t wav_speaker_flags;

static void store4(uint8_t *b, uint32_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 24); }
static void store2(uint8_t *b, uint32_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); }

/* RIFF headers: 44 bytes WAVE_FORMAT_PCM for front-only layouts, 68 bytes WAVE_FORMAT_EXTENSIBLE otherwise.
 * Until the sizes are known they hold 0xffffffff minus the header bytes that follow the field. */
static int wav_header(uint8_t *h, int chans, int rate, uint32_t speakers, int data_bytes)
{
    const int plain = speakers == 3 || speakers == 4;
    const int n = plain ? 44 : 68;
    static const uint8_t guid_pcm[16] = {1, 0, 0, 0, 0, 0, 0x10, 0x00, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71};
    memset(h, 0, (size_t)n);
    memcpy(h, "RIFF", 4);
    store4(h + 4, data_bytes < 0 ? 0xffffffffu - 3 - (plain ? 0 : 12) : (uint32_t)(data_bytes + n - 8));
    memcpy(h + 8, "WAVEfmt ", 8);
    store4(h + 16, plain ? 16 : 40);
    store2(h + 20, plain ? 1 : 0xfffe);
    store2(h + 22, (uint32_t)chans);
    store4(h + 24, (uint32_t)rate);
    store4(h + 28, (uint32_t)(rate * 2 * chans));
    store2(h + 32, (uint32_t)(2 * chans));
    store2(h + 34, 16);
    if (!plain) {
        store2(h + 36, 22);
        store2(h + 38, 16);
        store4(h + 40, speakers);
        memcpy(h + 44, guid_pcm, 16);
    }
    memcpy(h + n - 8, "data", 4);
    store4(h + n - 4, data_bytes < 0 ? (plain ? 0xffffffd8u : 0xffffffb4u) : (uint32_t)data_bytes);
    return n;
}

static int wav_channels(int flags, uint32_t *speakers)
{
    static const uint16_t spk[11] = {3, 4, 3, 7, 0x103, 0x107, 0x33, 0x37, 4, 4, 3};
    static const uint8_t nf[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};
    int chans = nf[flags & A52_CHANNEL_MASK];
    *speakers = spk[flags & A52_CHANNEL_MASK];
    if (flags & A52_LFE) { *speakers |= 8; chans++; }
    return chans;
}

static int16_t cvt(int32_t i)                   /* convert2s16.c:33-41: bias 384, level 1 */
{
    i -= 0x43c00000;
    return (int16_t)((i > 32767) ? 32767 : ((i < -32768) ? -32768 : i));
}

/* convert2s16_wav (convert2s16.c:199-306): plane order of liba52 -> WAVE channel order */
static void to_s16_wav(const float *samples, int16_t *s16, int flags)
{
    const int32_t *f = (const int32_t *)samples;
    static const int8_t order[32][6] = {
        /* no LFE: flags 0..10 */
        [A52_CHANNEL] = {0, 1, -1}, [A52_MONO] = {0, -1}, [A52_STEREO] = {0, 1, -1}, [A52_3F] = {0, 2, 1, -1},
        [A52_2F1R] = {0, 1, 2, -1}, [A52_3F1R] = {0, 2, 1, 3, -1}, [A52_2F2R] = {0, 1, 2, 3, -1}, [A52_3F2R] = {0, 2, 1, 3, 4, -1},
        [A52_CHANNEL1] = {0, -1}, [A52_CHANNEL2] = {0, -1}, [A52_DOLBY] = {0, 1, -1},
        /* with LFE (plane 0) */
        [A52_CHANNEL | 16] = {1, 2, 0, -1}, [A52_MONO | 16] = {1, 0, -1}, [A52_STEREO | 16] = {1, 2, 0, -1},
        [A52_3F | 16] = {1, 3, 2, 0, -1}, [A52_2F1R | 16] = {1, 2, 0, 3, -1}, [A52_3F1R | 16] = {1, 3, 2, 0, 4, -1},
        [A52_2F2R | 16] = {1, 2, 0, 3, 4, -1}, [A52_3F2R | 16] = {1, 3, 2, 0, 4, 5},
        [A52_CHANNEL1 | 16] = {1, 0, -1}, [A52_CHANNEL2 | 16] = {1, 0, -1}, [A52_DOLBY | 16] = {1, 2, 0, -1}};
    int fl = flags & 31, n = 0;
    /* the reference's 2F1R+LFE case has no break and runs on into 3F1R+LFE (convert2s16.c:270-285): same bytes here */
    if (fl == (A52_2F1R | A52_LFE)) fl = A52_3F1R | A52_LFE;
    while (n < 6 && order[fl][n] >= 0) n++;
    for (int i = 0; i < 256; i++)
        for (int c = 0; c < n; c++) s16[n * i + c] = cvt(f[i + 256 * order[fl][c]]);
}

static int output_setup(int sample_rate, int *flags, float *level, float *bias)
{
    if (out_kind == OUT_FLOAT) { *flags = A52_STEREO; *level = 1; *bias = 0; return 0; }
    if (out_kind == OUT_WAV) {
        if (wav_set_params == 0 && wav_sample_rate != sample_rate) return 1;
        wav_sample_rate = sample_rate;
    }
    if (out_flags >= 0) *flags = out_flags;
    *level = 1;
    *bias = 384;
    return 0;
}

static int output_play(int flags, const float *samples)
{
    if (out_kind == OUT_NULL) return 0;
    if (out_kind == OUT_FLOAT) { fwrite(samples, sizeof(float), 256 * 2, stdout); return 0; }
    int16_t s16[256 * 6];
    uint8_t le[256 * 6 * 2], hdr[68];
    uint32_t speakers;
    const int chans = wav_channels(flags, &speakers);
    if (wav_set_params) {
        wav_set_params = 0;
        wav_speaker_flags = speakers;
        fwrite(hdr, (size_t)wav_header(hdr, chans, wav_sample_rate, speakers, -1), 1, stdout);
    } else if (speakers != wav_speaker_flags) {
        return 1;
    }
    to_s16_wav(samples, s16, flags);
    for (int i = 0; i < 256 * chans; i++) store2(le + 2 * i, (uint16_t)s16[i]);
    fwrite(le, (size_t)(256 * 2 * chans), 1, stdout);
    wav_size += 256 * 2 * chans;
    return 0;
}

static void output_close(void)
{
    uint8_t hdr[68];
    uint32_t speakers = wav_speaker_flags;
    if (out_kind != OUT_WAV || wav_set_params) return;
    fflush(stdout);
    if (fseek(stdout, 0, SEEK_SET) < 0) return;         /* a pipe keeps the placeholder sizes */
    int chans = 0;
    for (uint32_t m = speakers; m; m &= m - 1) chans++;
    fwrite(hdr, (size_t)wav_header(hdr, chans, wav_sample_rate, speakers, wav_size), 1, stdout);
}

static void usage(const char *argv0)
{
    fprintf(stderr, "usage: %s [-h] [-o <mode>] [-s [<track>]] [-t <pid>] [-T] [-c] [-r] [-a] [-g <gain>] <file>\n"
                    "\t-h\tdisplay help and available audio output modes\n"
                    "\t-s\tuse program stream demultiplexer, track 0-7 or 0x80-0x87\n"
                    "\t-t\tuse transport stream demultiplexer, pid 0x10-0x1ffe\n"
                    "\t-T\tuse transport stream PES demultiplexer\n"
                    "\t-c\taccepted for compatibility (no accelerations to disable)\n"
                    "\t-r\tdisable dynamic range compression\n"
                    "\t-a\tdisable level adjustment based on output mode\n"
                    "\t-g\tadd specified gain in decibels, -96.0 to +96.0\n"
                    "\t-o\taudio output mode\n"
                    "\t\t\twav\n\t\t\twavdolby\n\t\t\twav6\n\t\t\tnull\n\t\t\tnull4\n\t\t\tnull6\n\t\t\tfloat\n", argv0);
    exit(1);
}

/* ---- elementary-stream framing, push model ----
 * a52dec's behaviour (a52dec.c:240-310): a frame starts wherever a52_syncinfo accepts seven bytes; a position it
 * rejects is skipped byte by byte ("skip" on stderr per byte); a frame that fails to set up or decode costs one
 * "error" and the stream resumes after it. */
static a52_state_t *state;
static uint8_t es_win[2 * 3840 + 4096];
static size_t es_have;

static void es_drain(void)
{
    size_t at = 0;
    while (es_have - at >= 7) {
        int flags, sample_rate, bit_rate;
        const int length = a52_syncinfo(es_win + at, &flags, &sample_rate, &bit_rate);
        if (!length) { fprintf(stderr, "skip\n"); at++; continue; }
        if (es_have - at < (size_t)length) break;           /* wait for the rest of the frame */
        float level, bias;
        int done = 0;
        if (!output_setup(sample_rate, &flags, &level, &bias)) {
            if (!disable_adjust) flags |= A52_ADJUST_LEVEL;
            level = (float)(level * gain);
            if (!a52_frame(state, es_win + at, &flags, &level, bias)) {
                if (disable_dynrng) a52_dynrng(state, NULL, NULL);
                while (done < 6 && !a52_block(state) && !output_play(flags, a52_samples(state))) done++;
            }
        }
        if (done != 6) fprintf(stderr, "error\n");
        at += (size_t)length;
    }
    memmove(es_win, es_win + at, es_have - at);
    es_have -= at;
}

static void es_push(const uint8_t *p, size_t n)
{
    while (n) {
        size_t room = sizeof es_win - es_have, take = n < room ? n : room;
        memcpy(es_win + es_have, p, take);
        es_have += take;
        p += take;
        n -= take;
        es_drain();
    }
}

/* ---- MPEG system layers (ISO 11172-1 / 13818-1), as far as a52dec looks at them ----
 * All three demultiplexers work on a byte queue: packets are parsed once they are complete. */
static uint8_t *q_buf;
static size_t q_len, q_cap;

static void q_append(const uint8_t *p, size_t n)
{
    if (q_len + n > q_cap) {
        q_cap = 2 * (q_len + n) + 65536;
        q_buf = realloc(q_buf, q_cap);
        if (!q_buf) { fprintf(stderr, "out of memory\n"); exit(1); }
    }
    memcpy(q_buf + q_len, p, n);
    q_len += n;
}
static void q_drop(size_t n) { memmove(q_buf, q_buf + n, q_len - n); q_len -= n; }

/* length of the PES header that ends right before the payload, for a private-stream-1 packet at p (p[3] == 0xbd);
 * 0 = need more bytes (have = bytes available) */
static size_t pes_header_len(const uint8_t *p, size_t have)
{
    if (have < 7) return 0;
    if ((p[6] & 0xc0) == 0x80) {                            /* MPEG-2: flags, flags, header_data_length */
        if (have < 9) return 0;
        return 9 + (size_t)p[8];
    }
    size_t len = 6;                                         /* MPEG-1: stuffing, STD buffer, PTS/DTS */
    int stuffing = 0;
    while (1) {
        if (have <= len) return 0;
        if (p[len] != 0xff) break;
        len++;
        if (++stuffing == 17) { fprintf(stderr, "too much stuffing\n"); break; }
    }
    if ((p[len] & 0xc0) == 0x40) { len += 2; if (have <= len) return 0; }
    static const int skip[16] = {0, 0, 4, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return len + (size_t)skip[p[len] >> 4] + 1;
}

/* program stream (track >= 0x80: private stream 1 sub-stream `track`) or bare PES stream of 0xbd packets (track < 0).
 * Returns 1 at a program end code. */
static int ps_parse(int track, int at_eof)
{
    for (;;) {
        size_t i = 0;
        while (i + 4 <= q_len && !(q_buf[i] == 0 && q_buf[i + 1] == 0 && q_buf[i + 2] == 1)) i++;
        if (i) q_drop(i);
        if (q_len < 4) return 0;
        const uint8_t *p = q_buf;
        const int id = p[3];
        size_t total;
        if (track < 0 && id != 0xbd) { fprintf(stderr, "bad stream id %x\n", id); exit(1); }
        if (id == 0xb9 && track >= 0) return 1;
        if (id == 0xba) {                                   /* pack header */
            if (q_len < 5) return 0;
            if ((p[4] & 0xc0) == 0x40) { if (q_len < 14) return 0; total = 14 + (size_t)(p[13] & 7); }
            else if ((p[4] & 0xf0) == 0x20) total = 12;
            else { fprintf(stderr, "weird pack header\n"); total = 5; }
            if (q_len < total) return 0;
            q_drop(total);
            continue;
        }
        if (id < 0xb9) { fprintf(stderr, "looks like a video stream, not system stream\n"); exit(1); }
        if (q_len < 6) return 0;
        total = 6 + (((size_t)p[4] << 8) | p[5]);
        if (q_len < total) {
            if (!at_eof) return 0;
            total = q_len;                                  /* truncated last packet: use what is there */
        }
        if (id == 0xbd) {
            size_t hl = pes_header_len(p, total);
            if (track < 0 && hl && (p[6] & 0xc0) != 0x80) { fprintf(stderr, "bad multiplex - not mpeg2\n"); exit(1); }
            if (hl && hl < total) {
                if (track < 0) es_push(p + hl, total - hl);
                else if (p[hl] == track && hl + 4 <= total) es_push(p + hl + 4, total - hl - 4);   /* id, count, pointer */
            }
        }
        q_drop(total);
        if (at_eof && q_len < 4) return 0;
    }
}

static void run_ps(FILE *in, int track)
{
    static uint8_t chunk[4096];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, in)) > 0) {
        q_append(chunk, got);
        if (ps_parse(track, 0)) return;
    }
    ps_parse(track, 1);
}

/* transport stream: the payloads of one PID form a PES stream of private-stream-1 packets (a52dec.c:554-592) */
static void run_ts(FILE *in, int pid)
{
    static uint8_t buf[188 * 64];
    size_t have = 0, got;
    int in_payload = 0;                                     /* inside a PES packet's data */
    while ((got = fread(buf + have, 1, sizeof buf - have, in)) > 0 || have >= 188) {
        have += got;
        size_t at = 0;
        while (at + 188 <= have) {
            const uint8_t *t = buf + at;
            if (t[0] != 0x47) { fprintf(stderr, "bad sync byte\n"); at++; continue; }
            at += 188;
            if ((((t[1] << 8) | t[2]) & 0x1fff) != pid) continue;
            const uint8_t *data = t + 4;
            if (t[3] & 0x20) { data = t + 5 + t[4]; if (data > t + 188) continue; }
            if (!(t[3] & 0x10)) continue;
            size_t n = (size_t)(t + 188 - data);
            if (t[1] & 0x40) {                              /* payload_unit_start: a PES header comes first */
                if (n < 9 || data[0] || data[1] || data[2] != 1) { in_payload = 0; continue; }
                if (data[3] != 0xbd) { fprintf(stderr, "bad stream id %x\n", data[3]); exit(1); }
                if ((data[6] & 0xc0) != 0x80) { fprintf(stderr, "bad multiplex - not mpeg2\n"); exit(1); }
                const size_t hl = 9 + (size_t)data[8];
                in_payload = 1;
                if (hl < n) es_push(data + hl, n - hl);
            } else if (in_payload) {
                es_push(data, n);
            }
        }
        memmove(buf, buf + at, have - at);
        have -= at;
        if (got == 0) break;
    }
}

int main(int argc, char **argv)
{
    static const struct { const char *name; int kind, flags; } modes[] = {
        {"wav", OUT_WAV, A52_STEREO}, {"wavdolby", OUT_WAV, A52_DOLBY}, {"wav6", OUT_WAV, -1}, {"null", OUT_NULL, A52_STEREO},
        {"null4", OUT_NULL, A52_2F2R}, {"null6", OUT_NULL, A52_3F2R | A52_LFE}, {"float", OUT_FLOAT, A52_STEREO}};
    int c, demux_track = 0, demux_pid = 0, demux_pes = 0;
    char *s;
    while ((c = getopt(argc, argv, "hs::t:Tcrag:o:")) != -1) switch (c) {
        case 'o': {
            int found = 0;
            for (size_t i = 0; i < sizeof modes / sizeof modes[0]; i++)
                if (strcmp(modes[i].name, optarg) == 0) { out_kind = modes[i].kind; out_flags = modes[i].flags; found = 1; }
            if (!found) { fprintf(stderr, "Invalid video driver: %s\n", optarg); usage(argv[0]); }
            break;
        }
        case 's':
            demux_track = 0x80;
            if (optarg) {
                demux_track = (int)strtol(optarg, &s, 0);
                if (demux_track < 0x80) demux_track += 0x80;
                if (demux_track < 0x80 || demux_track > 0x87 || *s) { fprintf(stderr, "Invalid track number: %s\n", optarg); usage(argv[0]); }
            }
            break;
        case 't':
            demux_pid = (int)strtol(optarg, &s, 0);
            if (demux_pid < 0x10 || demux_pid > 0x1ffe || *s) { fprintf(stderr, "Invalid pid: %s\n", optarg); usage(argv[0]); }
            break;
        case 'T': demux_pes = 1; break;
        case 'c': break;
        case 'r': disable_dynrng = 1; break;
        case 'a': disable_adjust = 1; break;
        case 'g':
            gain = strtod(optarg, &s);
            if (gain < -96 || gain > 96 || *s) { fprintf(stderr, "Invalid gain: %s\n", optarg); usage(argv[0]); }
            gain = pow(2, gain / 6);
            break;
        default: usage(argv[0]);
    }
    in_file = stdin;
    if (optind < argc && !(in_file = fopen(argv[optind], "rb"))) {
        fprintf(stderr, "%s - could not open file %s\n", strerror(errno), argv[optind]);
        return 1;
    }
    state = a52_init(0);
    if (!state) { fprintf(stderr, "A52 init failed\n"); return 1; }

    if (demux_pid) run_ts(in_file, demux_pid);
    else if (demux_track) run_ps(in_file, demux_track);
    else if (demux_pes) run_ps(in_file, -1);
    else {
        static uint8_t chunk[4096];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, in_file)) > 0) es_push(chunk, got);
    }
    output_close();
    a52_free(state);
    return 0;
}
