// dropin.hip — the reference's per-stream call surface (include/ac3mi_dropin.h) on top of the
// batched engine, plus the s16 conversion kernel.  Host logic only where the reference's API hands
// back values the caller needs before any audio exists (a52_syncinfo, the flags/level negotiation of
// a52_frame); every sample is produced on the GPU.
#include "ac3mi_internal.h"
#include "a52_levels.h"
#include "../../include/ac3mi_dropin.h"
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace ac3mi {

// ---- float(bias 384) -> s16 (src/AC3ASM.asm MMX flavour) ----------------------------------------
struct CvtParams {
    const float *planes;
    int16_t *out;
    size_t n_blocks;
    int n_out;
    int map[6];         // WAVE slot -> liba52 plane
};

__global__ __launch_bounds__(256) void convert_s16_kernel(const CvtParams P)
{
    const size_t blk = blockIdx.x;
    const int n = threadIdx.x;
    if (blk >= P.n_blocks) return;
    const float *src = P.planes + blk * P.n_out * 256;
    int16_t *dst = P.out + (blk * 256 + n) * P.n_out;
    for (int c = 0; c < P.n_out; c++) {
        int v = (int)(__float_as_uint(src[P.map[c] * 256 + n]) - 0x43c00000u);   // psubd (wraps)
        v = v > 32767 ? 32767 : v < -32768 ? -32768 : v;                   // packssdw
        dst[c] = (int16_t)v;
    }
}

// WAVE slot -> liba52 plane for the output described by `flags` (see ac3mi.h)
int s16_channel_map(int flags, int map[6])
{
    static const uint8_t nfch[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};
    const int cfg = flags & 15;
    if (cfg > 10) return -1;
    const int lfe = (flags & 16) ? 1 : 0, o = lfe;
    int n = 0;
    bool lfe_done = false;
    auto put_lfe = [&]() { if (lfe && !lfe_done) { map[n++] = 0; lfe_done = true; } };
    switch (cfg) {
    case 3: map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; break;                                   // L C R
    case 4: map[n++] = o; map[n++] = o + 1; put_lfe(); map[n++] = o + 2; break;                        // L R S
    case 5: map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; put_lfe(); map[n++] = o + 3; break;      // L C R S
    case 6: map[n++] = o; map[n++] = o + 1; put_lfe(); map[n++] = o + 2; map[n++] = o + 3; break;      // L R SL SR
    case 7: map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; put_lfe(); map[n++] = o + 3; map[n++] = o + 4; break;
    default:
        for (int i = 0; i < nfch[cfg]; i++) map[n++] = o + i;                                          // mono / 2 ch
        break;
    }
    put_lfe();
    return n;
}

hipError_t launch_convert_s16(const float *planes, int16_t *out, int flags, size_t n_blocks, hipStream_t stream)
{
    CvtParams P;
    P.planes = planes;
    P.out = out;
    P.n_blocks = n_blocks;
    P.n_out = s16_channel_map(flags, P.map);
    if (P.n_out <= 0) return hipErrorInvalidValue;
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(convert_s16_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, P);
    return hipGetLastError();
}

}  // namespace ac3mi

using namespace ac3mi;

// -------------------------------------------------------------------------------------------------
// shared context of the drop-in layer (device AC3MI_DEVICE, default 0)

static std::mutex g_mu;
static ac3mi_ctx *g_ctx;
// Every a52_state_t, the encoder entry points and the MapTab converters share g_ctx (one HIP stream, its workspaces and
// its error string).  liba52 is re-entrant per state and the ACM driver decodes different streams on different client
// threads (src/AC3ACM.cpp:92-102), so calls that touch the context are serialised here; ac3enc itself is one stream
// per process in the reference as well (ENC/ac3enc.cpp:78-87).
static std::recursive_mutex g_call_mu;
#define DROPIN_LOCK() std::lock_guard<std::recursive_mutex> dropin_lock_(g_call_mu)

static ac3mi_ctx *dropin_ctx()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        const char *e = getenv("AC3MI_DEVICE");
        g_ctx = ac3mi_create(e ? atoi(e) : 0);
    }
    return g_ctx;
}

struct a52_state_s {
    float *samples;             // 12 x 256 floats like liba52 (planes 0-5 are what a52_samples() exposes)
    uint8_t frame[3840 + 8];
    int frame_bytes, have_frame, decoded, next_block;
    int req_flags, out_flags, n_out, acmod, lfeon, dynrng;
    float level_in, bias;
    uint32_t status;
    float pcm[6 * 6 * 256];
    level_t (*dyn_call)(level_t, void *);     // a52_dynrng callback of the current frame (parse.c:207-216)
    void *dyn_data;
    // device side
    uint8_t *d_frame;
    float *d_delay, *d_pcm;     // d_delay: overlap tails [6][128], then the mix state of ac3mi_set_mix_state: pending [6][128], flags [6]
    uint16_t *d_lfsr;
    uint32_t *d_status;
    float *d_dyn;               // 12 range factors out + 12 in (callback frames only)
    float *d_scratch;           // overlap tails + dither state of the look-ahead pass
};

// layout of a52_state_s::d_delay (and of d_scratch, which has the dither state behind it)
constexpr size_t ST_PENDING = 6 * 128, ST_FLAGS = 2 * 6 * 128, ST_FLOATS = 2 * 6 * 128 + 8;

extern "C" {

int ac3mi_convert_s16_batch(ac3mi_ctx *ctx, const float *d_planes, int16_t *d_out, int flags, size_t n_blocks)
{
    if (!ctx || !d_planes || !d_out) return AC3MI_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return AC3MI_ERR_HIP;
    hipError_t e = launch_convert_s16(d_planes, d_out, flags, n_blocks, ctx->stream);
    if (e != hipSuccess) {
        ctx->err = std::string("ac3mi_convert_s16_batch: ") + hipGetErrorString(e);
        return e == hipErrorInvalidValue ? AC3MI_ERR_ARG : AC3MI_ERR_HIP;
    }
    return AC3MI_OK;
}

// ---- decoder (a52.h:56-65) ------------------------------------------------------------------------

a52_state_t *a52_init(uint32_t mm_accel)
{
    (void)mm_accel;                                     // no effect in the reference either (AC3ACM.cpp:2040-2043)
    ac3mi_ctx *ctx = dropin_ctx();
    if (!ctx) return nullptr;
    DROPIN_LOCK();
    a52_state_t *st = (a52_state_t *)calloc(1, sizeof *st);
    if (!st) return nullptr;
    st->samples = (float *)calloc(256 * 12, sizeof(float));
    st->d_frame = (uint8_t *)ac3mi_dev_alloc(ctx, 3840 + 8);
    st->d_delay = (float *)ac3mi_dev_alloc(ctx, ST_FLOATS * sizeof(float));
    st->d_pcm = (float *)ac3mi_dev_alloc(ctx, sizeof st->pcm);
    st->d_lfsr = (uint16_t *)ac3mi_dev_alloc(ctx, 4);
    st->d_status = (uint32_t *)ac3mi_dev_alloc(ctx, 4);
    st->d_dyn = (float *)ac3mi_dev_alloc(ctx, 24 * sizeof(float));
    st->d_scratch = (float *)ac3mi_dev_alloc(ctx, ST_FLOATS * sizeof(float) + 16);
    if (!st->samples || !st->d_frame || !st->d_delay || !st->d_pcm || !st->d_lfsr || !st->d_status || !st->d_dyn || !st->d_scratch) {
        a52_free(st);
        return nullptr;
    }
    const uint16_t one = 1;                              // lfsr_state = 1 (parse.c:75)
    ac3mi_memset(ctx, st->d_delay, 0, ST_FLOATS * sizeof(float));
    ac3mi_memcpy_h2d(ctx, st->d_lfsr, &one, 2);
    st->dynrng = 1;
    return st;
}

sample_t *a52_samples(a52_state_t *st) { return st->samples; }

int a52_syncinfo(uint8_t *buf, int *flags, int *sample_rate, int *bit_rate)
{
    return ac3mi_syncinfo(buf, flags, sample_rate, bit_rate);
}

int a52_frame(a52_state_t *st, uint8_t *buf, int *flags, level_t *level, sample_t bias)
{
    static const float clev_tab[4] = {(float)AC3MI_G_3DB, (float)AC3MI_G_45DB, (float)AC3MI_G_6DB, (float)AC3MI_G_45DB};
    static const float slev_tab[4] = {(float)AC3MI_G_3DB, (float)AC3MI_G_6DB, 0.f, (float)AC3MI_G_6DB};
    int sflags, sr, br;
    const int n = ac3mi_syncinfo(buf, &sflags, &sr, &br);
    st->have_frame = 0;
    // negotiation of a52_frame (parse.c:131-170) on the host: the caller needs *flags / *level now
    int acmod = buf[6] >> 5;
    st->acmod = acmod;
    uint32_t bits = ((uint32_t)buf[6] << 24) | ((uint32_t)buf[7] << 16) | ((uint32_t)buf[8] << 8);
    int pos = 3;
    auto get = [&](int k) { uint32_t v = (bits << pos) >> (32 - k); pos += k; return (int)v; };
    if (acmod == 2 && get(2) == 2) acmod = 10;
    float clev = 0.f, slev = 0.f;
    if ((acmod & 1) && acmod != 1) clev = clev_tab[get(2)];
    if (acmod & 4) slev = slev_tab[get(2)];
    st->lfeon = get(1);
    st->level_in = *level;
    st->req_flags = *flags;
    int out = a52_downmix_init_hd(acmod, *flags, level, clev, slev);
    if (out < 0) return 1;
    if (st->lfeon && (*flags & AC3MI_LFE)) out |= AC3MI_LFE;
    *flags = out;
    st->out_flags = out;
    st->bias = bias;
    st->dynrng = 1;                                      // state->dynrnge = 1, state->dynrngcall = NULL (parse.c:171-172)
    st->dyn_call = nullptr;
    st->dyn_data = nullptr;
    if (n <= 0 || n > 3840) {                            // liba52 would parse garbage; we report block errors
        st->frame_bytes = 0;
        st->have_frame = 1;
        st->decoded = 1;
        st->status = 0x13f;
        st->next_block = 0;
        return 0;
    }
    memcpy(st->frame, buf, n);
    st->frame_bytes = n;
    st->have_frame = 1;
    st->decoded = 0;
    st->next_block = 0;
    return 0;
}

void a52_dynrng(a52_state_t *st, level_t (*call)(level_t, void *), void *data)
{
    st->dynrng = call ? 1 : 0;                           // parse.c:207-216
    st->dyn_call = call;
    st->dyn_data = data;
}

static int dropin_decode(a52_state_t *st)
{
    ac3mi_ctx *ctx = dropin_ctx();
    DROPIN_LOCK();
    ac3mi_decode_desc d;
    d.flags = st->req_flags;
    d.level = st->level_in;
    d.bias = st->bias;
    d.dynrng = st->dynrng;
    d.acmod = st->acmod;
    d.lfeon = st->lfeon;
    d.frame_bytes = st->frame_bytes;
    int n_out = 0, fl = 0;
    if (ac3mi_decode_planes(&d, &n_out, &fl) != AC3MI_OK) return -1;
    st->n_out = n_out;
    const int stride = (st->frame_bytes + 3) & ~3;
    if (ac3mi_memcpy_h2d(ctx, st->d_frame, st->frame, stride) != AC3MI_OK) return -1;
    ac3mi_decode_taps taps;
    memset(&taps, 0, sizeof taps);
    if (st->dyn_call) {
        // The callback is host code: a look-ahead pass on scratch copies of the carry-over state reports the range factor of
        // every dynamic-range word of the frame, the callback maps them in stream order (liba52 calls it from inside
        // a52_block, parse.c:593-594; here all calls of a frame happen at its first a52_block), and the frame is decoded
        // with the mapped values in their place.
        float words[12], mapped[12];
        uint16_t *d_lfsr2 = (uint16_t *)(st->d_scratch + ST_FLOATS);
        if (hipMemcpyAsync(st->d_scratch, st->d_delay, ST_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(d_lfsr2, st->d_lfsr, 2, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return -1;
        for (int i = 0; i < 12; i++) words[i] = __builtin_nanf("");
        if (ac3mi_memcpy_h2d(ctx, st->d_dyn, words, sizeof words) != AC3MI_OK) return -1;
        taps.d_dynrng_out = st->d_dyn;
        ac3mi_set_mix_state(ctx, st->d_scratch + ST_PENDING, (int32_t *)(st->d_scratch + ST_FLAGS));
        const int rc0 = ac3mi_decode_batch(ctx, &d, st->d_frame, stride, 1, 1, st->d_scratch, d_lfsr2, st->d_pcm, st->d_status, &taps);
        ac3mi_set_mix_state(ctx, nullptr, nullptr);
        if (rc0 != AC3MI_OK) return -1;
        if (ac3mi_memcpy_d2h(ctx, words, st->d_dyn, sizeof words) != AC3MI_OK) return -1;
        for (int i = 0; i < 12; i++) mapped[i] = words[i] == words[i] ? st->dyn_call(words[i], st->dyn_data) : words[i];
        if (ac3mi_memcpy_h2d(ctx, st->d_dyn + 12, mapped, sizeof mapped) != AC3MI_OK) return -1;
        taps.d_dynrng_out = nullptr;
        taps.d_dynrng_in = st->d_dyn + 12;
    }
    // liba52's own overlap bookkeeping around frames with surround level 0 (include/ac3mi.h, ac3mi_set_mix_state)
    ac3mi_set_mix_state(ctx, st->d_delay + ST_PENDING, (int32_t *)(st->d_delay + ST_FLAGS));
    const int rc1 = ac3mi_decode_batch(ctx, &d, st->d_frame, stride, 1, 1, st->d_delay, st->d_lfsr, st->d_pcm, st->d_status,
                                       st->dyn_call ? &taps : nullptr);
    ac3mi_set_mix_state(ctx, nullptr, nullptr);
    if (rc1 != AC3MI_OK) return -1;
    if (ac3mi_memcpy_d2h(ctx, st->pcm, st->d_pcm, (size_t)6 * n_out * 256 * sizeof(float)) != AC3MI_OK) return -1;
    if (ac3mi_memcpy_d2h(ctx, &st->status, st->d_status, 4) != AC3MI_OK) return -1;
    st->decoded = 1;
    return 0;
}

int a52_block(a52_state_t *st)
{
    if (!st->have_frame || st->next_block >= 6) return 1;
    if (!st->decoded && dropin_decode(st) != 0) return 1;
    const int b = st->next_block++;
    if (st->status & 0x100) return 1;
    if ((st->status >> b) & 1) return 1;
    memcpy(st->samples, st->pcm + (size_t)b * st->n_out * 256, (size_t)st->n_out * 256 * sizeof(float));
    return 0;
}

void a52_free(a52_state_t *st)
{
    if (!st) return;
    ac3mi_ctx *ctx = dropin_ctx();
    DROPIN_LOCK();
    if (ctx) {
        ac3mi_dev_free(ctx, st->d_frame);
        ac3mi_dev_free(ctx, st->d_delay);
        ac3mi_dev_free(ctx, st->d_pcm);
        ac3mi_dev_free(ctx, st->d_lfsr);
        ac3mi_dev_free(ctx, st->d_status);
        ac3mi_dev_free(ctx, st->d_dyn);
        ac3mi_dev_free(ctx, st->d_scratch);
    }
    free(st->samples);
    free(st);
}

// ---- encoder (ac3enc.h:6-7): one stream per process, like the reference's static context ------------

static struct {
    int ready, nch, frame_bytes, stride;
    ac3mi_encode_desc desc;
    int16_t *d_pcm, *d_last;
    int32_t *d_csnr;
    uint8_t *d_frame;
} g_enc;

int ac3mi_AC3_encode_init(int freq, int bitrate, int channels)
{
    ac3mi_encode_desc d = {freq, bitrate, channels};
    const int fb = ac3mi_encode_frame_bytes(&d);
    if (fb <= 0) return 0;
    ac3mi_ctx *ctx = dropin_ctx();
    if (!ctx) return 0;
    DROPIN_LOCK();
    if (g_enc.ready) {
        ac3mi_dev_free(ctx, g_enc.d_pcm);
        ac3mi_dev_free(ctx, g_enc.d_last);
        ac3mi_dev_free(ctx, g_enc.d_csnr);
        ac3mi_dev_free(ctx, g_enc.d_frame);
        g_enc.ready = 0;
    }
    g_enc.desc = d;
    g_enc.nch = channels;
    g_enc.frame_bytes = fb;
    g_enc.stride = (fb + 3) & ~3;
    g_enc.d_pcm = (int16_t *)ac3mi_dev_alloc(ctx, (size_t)1536 * channels * 2);
    g_enc.d_last = (int16_t *)ac3mi_dev_alloc(ctx, (size_t)channels * 256 * 2);
    g_enc.d_csnr = (int32_t *)ac3mi_dev_alloc(ctx, 4);
    g_enc.d_frame = (uint8_t *)ac3mi_dev_alloc(ctx, g_enc.stride);
    if (!g_enc.d_pcm || !g_enc.d_last || !g_enc.d_csnr || !g_enc.d_frame) return 0;
    const int32_t csnr0 = 40;                            // ac3enc.cpp:1092
    ac3mi_memset(ctx, g_enc.d_last, 0, (size_t)channels * 256 * 2);
    ac3mi_memcpy_h2d(ctx, g_enc.d_csnr, &csnr0, 4);
    g_enc.ready = 1;
    return fb;
}

int ac3mi_AC3_encode_frame(unsigned char *dst, short *samples, unsigned char *chmap)
{
    ac3mi_ctx *ctx = dropin_ctx();
    DROPIN_LOCK();
    if (!ctx || !g_enc.ready) return 0;
    if (ac3mi_memcpy_h2d(ctx, g_enc.d_pcm, samples, (size_t)1536 * g_enc.nch * 2) != AC3MI_OK) return 0;
    if (ac3mi_encode_batch(ctx, &g_enc.desc, g_enc.d_pcm, chmap, g_enc.d_last, g_enc.d_csnr, g_enc.d_frame, g_enc.stride,
                           1, 1, nullptr) != AC3MI_OK)
        return 0;
    if (ac3mi_memcpy_d2h(ctx, dst, g_enc.d_frame, g_enc.frame_bytes) != AC3MI_OK) return 0;
    return g_enc.frame_bytes;
}

// ---- converters (AC3ACM.cpp:87-90) ------------------------------------------------------------------

static void dropin_convert(const void *src, void *dst, int flags)
{
    ac3mi_ctx *ctx = dropin_ctx();
    int map[6];
    const int n_out = s16_channel_map(flags, map);
    if (!ctx || n_out <= 0) return;
    DROPIN_LOCK();
    static float *d_src;
    static int16_t *d_dst;
    if (!d_src) {
        d_src = (float *)ac3mi_dev_alloc(ctx, 6 * 256 * sizeof(float));
        d_dst = (int16_t *)ac3mi_dev_alloc(ctx, 6 * 256 * sizeof(int16_t));
    }
    if (!d_src || !d_dst) return;
    ac3mi_memcpy_h2d(ctx, d_src, src, (size_t)n_out * 256 * sizeof(float));
    ac3mi_convert_s16_batch(ctx, d_src, d_dst, flags, 1);
    ac3mi_memcpy_d2h(ctx, dst, d_dst, (size_t)n_out * 256 * sizeof(int16_t));
}

#define CV dropin_convert
// [MMX ok][source channels - 1][destination channels - 1]; 0 = unsupported, as AC3ASM.asm:40-112
ConvertProc MapTab[2][6][6] = {
    {{CV, CV, 0, 0, 0, 0}, {CV, CV, 0, 0, 0, 0}, {CV, CV, CV, 0, 0, 0}, {CV, CV, 0, CV, 0, 0}, {CV, CV, 0, 0, CV, 0}, {CV, CV, 0, 0, 0, CV}},
    {{CV, CV, 0, 0, 0, 0}, {CV, CV, 0, 0, 0, 0}, {CV, CV, CV, 0, 0, 0}, {CV, CV, 0, CV, 0, 0}, {CV, CV, 0, 0, CV, 0}, {CV, CV, 0, 0, 0, CV}},
};
#undef CV

bool IsMMX(void) { return true; }                        // as the x64 build (AC3ASM.asm:199-204)

// ---- secondary liba52 entry points (liba52/a52_internal.h:106-120) --------------------------------------
// Host pointers in and out like liba52's: one transform = one launch of the batched transform kernel on a single
// mono plane (the per-transform hook the reference's own tools use; not a fast path).

void a52_imdct_init(uint32_t mm_accel) { (void)mm_accel; (void)dropin_ctx(); }      // tables are built with the context

static void dropin_imdct(sample_t *data, sample_t *delay, sample_t bias, int short_block)
{
    ac3mi_ctx *ctx = dropin_ctx();
    if (!ctx || !data || !delay) return;
    DROPIN_LOCK();
    static float *d_buf;                                 // [256 coefficients | 256 PCM | 128 overlap | flag]
    if (!d_buf) d_buf = (float *)ac3mi_dev_alloc(ctx, (256 + 256 + 128 + 4) * sizeof(float));
    if (!d_buf) return;
    const uint8_t sw = short_block ? 1 : 0;
    XformLaunch L;
    if (build_mix_plan(1, 0, AC3MI_MONO, &L.plan) != AC3MI_OK) return;
    ac3mi_memcpy_h2d(ctx, d_buf, data, 256 * sizeof(float));
    ac3mi_memcpy_h2d(ctx, d_buf + 512, delay, 128 * sizeof(float));      // delay[128..255] is dead in liba52 (imdct.c:276-292)
    ac3mi_memcpy_h2d(ctx, d_buf + 640, &sw, 1);
    L.coef = d_buf;
    L.blksw = (const uint8_t *)(d_buf + 640);
    L.delay = d_buf + 512;
    L.pcm = d_buf + 256;
    L.n_streams = 1;
    L.frames = 1;
    L.blocks = 1;
    L.bias = bias;
    L.slot = nullptr;
    L.delay_stride = 128;
    if (hipSetDevice(ctx->device) != hipSuccess || launch_xform(ctx->tab, L, ctx->stream) != hipSuccess) return;
    ac3mi_memcpy_d2h(ctx, data, d_buf + 256, 256 * sizeof(float));
    ac3mi_memcpy_d2h(ctx, delay, d_buf + 512, 128 * sizeof(float));
}

void a52_imdct_512(sample_t *data, sample_t *delay, sample_t bias) { dropin_imdct(data, delay, bias, 0); }
void a52_imdct_256(sample_t *data, sample_t *delay, sample_t bias) { dropin_imdct(data, delay, bias, 1); }

// pure host arithmetic, shared with the kernels (a52_levels.h)
int a52_downmix_init(int input, int flags, level_t *level, level_t clev, level_t slev)
{
    return a52_downmix_init_hd(input, flags, level, clev, slev);
}
int a52_downmix_coeff(level_t *coeff, int acmod, int output, level_t level, level_t clev, level_t slev)
{
    return a52_downmix_coeff_hd(coeff, acmod, output, level, clev, slev);
}

}  // extern "C"

// the reference declares the encoder entry points with C++ linkage (ac3enc.h has no extern "C")
int AC3_encode_init(int freq, int bitrate, int channels) { return ac3mi_AC3_encode_init(freq, bitrate, channels); }
int AC3_encode_frame(unsigned char *dst, short *samples, unsigned char *chmap)
{
    return ac3mi_AC3_encode_frame(dst, samples, chmap);
}
