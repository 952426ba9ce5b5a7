// capi.hip — the C-ABI of libac3mi.so (include/ac3mi.h): context, device memory,
// table construction and the batched entry points.
#include "ac3mi_internal.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace ac3mi {

static std::string g_err;   // errors raised without a context

static const uint8_t kNfchans[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};

// ---------------------------------------------------------------------------
// tables (double precision on the host, rounded once to float)

static double bessel_i0(double x)
{
    // L52/imdct.c:347-356: 100-term series in Horner form
    double b = 1;
    for (int i = 100; i > 0; i--) b = b * x / (i * i) + 1;
    return b;
}

void build_host_tables(float *window, float2 *tw_long, float2 *tw_short)
{
    const double pi = 3.14159265358979323846;
    // KBD window, alpha = 5 (L52/imdct.c:364-372)
    double acc = 0, cum[256];
    for (int i = 0; i < 256; i++) {
        acc += bessel_i0(i * (256 - i) * (5 * pi / 256) * (5 * pi / 256));
        cum[i] = acc;
    }
    acc += 1;
    for (int i = 0; i < 256; i++) window[i] = (float)sqrt(cum[i] / acc);

    // merged lane twiddles (xform_core.h): lane n2, register k1
    //   long : (-1)^n2 e^{-j pi (n2+63.75)/256} . e^{-j 2 pi n2 k1/128} . e^{-j pi (k1+.5)/256}
    //   short: e^{-j pi (n-.25)/128} . e^{-j 2 pi n k1/64} . e^{-j pi (k1+.5)/128},  n = lane & 3
    for (int l = 0; l < 8; l++)
        for (int k = 0; k < 16; k++) {
            double a = -pi * (l + 63.75) / 256 - 2 * pi * l * k / 128 - pi * (k + 0.5) / 256 + (l & 1 ? pi : 0);
            tw_long[l * 16 + k] = make_float2((float)cos(a), (float)sin(a));
            int n = l & 3;
            double b = -pi * (n - 0.25) / 128 - 2 * pi * n * k / 64 - pi * (k + 0.5) / 128;
            tw_short[l * 16 + k] = make_float2((float)cos(b), (float)sin(b));
        }
}

// ---------------------------------------------------------------------------
// a52_downmix() as a plane-mixing matrix (L52/downmix.c:480-619) and the set of
// outputs a52_downmix_init() can grant (L52/downmix.c:37-67)

int build_mix_plan(int acmod, int lfeon, int output, MixPlan *plan)
{
    // row = requested output, column = coded acmod
    static const uint8_t grant[11][8] = {
        {0, 10, 2, 2, 2, 2, 2, 2},  {1, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 2, 2, 2, 2, 2}, {0, 10, 2, 3, 2, 3, 2, 3},
        {0, 10, 2, 2, 4, 4, 4, 4},  {0, 10, 2, 2, 4, 5, 4, 5}, {0, 10, 2, 3, 6, 6, 6, 6}, {0, 10, 2, 3, 6, 7, 6, 7},
        {8, 1, 1, 1, 1, 1, 1, 1},   {9, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 10, 10, 10, 10, 10}};
    if (acmod < 0 || acmod > 7) return AC3MI_ERR_ARG;
    const int out = output & AC3MI_CHANNEL_MASK;
    if (out > AC3MI_DOLBY) return AC3MI_ERR_ARG;
    if (grant[out][acmod] != out) return AC3MI_ERR_ARG;
    if ((output & AC3MI_LFE) && !lfeon) return AC3MI_ERR_ARG;

    memset(plan, 0, sizeof *plan);
    plan->nfchans = kNfchans[acmod];
    plan->in_lfe = lfeon ? 1 : 0;
    plan->n_in = plan->nfchans + plan->in_lfe;
    const int out_lfe = (output & AC3MI_LFE) ? 1 : 0;
    const int nfo = kNfchans[out];
    plan->n_out = nfo + out_lfe;

    int8_t m[5][5];
    memset(m, 0, sizeof m);
    auto id = [&](int n) { for (int i = 0; i < n; i++) m[i][i] = 1; };
    auto fold_centre = [&]() { m[0][0] = 1; m[0][1] = 1; m[1][2] = 1; m[1][1] = 1; };   // mix3to2
    const int A = acmod;

    switch (out) {
    case AC3MI_CHANNEL:  case AC3MI_CHANNEL1:
        id(nfo);                                                    // (0,0) identity; CHANNEL1 keeps plane 0
        break;
    case AC3MI_CHANNEL2:
        m[0][1] = 1;                                                // downmix.c:485-487
        break;
    case AC3MI_MONO:
        for (int c = 0; c < plan->nfchans; c++) m[0][c] = 1;        // mix2to1..mix5to1
        break;
    case AC3MI_STEREO:
        switch (A) {
        case 2: id(2); break;
        case 3: fold_centre(); break;
        case 4: m[0][0] = 1; m[1][1] = 1; m[0][2] = 1; m[1][2] = 1; break;                     // mix21to2
        case 5: fold_centre(); m[0][3] = 1; m[1][3] = 1; break;                                // mix31to2
        case 6: m[0][0] = 1; m[0][2] = 1; m[1][1] = 1; m[1][3] = 1; break;                     // 2x mix2to1
        case 7: fold_centre(); m[0][3] = 1; m[1][4] = 1; break;                                // mix32to2
        default: return AC3MI_ERR_ARG;
        }
        break;
    case AC3MI_DOLBY:
        switch (A) {
        case 1: m[0][0] = 1; m[1][0] = 1; break;                                               // memcpy
        case 2: id(2); break;
        case 3: fold_centre(); break;
        case 4: m[0][0] = 1; m[1][1] = 1; m[0][2] = -1; m[1][2] = 1; break;                    // mix21toS
        case 5: fold_centre(); m[0][3] = -1; m[1][3] = 1; break;                               // mix31toS
        case 6: m[0][0] = 1; m[1][1] = 1; m[0][2] = m[0][3] = -1; m[1][2] = m[1][3] = 1; break; // mix22toS
        case 7: fold_centre(); m[0][3] = m[0][4] = -1; m[1][3] = m[1][4] = 1; break;           // mix32toS
        default: return AC3MI_ERR_ARG;
        }
        break;
    case AC3MI_3F:
        id(3);
        if (A == 5) { m[0][3] = 1; m[2][3] = 1; }                                              // mix21to2
        if (A == 7) { m[0][3] = 1; m[2][4] = 1; }
        break;
    case AC3MI_2F1R:
        if (A == 4) id(3);
        else if (A == 5) { fold_centre(); m[2][3] = 1; }
        else if (A == 6) { id(2); m[2][2] = 1; m[2][3] = 1; }
        else if (A == 7) { fold_centre(); m[2][3] = 1; m[2][4] = 1; }                          // move2to1
        else return AC3MI_ERR_ARG;
        break;
    case AC3MI_3F1R:
        id(4);
        if (A == 7) m[3][4] = 1;
        break;
    case AC3MI_2F2R:
        if (A == 6) id(4);
        else if (A == 4) { id(3); m[3][2] = 1; }
        else if (A == 5) { fold_centre(); m[2][3] = 1; m[3][3] = 1; }
        else if (A == 7) { fold_centre(); m[2][3] = 1; m[3][4] = 1; }
        else return AC3MI_ERR_ARG;
        break;
    case AC3MI_3F2R:
        if (A == 7) id(5);
        else if (A == 5) { id(4); m[4][3] = 1; }
        else return AC3MI_ERR_ARG;
        break;
    }
    for (int o = 0; o < nfo; o++)
        for (int c = 0; c < plan->nfchans; c++) plan->mix[o + out_lfe][c + plan->in_lfe] = m[o][c];
    if (out_lfe) plan->mix[0][0] = 1;
    return AC3MI_OK;
}

}  // namespace ac3mi

using namespace ac3mi;

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            std::string msg_ = std::string(#call) + ": " + hipGetErrorString(e_);      \
            if (ctx) (ctx)->err = msg_; else g_err = msg_;                             \
            return AC3MI_ERR_HIP;                                                      \
        }                                                                              \
    } while (0)

extern "C" {

int ac3mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ac3mi_last_error(const ac3mi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static int ctx_init(ac3mi_ctx *ctx)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    std::vector<float> win(256);
    std::vector<float2> twl(128), tws(128);
    build_host_tables(win.data(), twl.data(), tws.data());
    HIPCHK(ctx, hipMalloc(&ctx->tab.window, 256 * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&ctx->tab.tw_long, 128 * sizeof(float2)));
    HIPCHK(ctx, hipMalloc(&ctx->tab.tw_short, 128 * sizeof(float2)));
    HIPCHK(ctx, hipMemcpy(ctx->tab.window, win.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->tab.tw_long, twl.data(), 128 * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->tab.tw_short, tws.data(), 128 * sizeof(float2), hipMemcpyHostToDevice));
    return AC3MI_OK;
}

ac3mi_ctx *ac3mi_create(int device)
{
    int n = ac3mi_device_count();
    if (n <= 0) {
        g_err = "ac3mi_create: no HIP device visible (libac3mi has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= n) {
        g_err = "ac3mi_create: device index out of range";
        return nullptr;
    }
    ac3mi_ctx *ctx = new ac3mi_ctx();
    ctx->device = device;
    ctx->stream = nullptr;
    ctx->tab = DeviceTables{nullptr, nullptr, nullptr};
    if (ctx_init(ctx) != AC3MI_OK) {
        g_err = ctx->err;
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void ac3mi_destroy(ac3mi_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->tab.window);
    (void)hipFree(ctx->tab.tw_long);
    (void)hipFree(ctx->tab.tw_short);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *ac3mi_dev_alloc(ac3mi_ctx *ctx, size_t bytes)
{
    void *p = nullptr;
    if (!ctx) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return nullptr;
    }
    return p;
}

void ac3mi_dev_free(ac3mi_ctx *ctx, void *d_ptr)
{
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}

int ac3mi_memcpy_h2d(ac3mi_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_memcpy_d2h(ac3mi_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_memset(ac3mi_ctx *ctx, void *d_dst, int byte, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemsetAsync(d_dst, byte, bytes, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_sync(ac3mi_ctx *ctx)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_timer_start(ac3mi_ctx *ctx)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_timer_stop(ac3mi_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return AC3MI_OK;
}

int ac3mi_xform_planes(const ac3mi_xform_desc *desc, int *n_in, int *n_out)
{
    MixPlan plan;
    if (!desc) return AC3MI_ERR_ARG;
    int r = build_mix_plan(desc->acmod, desc->lfeon, desc->output, &plan);
    if (r != AC3MI_OK) return r;
    if (n_in) *n_in = plan.n_in;
    if (n_out) *n_out = plan.n_out;
    return AC3MI_OK;
}

int ac3mi_imdct_batch(ac3mi_ctx *ctx, const ac3mi_xform_desc *desc, const float *d_coeffs,
                      const uint8_t *d_blksw, float *d_delay, float *d_pcm, int n_streams,
                      int frames_per_stream)
{
    if (!ctx) return AC3MI_ERR_ARG;
    if (!desc || !d_coeffs || !d_delay || !d_pcm || n_streams < 0 || frames_per_stream < 0) {
        ctx->err = "ac3mi_imdct_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    XformLaunch L;
    int r = build_mix_plan(desc->acmod, desc->lfeon, desc->output, &L.plan);
    if (r != AC3MI_OK) {
        ctx->err = "ac3mi_imdct_batch: output configuration not reachable from acmod (a52_downmix_init)";
        return r;
    }
    if ((long long)n_streams * L.plan.n_out > 0x7fffffffLL) {
        ctx->err = "ac3mi_imdct_batch: too many chains for one launch";
        return AC3MI_ERR_ARG;
    }
    L.coef = d_coeffs;
    L.blksw = d_blksw;
    L.delay = d_delay;
    L.pcm = d_pcm;
    L.n_streams = n_streams;
    L.frames = frames_per_stream;
    L.bias = desc->bias;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_xform(ctx->tab, L, ctx->stream));
    return AC3MI_OK;
}

}  // extern "C"
