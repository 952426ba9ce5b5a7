// stream.hip — the ACM driver's stream messages (open / size / convert / close) as plain host code
// over the batched engine.  See include/ac3mi_stream.h for the contract and the reference lines.
//
// Every stream is a small resumable state machine that follows stream_convert_ac3 /
// stream_convert_pcm (src/AC3ACM.cpp:1430-1628, 1665-1798) statement by statement and stops where the
// reference calls a52_frame or AC3_encode_frame.  ac3mi_stream_convert_many runs all machines to their
// next stop, hands the frames that became ready to ONE ac3mi_decode_batch / ac3mi_encode_batch per
// configuration (state slots select each stream's carry-over state), resumes, and repeats until every
// header is finished.  No kernels of its own.
#include "ac3mi_internal.h"
#include "../../include/ac3mi_stream.h"
#include "a52_levels.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string.h>
#include <thread>
#include <tuple>
#include <vector>

namespace {

// src/AC3ACM.cpp:128-149
const short k_framesizes[19][4] = {
    {96, 69, 64, 32},      {120, 87, 80, 40},     {144, 104, 96, 48},    {168, 121, 112, 56},   {192, 139, 128, 64},
    {240, 174, 160, 80},   {288, 208, 192, 96},   {336, 243, 224, 112},  {384, 278, 256, 128},  {480, 348, 320, 160},
    {576, 417, 384, 192},  {672, 487, 448, 224},  {768, 557, 512, 256},  {960, 696, 640, 320},  {1152, 835, 768, 384},
    {1344, 975, 896, 448}, {1536, 1114, 1024, 512}, {1728, 1253, 1152, 576}, {1920, 1393, 1280, 640}};
const uint32_t k_channel_masks[6] = {0x4, 0x3, 0x7, 0x33, 0x37, 0x3f};     // src/AC3ACM.cpp:156-163

constexpr int FRAME_STRIDE = 3840;
constexpr int PCM_FRAME_BYTES = 1536 * 6 * 2;

bool valid_rate(uint32_t sps)
{
    return sps == 48000 || sps == 44100 || sps == 32000 || sps == 24000 || sps == 22050 || sps == 16000 || sps == 12000 ||
           sps == 11025 || sps == 8000;
}

// IsValidPCM / IsValidPCMEX, src/AC3ACM.cpp:207-277
bool is_valid_pcm(const ac3mi_wavefmt *f, uint32_t driver_flags)
{
    if (!f) return false;
    const bool ex = f->format_tag == AC3MI_WAVE_FORMAT_EXTENSIBLE;
    if (f->format_tag != AC3MI_WAVE_FORMAT_PCM && !(ex && !(driver_flags & AC3MI_ACM_NOEXTENSIBLE))) return false;
    if (f->channels < 1 || f->channels > 6) return false;
    if (f->bits_per_sample != 16) return false;
    if (f->block_align != f->channels * 2u) return false;
    if (f->avg_bytes_per_sec != (uint32_t)f->block_align * f->samples_per_sec) return false;
    if (!valid_rate(f->samples_per_sec)) return false;
    if (ex && f->channel_mask != k_channel_masks[f->channels - 1]) return false;
    return true;
}

// IsValidAC3, src/AC3ACM.cpp:315-352 (the EXTENSIBLE branch can never succeed, :303-304)
bool is_valid_ac3(const ac3mi_wavefmt *f)
{
    if (!f || f->format_tag != AC3MI_WAVE_FORMAT_AC3) return false;
    if (f->channels < 1 || f->channels > 6) return false;
    if (!valid_rate(f->samples_per_sec)) return false;
    if (f->avg_bytes_per_sec < 3000 || f->avg_bytes_per_sec > 81000u) return false;
    if (f->block_align == 0) return false;
    return true;
}

// create_channel_map, src/AC3ACM.cpp:1631-1662
void channel_map(int channels, uint8_t *dst)
{
    for (int i = 0; i < 8; i++) dst[i] = (uint8_t)i;
    if (channels == 3 || channels == 5) { dst[1] = 2; dst[2] = 1; }
    if (channels == 6) { dst[0] = 0; dst[1] = 2; dst[2] = 1; dst[3] = 4; dst[4] = 5; dst[5] = 3; }
}

// bytes of one AC-3 frame at `kbps` and `rate`, as stream_size computes it (src/AC3ACM.cpp:2187-2207); 0 = unknown
int frame_bytes_for(int kbps, uint32_t rate)
{
    for (int i = 0; i < 19; i++)
        if (kbps == k_framesizes[i][3]) {
            if (rate == 32000) return k_framesizes[i][0] * 2;
            if (rate == 44100) return k_framesizes[i][1] * 2 + 2;
            if (rate == 48000) return k_framesizes[i][2] * 2;
            return kbps;        // the reference leaves `len` as the bit rate for other sample rates
        }
    return 0;
}

enum Stop { STOP_DONE = 0, STOP_DECODE = 1, STOP_ENCODE = 2 };
constexpr int MAX_CHUNKS = 8;
// AC3MI_STREAM_TRACE=1: wall-clock split of ac3mi_stream_convert_many on stderr (per call), a measurement aid
struct Trace {
    bool on = getenv("AC3MI_STREAM_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) { if (on) { auto t = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::milli>(t - t0).count(); t0 = t; } }
    void report() {
        if (!on) return;
        fprintf(stderr, "ac3mi_stream_convert_many: machines %.2f + %.2f  grouping %.2f  staging %.2f  issue %.2f ms\n", acc[0], acc[5], acc[1], acc[2], acc[3]);
        for (double &a : acc) a = 0;
    }
};
Trace g_trace;

// a round's batch crosses PCIe in chunks of about 1 024 streams (AC3MI_STREAM_CHUNKS overrides: measurement aid)
int chunks_for(int k)
{
    static const int forced = getenv("AC3MI_STREAM_CHUNKS") ? atoi(getenv("AC3MI_STREAM_CHUNKS")) : 0;
    int n = forced > 0 ? forced : k / 1024;
    if (n > k) n = k;
    if (n < 1) n = 1;
    return n > MAX_CHUNKS ? MAX_CHUNKS : n;
}

// The per-stream host work (buffering state machines, staging copies) is independent per stream: a pool of worker
// threads that lives as long as the library, woken per call (thread creation per call cost more than the work of a
// round of a few thousand streams).
class Workers {
public:
    static Workers &get() { static Workers w; return w; }
    int threads() const { return (int)th_.size() + 1; }
    // share(t, nt): worker t of nt takes the items of its share; the caller is worker 0
    void run(const std::function<void(int, int)> &share)
    {
        std::lock_guard<std::mutex> one_at_a_time(run_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &share;
            pending_ = (int)th_.size();
            generation_++;
        }
        cv_.notify_all();
        share(0, threads());
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    Workers()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int nt = (int)(hw ? hw : 1);
        if (nt > 16) nt = 16;
        for (int t = 1; t < nt; t++) th_.emplace_back([this, t] { loop(t); });
    }
    ~Workers()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; generation_++; }
        cv_.notify_all();
        for (auto &x : th_) x.join();
    }
    void loop(int t)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int, int)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
            }
            if (job) (*job)(t, threads());
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *job_ = nullptr;
    int pending_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

template <class F> void parallel_for(int n, F f)
{
    if (n < 512 || Workers::get().threads() < 2) { for (int i = 0; i < n; i++) f(i); return; }
    // shares are block-cyclic (16 items a block): every worker walks the whole index range in order, so when the items
    // become ready in index order (the chunks of decode_group) all workers start early and finish together
    Workers::get().run([&](int t, int nt) {
        for (int lo = t * 16; lo < n; lo += nt * 16)
            for (int i = lo, hi = lo + 16 < n ? lo + 16 : n; i < hi; i++) f(i);
    });
}

}  // namespace

struct ac3mi_pool {
    ac3mi_ctx *ctx;
    int capacity;
    std::vector<int> free_slots;
    std::vector<char> dirty;        // slot state differs from a fresh a52_init / AC3_encode_init
    // per-slot carry-over state (device)
    float *d_delay;         // [cap][6][128]
    float *d_mixp;          // [cap][6][128]  ac3mi_set_mix_state: liba52's overlap bookkeeping around surround level 0
    int32_t *d_mixf;        // [cap][6]
    uint16_t *d_lfsr;       // [cap]
    int16_t *d_last;        // [cap][6][256]
    int32_t *d_csnr;        // [cap]
    // per-round staging (device / pinned host), one entry per stream taking part in the round
    int32_t *d_slots, *h_slots;
    uint8_t *d_frames, *h_frames;       // [cap][FRAME_STRIDE]
    int16_t *d_s16, *h_s16;             // [cap][6][256][6]  (decode out)  /  [cap][1536][6] (encode in)
    uint32_t *d_status, *h_status;      // [cap]
    // decode_group's chunk pipeline: front end + transform of chunk c done / its samples on the host
    hipEvent_t ev_kernel[MAX_CHUNKS], ev_host[MAX_CHUNKS], ev_join;
    std::atomic<int> host_seen[MAX_CHUNKS];
};

struct ac3mi_stream {
    ac3mi_pool *pool;
    int slot;
    bool decode;                    // AC-3 -> PCM, else PCM -> AC-3
    ac3mi_wavefmt src, dst;
    uint32_t driver_flags;
    int framelen;                   // msd->framelen
    // MyStreamData, src/AC3ACM.cpp:92-102
    uint8_t buf[1536 * 6 * 2 + 64]; // decode: 4096+16 used; encode: one frame of PCM
    uint8_t frame[4096 + 64];       // encode: the frame last produced
    uint8_t *bufptr, *bufend;
    int blocks;                     // decode: blocks not yet handed out; encode: bytes not yet handed out
    int flags;                      // msd->flags
    int16_t pcm[6][256 * 6];        // decode: blocks of the current frame that are still owed, interleaved for dst.channels
    const uint8_t *pcm_cur;         // where the current frame's six blocks are: the pool's staging right after a batch, else pcm
    hipEvent_t pcm_ready;           // set by decode_group: the copy that fills pcm_cur / status_src is still in flight
    std::atomic<int> *pcm_seen;     // ... and the flag the first stream to see that event finished raises for the rest of its chunk
    const uint32_t *status_src;
    // per-call locals of the reference's functions, kept across stops
    ac3mi_stream_header *hdr;
    const uint8_t *src_p;
    uint8_t *dst_p;
    long src_left, dst_left;
    int phase;                      // 0 = entry, 1 = main loop, 2 = resume after a batch, 3 = finished
    unsigned long long round_tag;   // the ac3mi_stream_convert_many call that last took this stream (a stream may appear once per call)
    // hand-over to / from the batch step
    int req_flags, acmod, lfeon, frame_bytes, granted, fs_next;
    uint32_t status;
    ac3mi::EncConfig enc_cfg;
    int enc_rate, enc_bitrate;
};

namespace {

void emit_block(ac3mi_stream *st, int index)
{
    const int sr = 512 * st->dst.channels;
    memcpy(st->dst_p, st->pcm_cur + (size_t)index * sr, (size_t)sr);
    st->hdr->dst_used += (uint32_t)sr;
    st->dst_p += sr;
}

// stream_convert_ac3, src/AC3ACM.cpp:1430-1628.  Returns STOP_DECODE with the frame at st->bufptr when the
// reference would call a52_frame; call again after the batch filled st->pcm / st->status.
Stop run_decode(ac3mi_stream *st)
{
    ac3mi_stream_header *sh = st->hdr;
    const int nch = st->dst.channels;
    int fs, sr;
    if (st->phase == 0) {
        sh->src_used = sh->dst_used = 0;
        st->src_p = sh->src;
        st->dst_p = sh->dst;
        st->src_left = (long)sh->src_len;
        st->dst_left = (long)sh->dst_len;
        st->phase = 1;
        if (sh->flags & AC3MI_STREAMCONVERTF_START) {
            st->bufptr = st->bufend = st->buf;
            st->blocks = 0;
        } else if (st->blocks > 0) {
            sr = 512 * nch;                                         // blocks left over from the last call
            do {
                if ((st->dst_left -= sr) < 0) { st->phase = 3; return STOP_DONE; }
                emit_block(st, 6 - st->blocks);
            } while (--st->blocks);
        }
    }
    if (st->phase == 2) {
        // back from a52_frame: hand out the blocks (:1574-1586)
        if (st->pcm_ready) {
            // the batch's samples come over in chunks; this stream's chunk may still be on its way
            if (!st->pcm_seen->load(std::memory_order_acquire)) {
                if (hipEventSynchronize(st->pcm_ready) != hipSuccess) { memset(st->pcm, 0, sizeof st->pcm); st->pcm_cur = (const uint8_t *)st->pcm; }
                else st->pcm_seen->store(1, std::memory_order_release);
            }
            st->pcm_ready = nullptr;
            st->status = *st->status_src;
            st->granted = (int)((st->status >> 16) & 0xff);
        }
        st->phase = 1;
        st->bufptr = st->bufend = st->buf;
        sr = 512 * nch;
        fs = ((long long)sh->src_len * 2 - sh->src_used) < (long long)st->fs_next ? 1 : 0;
        for (st->blocks = 6; st->blocks > fs; --st->blocks) {
            if ((st->dst_left -= sr) < 0) break;
            emit_block(st, 6 - st->blocks);
        }
        // blocks still owed move out of the pool's staging (it is reused by the next round)
        for (int b = 6 - st->blocks; b < 6 && st->pcm_cur != (const uint8_t *)st->pcm; b++)
            memcpy((uint8_t *)st->pcm + (size_t)b * sr, st->pcm_cur + (size_t)b * sr, (size_t)sr);
        st->pcm_cur = (const uint8_t *)st->pcm;
        if (st->dst_left < 0) { st->phase = 3; return STOP_DONE; }
        fs = 128;
        goto refill;
    }
    if (st->phase == 3) return STOP_DONE;

    for (;;) {
        fs = 128 - (int)(st->bufend - st->buf);
        sr = (int)(st->bufend - st->bufptr);
        if (sr >= 8) {
            for (;;) {
                int a52flags, sr2, br;
                if ((fs = ac3mi_syncinfo(st->bufptr, &a52flags, &sr2, &br)) != 0) {
                    if (sr < fs) { fs -= sr; break; }               // need fs more bytes of data
                    st->acmod = a52flags & 7;
                    if ((a52flags & AC3MI_CHANNEL_MASK) == AC3MI_DOLBY) st->acmod = 2;
                    st->lfeon = (a52flags & AC3MI_LFE) ? 1 : 0;
                    st->frame_bytes = fs;
                    if (st->dst.channels != st->src.channels) {     // :1520-1553
                        if (st->dst.channels < st->src.channels) {
                            if (st->dst.channels == 1) a52flags = AC3MI_MONO;
                            else if ((a52flags & AC3MI_CHANNEL_MASK) != AC3MI_DOLBY)
                                a52flags = (st->driver_flags & AC3MI_ACM_DOLBYSURROUND) ? AC3MI_DOLBY : AC3MI_STEREO;
                        } else {
                            a52flags = AC3MI_STEREO;
                        }
                    }
                    st->flags = a52flags;
                    st->req_flags = a52flags | AC3MI_ADJUST_LEVEL;
                    st->fs_next = fs;
                    st->phase = 2;
                    return STOP_DECODE;
                }
                ++st->bufptr;
                --sr;
                if (sr < 8) {
                    for (sr = 0; sr < 8; sr++) st->buf[sr] = st->bufptr[sr];
                    st->bufptr = st->buf;
                    st->bufend = st->buf + 8;
                    fs = 120;
                    break;
                }
            }
        }
    refill:
        if (st->src_left <= 0) break;
        if (fs > st->src_left) fs = (int)st->src_left;
        memcpy(st->bufend, st->src_p, (size_t)fs);
        st->bufend += fs;
        sh->src_used += (uint32_t)fs;
        st->src_left -= fs;
        st->src_p += fs;
    }
    st->phase = 3;
    return STOP_DONE;
}

// stream_convert_pcm, src/AC3ACM.cpp:1665-1798.  STOP_ENCODE when st->buf holds 1536 samples per channel.
Stop run_encode(ac3mi_stream *st)
{
    ac3mi_stream_header *sh = st->hdr;
    const int needed = 1536 * st->src.channels * 2;
    if (st->phase == 0) {
        sh->src_used = sh->dst_used = 0;
        st->src_p = sh->src;
        st->dst_p = sh->dst;
        st->src_left = (long)sh->src_len;
        st->dst_left = (long)sh->dst_len;
        st->phase = 1;
        if (sh->flags & AC3MI_STREAMCONVERTF_START) {
            st->bufptr = st->buf;
            st->bufend = st->frame;
            st->blocks = 0;
        } else if (st->blocks > 0) {
            int tc = st->blocks;
            if (tc > st->dst_left) tc = (int)st->dst_left;
            if (tc > 0) {
                memcpy(st->dst_p, st->bufend, (size_t)tc);
                sh->dst_used += (uint32_t)tc;
                st->blocks -= tc;
                st->bufend += tc;
                st->dst_left -= tc;
                st->dst_p += tc;
            }
            if (st->dst_left <= 0) { st->phase = 3; return STOP_DONE; }
        }
    }
    if (st->phase == 3) return STOP_DONE;
    if (st->phase == 2) {
        // back from AC3_encode_frame (:1762-1786): st->frame holds st->frame_bytes bytes
        st->phase = 1;
        int tc = st->frame_bytes;
        st->bufptr = st->buf;
        st->bufend = st->frame;
        st->blocks = tc;
        if (tc > st->dst_left) tc = (int)st->dst_left;
        if (tc > 0) {
            memcpy(st->dst_p, st->frame, (size_t)tc);
            sh->dst_used += (uint32_t)tc;
            st->blocks -= tc;
            st->bufend += tc;
            st->dst_left -= tc;
            st->dst_p += tc;
        }
        if (st->dst_left <= 0) { st->phase = 3; return STOP_DONE; }
    }
    while (st->src_left > 0) {
        int fs = (int)(st->bufptr - st->buf);
        if (fs < needed) {
            int tc = needed - fs;
            if (tc > st->src_left) tc = (int)st->src_left;
            memcpy(st->bufptr, st->src_p, (size_t)tc);
            sh->src_used += (uint32_t)tc;
            st->bufptr += tc;
            st->src_left -= tc;
            st->src_p += tc;
            fs += tc;
        }
        if (fs >= needed) {
            st->phase = 2;
            return STOP_ENCODE;
        }
    }
    st->phase = 3;
    return STOP_DONE;
}

int fail(ac3mi_pool *p, int code, const char *what)
{
    p->ctx->err = std::string("ac3mi_stream: ") + what + (p->ctx->err.empty() ? "" : (": " + p->ctx->err));
    return code;
}

// one batched decode for the streams in `group` (same coded configuration and request)
int decode_group(ac3mi_pool *p, const std::vector<ac3mi_stream *> &group, int base)
{
    // staging entries [base, base + k): results stay there until the state machines have taken them
    ac3mi_ctx *ctx = p->ctx;
    int32_t *d_slots = p->d_slots + base, *h_slots = p->h_slots + base;
    uint8_t *d_frames = p->d_frames + (size_t)base * FRAME_STRIDE, *h_frames = p->h_frames + (size_t)base * FRAME_STRIDE;
    int16_t *d_s16 = (int16_t *)((uint8_t *)p->d_s16 + (size_t)base * PCM_FRAME_BYTES), *h_s16 = (int16_t *)((uint8_t *)p->h_s16 + (size_t)base * PCM_FRAME_BYTES);
    uint32_t *d_status = p->d_status + base, *h_status = p->h_status + base;
    (void)d_status; (void)h_status;
    const int k = (int)group.size();
    const ac3mi_stream *s0 = group[0];
    ac3mi_decode_desc d;
    d.flags = s0->req_flags;
    d.level = 1.0f;
    d.bias = 384.0f;                                                // :1568
    d.dynrng = (s0->driver_flags & AC3MI_ACM_DYNAMICRANGE) ? 1 : 0; // :2040-2043
    d.acmod = s0->acmod;
    d.lfeon = s0->lfeon;
    d.frame_bytes = s0->frame_bytes;
    int n_out = 0, granted = 0;
    const bool grantable = ac3mi_decode_planes(&d, &n_out, &granted) == AC3MI_OK && n_out == s0->dst.channels;
    if (!grantable) {
        // liba52 would hand fewer (or other) planes than the destination format has: silence (see header)
        for (ac3mi_stream *st : group) { memset(st->pcm, 0, sizeof st->pcm); st->pcm_cur = (const uint8_t *)st->pcm; }
        // the frame still advances the stream's dither / overlap state in the reference; it cannot be reproduced
        return AC3MI_MMSYSERR_NOERROR;
    }
    // frames packed at their own size (rounded up to 4): what crosses PCIe is what the streams sent.  The staging regions of
    // the groups of a round do not overlap: a group of k streams owns k * FRAME_STRIDE bytes from its base.
    const int fstride = (s0->frame_bytes + 3) & ~3;
    parallel_for(k, [&](int i) {
        h_slots[i] = group[i]->slot;
        memcpy(h_frames + (size_t)i * fstride, group[i]->bufptr, (size_t)s0->frame_bytes);
    });
    g_trace.lap(2);
    // the batch runs in chunks of whole streams on the context's stream (frames in, kernels) while a second stream brings
    // each finished chunk's samples over: 18 KB per frame across PCIe is the longest
    // leg of a round, and the state machines take a chunk's blocks (run_decode waits on ev_host) while the next is on its way
    const size_t blk = (size_t)256 * n_out * 2;
    if (hipMemcpyAsync(d_slots, h_slots, (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
    const int n_chunks = chunks_for(k);
    for (int c = 0; c < n_chunks; c++) {
        const int lo = (int)((long long)k * c / n_chunks), hi = (int)((long long)k * (c + 1) / n_chunks), kc = hi - lo;
        p->host_seen[c].store(0, std::memory_order_relaxed);
        if (hipMemcpyAsync(d_frames + (size_t)lo * fstride, h_frames + (size_t)lo * fstride, (size_t)kc * fstride, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            (void)hipDeviceSynchronize();
            return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
        }
        ac3mi_set_state_slots(ctx, d_slots + lo);
        ac3mi_set_mix_state(ctx, p->d_mixp, p->d_mixf);
        const int rc = ac3mi_decode_s16_batch(ctx, &d, d_frames + (size_t)lo * fstride, fstride, kc, 1, p->d_delay, p->d_lfsr,
                                              (int16_t *)((uint8_t *)d_s16 + (size_t)lo * 6 * blk), d_status + lo);
        ac3mi_set_mix_state(ctx, NULL, NULL);
        ac3mi_set_state_slots(ctx, NULL);
        if (rc != AC3MI_OK) { (void)hipDeviceSynchronize(); return fail(p, AC3MI_MMSYSERR_NOMEM, "decode batch"); }
        if (hipEventRecord(p->ev_kernel[c], ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream2, p->ev_kernel[c], 0) != hipSuccess ||
            hipMemcpyAsync((uint8_t *)h_s16 + (size_t)lo * 6 * blk, (uint8_t *)d_s16 + (size_t)lo * 6 * blk, (size_t)kc * 6 * blk, hipMemcpyDeviceToHost, ctx->stream2) != hipSuccess ||
            hipMemcpyAsync(h_status + lo, d_status + lo, (size_t)kc * 4, hipMemcpyDeviceToHost, ctx->stream2) != hipSuccess ||
            hipEventRecord(p->ev_host[c], ctx->stream2) != hipSuccess) {
            (void)hipDeviceSynchronize();
            return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
        }
        for (int i = lo; i < hi; i++) {
            ac3mi_stream *st = group[i];
            st->pcm_cur = (const uint8_t *)h_s16 + (size_t)i * 6 * blk;      // handed out (or saved) by run_decode before the next batch
            st->status_src = h_status + i;
            st->pcm_ready = p->ev_host[c];
            st->pcm_seen = &p->host_seen[c];
        }
    }
    // the next call on this context (and the next round's staging writes) must find the second stream idle as well
    if (hipEventRecord(p->ev_join, ctx->stream2) != hipSuccess || hipStreamWaitEvent(ctx->stream, p->ev_join, 0) != hipSuccess)
        return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
    g_trace.lap(3);
    return AC3MI_MMSYSERR_NOERROR;
}

int encode_group(ac3mi_pool *p, const std::vector<ac3mi_stream *> &group, int base)
{
    ac3mi_ctx *ctx = p->ctx;
    int32_t *d_slots = p->d_slots + base, *h_slots = p->h_slots + base;
    uint8_t *d_frames = p->d_frames + (size_t)base * FRAME_STRIDE, *h_frames = p->h_frames + (size_t)base * FRAME_STRIDE;
    int16_t *d_s16 = (int16_t *)((uint8_t *)p->d_s16 + (size_t)base * PCM_FRAME_BYTES), *h_s16 = (int16_t *)((uint8_t *)p->h_s16 + (size_t)base * PCM_FRAME_BYTES);
    uint32_t *d_status = p->d_status + base, *h_status = p->h_status + base;
    (void)d_status; (void)h_status;
    const int k = (int)group.size();
    const ac3mi_stream *s0 = group[0];
    const int nch = s0->src.channels;
    const size_t in_bytes = (size_t)1536 * nch * 2;
    ac3mi_encode_desc d = {s0->enc_rate, s0->enc_bitrate, nch};
    const int fb = ac3mi_encode_frame_bytes(&d);
    const int stride = (fb + 3) & ~3;
    uint8_t chmap[8];
    channel_map(nch, chmap);
    uint8_t *h_in = (uint8_t *)h_s16;
    parallel_for(k, [&](int i) {
        h_slots[i] = group[i]->slot;
        memcpy(h_in + (size_t)i * in_bytes, group[i]->buf, in_bytes);
    });
    // the samples (18 KB per frame) cross PCIe in chunks on the second stream while the context's stream encodes the chunk before
    if (hipMemcpyAsync(d_slots, h_slots, (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
    if (hipEventRecord(p->ev_join, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream2, p->ev_join, 0) != hipSuccess)      // earlier work on this staging
        return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
    const int n_chunks = chunks_for(k);
    for (int c = 0; c < n_chunks; c++) {
        const int lo = (int)((long long)k * c / n_chunks), hi = (int)((long long)k * (c + 1) / n_chunks), kc = hi - lo;
        if (hipMemcpyAsync((uint8_t *)d_s16 + (size_t)lo * in_bytes, h_in + (size_t)lo * in_bytes, (size_t)kc * in_bytes, hipMemcpyHostToDevice, ctx->stream2) != hipSuccess ||
            hipEventRecord(p->ev_kernel[c], ctx->stream2) != hipSuccess || hipStreamWaitEvent(ctx->stream, p->ev_kernel[c], 0) != hipSuccess) {
            (void)hipDeviceSynchronize();
            return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
        }
        ac3mi_set_state_slots(ctx, d_slots + lo);
        const int rc = ac3mi_encode_batch(ctx, &d, (const int16_t *)((const uint8_t *)d_s16 + (size_t)lo * in_bytes), chmap, p->d_last, p->d_csnr,
                                          d_frames + (size_t)lo * stride, stride, kc, 1, NULL);
        ac3mi_set_state_slots(ctx, NULL);
        if (rc != AC3MI_OK) { (void)hipDeviceSynchronize(); return fail(p, AC3MI_MMSYSERR_NOMEM, "encode batch"); }
    }
    if (hipMemcpyAsync(h_frames, d_frames, (size_t)k * stride, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return fail(p, AC3MI_MMSYSERR_NOMEM, "copy");
    if (ac3mi_sync(ctx) != AC3MI_OK) return fail(p, AC3MI_MMSYSERR_NOMEM, "sync");
    parallel_for(k, [&](int i) {
        memcpy(group[i]->frame, h_frames + (size_t)i * stride, (size_t)fb);
        group[i]->frame_bytes = fb;
    });
    return AC3MI_MMSYSERR_NOERROR;
}

}  // namespace

extern "C" {

ac3mi_pool *ac3mi_pool_create(ac3mi_ctx *ctx, int capacity)
{
    if (!ctx || capacity <= 0) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    ac3mi_pool *p = new ac3mi_pool();
    p->ctx = ctx;
    p->capacity = capacity;
    for (int i = capacity - 1; i >= 0; i--) p->free_slots.push_back(i);
    p->dirty.assign((size_t)capacity, 0);
    const size_t n = (size_t)capacity;
    bool ok = true;
    auto dev = [&](size_t bytes) -> void * { void *q = ac3mi_dev_alloc(ctx, bytes); ok = ok && q; return q; };
    auto pin = [&](size_t bytes) -> void * { void *q = nullptr; if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) { q = nullptr; ok = false; } return q; };
    p->d_delay = (float *)dev(n * 6 * 128 * 4);
    p->d_mixp = (float *)dev(n * 6 * 128 * 4);
    p->d_mixf = (int32_t *)dev(n * 6 * 4);
    p->d_lfsr = (uint16_t *)dev(n * 2);
    p->d_last = (int16_t *)dev(n * 6 * 256 * 2);
    p->d_csnr = (int32_t *)dev(n * 4);
    p->d_slots = (int32_t *)dev(n * 4);
    p->d_frames = (uint8_t *)dev(n * FRAME_STRIDE);
    p->d_s16 = (int16_t *)dev(n * PCM_FRAME_BYTES);
    p->d_status = (uint32_t *)dev(n * 4);
    p->h_slots = (int32_t *)pin(n * 4);
    p->h_frames = (uint8_t *)pin(n * FRAME_STRIDE);
    p->h_s16 = (int16_t *)pin(n * PCM_FRAME_BYTES);
    p->h_status = (uint32_t *)pin(n * 4);
    for (int c = 0; c < MAX_CHUNKS; c++)
        ok = ok && hipEventCreateWithFlags(&p->ev_kernel[c], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&p->ev_host[c], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        // every slot starts as a52_init (delay 0, lfsr 1, L52/parse.c:75) and AC3_encode_init (history 0, csnroffst 40,
        // ENC/ac3enc.cpp:1092) leave a new stream
        std::vector<uint16_t> ones(n, 1);
        std::vector<int32_t> forty(n, 40);
        ok = ac3mi_memset(ctx, p->d_delay, 0, n * 6 * 128 * 4) == AC3MI_OK && ac3mi_memset(ctx, p->d_last, 0, n * 6 * 256 * 2) == AC3MI_OK &&
             ac3mi_memset(ctx, p->d_mixp, 0, n * 6 * 128 * 4) == AC3MI_OK && ac3mi_memset(ctx, p->d_mixf, 0, n * 6 * 4) == AC3MI_OK &&
             ac3mi_memcpy_h2d(ctx, p->d_lfsr, ones.data(), n * 2) == AC3MI_OK &&
             ac3mi_memcpy_h2d(ctx, p->d_csnr, forty.data(), n * 4) == AC3MI_OK && ac3mi_sync(ctx) == AC3MI_OK;
    }
    if (!ok) { ac3mi_pool_destroy(p); return nullptr; }
    return p;
}

void ac3mi_pool_destroy(ac3mi_pool *p)
{
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)ac3mi_sync(p->ctx);
    void *dv[] = {p->d_mixp, p->d_mixf, p->d_delay, p->d_lfsr, p->d_last, p->d_csnr, p->d_slots, p->d_frames, p->d_s16, p->d_status};
    for (void *q : dv) if (q) ac3mi_dev_free(p->ctx, q);
    void *hv[] = {p->h_slots, p->h_frames, p->h_s16, p->h_status};
    for (void *q : hv) if (q) (void)hipHostFree(q);
    for (int c = 0; c < MAX_CHUNKS; c++) {
        if (p->ev_kernel[c]) (void)hipEventDestroy(p->ev_kernel[c]);
        if (p->ev_host[c]) (void)hipEventDestroy(p->ev_host[c]);
    }
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    delete p;
}

int ac3mi_stream_framesize(const ac3mi_wavefmt *f)
{
    // ac3_framesize, src/AC3ACM.cpp:432-488
    if (!f) return 0;
    // 24 kHz and 12 kHz hash to column 3 (the kbps column) exactly as in the reference: a 24 kHz / 384 kbps stream is
    // given 2 x 384 = 768 bytes per frame
    const int spsindex = (int)(f->samples_per_sec >> 6) & 3;
    if (f->block_align > 1)
        for (int i = 0; i < 19; i++)
            if (f->block_align == k_framesizes[i][spsindex] * 2) return f->block_align;
    if (f->avg_bytes_per_sec <= 81000u) {
        int selec = 0, diff = 0x7fffffff;
        for (int i = 0; i < 19; i++) {
            int d = (int)(f->avg_bytes_per_sec - 125u * (uint32_t)k_framesizes[i][3]);
            if (d == 0) return k_framesizes[i][spsindex] * 2;
            if (d < 0) d = -d;
            if (d < diff) { selec = i; diff = d; }
        }
        return k_framesizes[selec][spsindex] * 2;
    }
    return k_framesizes[18][spsindex] * 2;
}

int ac3mi_stream_open(ac3mi_pool *pool, const ac3mi_wavefmt *src, const ac3mi_wavefmt *dst, uint32_t driver_flags,
                      int query, ac3mi_stream **out)
{
    if (!pool || !src || !dst || (!query && !out)) return AC3MI_MMSYSERR_INVALPARAM;
    if (out && !query) *out = nullptr;
    int kbps = 0;
    bool decode;
    if (!is_valid_ac3(src)) {
        if (!is_valid_pcm(src, driver_flags)) return AC3MI_ACMERR_NOTPOSSIBLE;
        const bool dst_ac3 = is_valid_ac3(dst);
        if (!dst_ac3) {
            if (!is_valid_pcm(dst, driver_flags)) return AC3MI_ACMERR_NOTPOSSIBLE;
            if (!(driver_flags & AC3MI_ACM_MULTICHANNEL) && dst->format_tag == AC3MI_WAVE_FORMAT_PCM && dst->channels > 2)
                return AC3MI_MMSYSERR_NOTSUPPORTED;
        } else if (src->samples_per_sec < 32000) {
            return AC3MI_ACMERR_NOTPOSSIBLE;                        // no encoding at low sample rates (:1899)
        }
        if (dst->channels != src->channels) return AC3MI_ACMERR_NOTPOSSIBLE;
        if (dst->samples_per_sec != src->samples_per_sec) return AC3MI_MMSYSERR_NOTSUPPORTED;
        if (!dst_ac3) return AC3MI_MMSYSERR_NOERROR;                 // PCM -> PCM: copied by the driver
        // nAvgBytesPerSec must name one of the 19 bit rates (:1921-1945)
        kbps = (int)(dst->avg_bytes_per_sec / 125);
        int i;
        for (i = 0; i < 19; i++) if (kbps == k_framesizes[i][3]) break;
        if (i == 19) {
            if (dst->samples_per_sec == 44100)
                for (i = 0; i < 19; i++)
                    if ((uint32_t)((k_framesizes[i][1] * 2 * 44100 + 768) / 1536) == dst->avg_bytes_per_sec) { kbps = k_framesizes[i][3]; break; }
            if (i == 19) return AC3MI_MMSYSERR_NOTSUPPORTED;
        }
        decode = false;
    } else {
        const bool dst_pcm = is_valid_pcm(dst, driver_flags);
        if (!dst_pcm) {
            if (!is_valid_ac3(dst)) return AC3MI_ACMERR_NOTPOSSIBLE;
            if (dst->channels != src->channels) return AC3MI_ACMERR_NOTPOSSIBLE;
        } else if (!(driver_flags & AC3MI_ACM_MULTICHANNEL) && dst->format_tag == AC3MI_WAVE_FORMAT_PCM && dst->channels > 2) {
            return AC3MI_MMSYSERR_NOTSUPPORTED;
        }
        if (dst->samples_per_sec != src->samples_per_sec) return AC3MI_MMSYSERR_NOTSUPPORTED;
        if (!dst_pcm) return AC3MI_MMSYSERR_NOERROR;                 // AC-3 -> AC-3: copied by the driver
        // MapTab (src/AC3ASM.asm:59-112): destination is mono, stereo or the source's channel count;
        // 1 -> 2 is refused here (see ac3mi_stream.h)
        const int sc = src->channels, dc = dst->channels;
        if (!(dc == 1 || dc == 2 || dc == sc) || (sc == 1 && dc == 2)) return AC3MI_MMSYSERR_NOTSUPPORTED;
        decode = true;
    }
    if (query) return AC3MI_MMSYSERR_NOERROR;

    ac3mi::EncConfig cfg;
    if (!decode && !ac3mi::enc_config((int)src->samples_per_sec, kbps * 1000, src->channels, &cfg)) return AC3MI_ACMERR_NOTPOSSIBLE;   // AC3_encode_init == 0 (:1953)
    if (pool->free_slots.empty()) return AC3MI_MMSYSERR_NOMEM;
    ac3mi_stream *st = new ac3mi_stream();
    st->pool = pool;
    st->slot = pool->free_slots.back();
    pool->free_slots.pop_back();
    st->decode = decode;
    st->src = *src;
    st->dst = *dst;
    st->driver_flags = driver_flags;
    st->bufptr = st->buf;
    st->bufend = decode ? st->buf : st->frame;
    st->blocks = 0;
    st->flags = 0;
    st->phase = 3;
    st->hdr = nullptr;
    st->pcm_cur = (const uint8_t *)st->pcm;
    ac3mi_ctx *ctx = pool->ctx;
    bool ok = true;
    if (decode) {
        st->framelen = ac3mi_stream_framesize(src);
    } else {
        st->enc_cfg = cfg;
        st->enc_rate = (int)src->samples_per_sec;
        st->enc_bitrate = kbps * 1000;
        ac3mi_encode_desc d = {st->enc_rate, st->enc_bitrate, src->channels};
        st->framelen = ac3mi_encode_frame_bytes(&d);                // msd->framelen = AC3_encode_init(...)
    }
    if (pool->dirty[(size_t)st->slot]) {                            // a slot another stream has used
        const uint16_t one = 1;
        const int32_t c40 = 40;
        ok = ok && ac3mi_memset(ctx, pool->d_delay + (size_t)st->slot * 6 * 128, 0, 6 * 128 * 4) == AC3MI_OK;
        ok = ok && ac3mi_memset(ctx, pool->d_mixp + (size_t)st->slot * 6 * 128, 0, 6 * 128 * 4) == AC3MI_OK;
        ok = ok && ac3mi_memset(ctx, pool->d_mixf + (size_t)st->slot * 6, 0, 6 * 4) == AC3MI_OK;
        ok = ok && ac3mi_memcpy_h2d(ctx, pool->d_lfsr + st->slot, &one, 2) == AC3MI_OK;
        ok = ok && ac3mi_memset(ctx, pool->d_last + (size_t)st->slot * 6 * 256, 0, 6 * 256 * 2) == AC3MI_OK;
        ok = ok && ac3mi_memcpy_h2d(ctx, pool->d_csnr + st->slot, &c40, 4) == AC3MI_OK;
        ok = ok && ac3mi_sync(ctx) == AC3MI_OK;
    }
    pool->dirty[(size_t)st->slot] = 1;
    if (!ok) { pool->free_slots.push_back(st->slot); delete st; return AC3MI_MMSYSERR_NOMEM; }
    *out = st;
    return AC3MI_MMSYSERR_NOERROR;
}

int ac3mi_stream_close(ac3mi_stream *st)
{
    if (!st) return AC3MI_MMSYSERR_NOERROR;
    st->pool->free_slots.push_back(st->slot);
    delete st;
    return AC3MI_MMSYSERR_NOERROR;
}

int ac3mi_stream_size(const ac3mi_stream *st, int query, uint32_t in_bytes, uint32_t *out_bytes)
{
    // stream_size, src/AC3ACM.cpp:2139-2363
    if (!st || !out_bytes) return AC3MI_MMSYSERR_INVALPARAM;
    long len;
    if (query == AC3MI_STREAMSIZEF_SOURCE) {
        if (st->decode) {
            len = ((long)in_bytes + (st->framelen - 1)) / st->framelen;
            if (len < 1) len = 1;
            *out_bytes = (uint32_t)(len * 1536 * st->dst.block_align);
        } else {
            const int i = 1536 * st->src.block_align;
            long frames = ((long)in_bytes + (i - 1)) / i;
            if (frames < 1) frames = 1;
            const int fb = frame_bytes_for((int)(st->dst.avg_bytes_per_sec / 125), st->dst.samples_per_sec);
            *out_bytes = fb ? (uint32_t)(fb * frames) : 3840u;
        }
        return AC3MI_MMSYSERR_NOERROR;
    }
    if (query == AC3MI_STREAMSIZEF_DESTINATION) {
        if (st->decode) {
            len = (long)(in_bytes / (1536u * st->dst.block_align));
            if (len < 1) {
                if (in_bytes < 256u * st->dst.block_align) return AC3MI_ACMERR_NOTPOSSIBLE;
                *out_bytes = (uint32_t)(st->framelen + 2);
            } else {
                *out_bytes = (uint32_t)(len * (st->framelen + 2));
            }
        } else {
            const int fb = frame_bytes_for((int)(st->dst.avg_bytes_per_sec / 125), st->dst.samples_per_sec);
            long frames = fb ? (long)(in_bytes / (uint32_t)fb) : 1;
            if (frames < 1) frames = 1;
            *out_bytes = (uint32_t)(1536 * st->src.block_align * frames);
        }
        return AC3MI_MMSYSERR_NOERROR;
    }
    return AC3MI_MMSYSERR_NOTSUPPORTED;
}

int ac3mi_stream_convert_many(ac3mi_stream *const *streams, ac3mi_stream_header *const *hdrs, int n)
{
    if (n <= 0) return AC3MI_MMSYSERR_NOERROR;
    if (!streams || !hdrs) return AC3MI_MMSYSERR_INVALPARAM;
    ac3mi_pool *pool = nullptr;
    static unsigned long long round_counter = 0;
    const unsigned long long tag = ++round_counter;
    for (int i = 0; i < n; i++) {
        if (!streams[i] || !hdrs[i] || (hdrs[i]->src_len && !hdrs[i]->src) || (hdrs[i]->dst_len && !hdrs[i]->dst)) return AC3MI_MMSYSERR_INVALPARAM;
        if (pool && streams[i]->pool != pool) return AC3MI_MMSYSERR_INVALPARAM;
        pool = streams[i]->pool;
        if (streams[i]->round_tag == tag) return AC3MI_MMSYSERR_INVALPARAM;       // the same stream twice in one call
        streams[i]->round_tag = tag;
        streams[i]->hdr = hdrs[i];
        streams[i]->phase = 0;
    }
    if (hipSetDevice(pool->ctx->device) != hipSuccess) return AC3MI_MMSYSERR_NOMEM;
    typedef std::tuple<int, int, int, int, int, int> Key;
    bool first_pass = true;
    for (;;) {
        std::map<Key, std::vector<ac3mi_stream *>> dec, enc;
        std::vector<int> stops((size_t)n, STOP_DONE);
        g_trace.start();
        parallel_for(n, [&](int i) {
            ac3mi_stream *st = streams[i];
            if (st->phase != 3) stops[(size_t)i] = st->decode ? run_decode(st) : run_encode(st);
        });
        g_trace.lap(first_pass ? 0 : 5);
        first_pass = false;
        // (neighbours in the caller's array mostly share a configuration: the last group found is tried first)
        Key last_key(-1, -1, -1, -1, -1, -1);
        std::vector<ac3mi_stream *> *last_group = nullptr;
        for (int i = 0; i < n; i++) {
            ac3mi_stream *st = streams[i];
            const int stop = stops[(size_t)i];
            if (stop == STOP_DONE) continue;
            const Key key = stop == STOP_DECODE
                ? Key(st->acmod, st->lfeon, st->frame_bytes, st->req_flags, (int)(st->driver_flags & AC3MI_ACM_DYNAMICRANGE), st->dst.channels)
                : Key(st->enc_rate, st->enc_bitrate, st->src.channels, -2, 0, 0);
            if (!last_group || key != last_key) {
                last_group = stop == STOP_DECODE ? &dec[key] : &enc[key];
                last_key = key;
                if (last_group->empty()) last_group->reserve((size_t)n);
            }
            last_group->push_back(st);
        }
        g_trace.lap(1);
        if (dec.empty() && enc.empty()) break;
        int base = 0;
        for (auto &g : dec) { const int rc = decode_group(pool, g.second, base); if (rc) return rc; base += (int)g.second.size(); }
        for (auto &g : enc) { const int rc = encode_group(pool, g.second, base); if (rc) return rc; base += (int)g.second.size(); }
    }
    g_trace.report();
    return AC3MI_MMSYSERR_NOERROR;
}

int ac3mi_stream_convert(ac3mi_stream *stream, ac3mi_stream_header *hdr)
{
    return ac3mi_stream_convert_many(&stream, &hdr, 1);
}

}  // extern "C"
