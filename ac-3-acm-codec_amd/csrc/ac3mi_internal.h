// ac3mi_internal.h — host-side declarations shared by the translation units of libac3mi.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/ac3mi.h"

namespace ac3mi {

// decoder front-end tables (device copy)
struct DecTables {
    int8_t la_neg[256];
    uint16_t hth[3][50];
    int8_t width[64];
    uint8_t band_end[30];
    uint8_t band_of_bin[256];   // band of a bin (bins 0..27 are their own band)
    float qlev[48];     // [0,3) 3-level  [3,8) 5-level  [8,16) 7-level  [16,27) 11-level  [27,43) 15-level
    // dequantised value of member m of a code: 3-level [code*3+m] at 0 (32 codes), 5-level at 96 (128 codes),
    // 11-level [code*2+m] at 480 (128 codes), 7-level [code] at 736, 15-level [code] at 744; reserved codes give 0
    float qtab[760];
    uint32_t desc[128];     // per row byte of the bap rows: see mant_desc (decode_common.h)
};

// encoder tables (device copy): ENC/ac3tab.h + the runtime tables of AC3_encode_init
struct EncTables {
    int16_t win[256];           // ac3_window, Q15
    int16_t cos[64], sin[64];   // costab / sintab (fft_init(7))
    int16_t xcos[128], xsin[128];
    uint8_t bitrev[128];
    uint8_t latab[256];
    uint16_t hth[50][3];
    uint8_t baptab[64];
    uint8_t band_of_bin[256];   // masktab
    uint8_t band_start[51];     // bndtab (entry 50 = 0 as in ac3_common_init)
    uint8_t band_size[50];      // bndsz
    uint16_t crc_tab[256];
};

// what AC3_encode_init derives from (freq, bitrate, channels): ENC/ac3enc.cpp:1019-1110
struct EncConfig {
    int nch, nfbw, lfe, acmod, fscod, halfrate, bsid, frmsizecod, frame_words;
};

// Device-resident constant tables, built on the host in double precision.
struct DeviceTables {
    float2 *tw_long;    // [8][16]  merged lane twiddles, long block
    float2 *tw_short;   // [8][16]  merged lane twiddles, short block
    float *window;      // [256]    KBD alpha=5 window (L52/imdct.c:364-372)
    DecTables *dec;     // bit-allocation / dequantiser tables
    uint16_t *lfsr_seq; // [65535] dither LFSR states in cycle order from state 1 (L52/parse.c:310-319)
    uint16_t *lfsr_idx; // [65536] inverse: position of a state in the cycle
    EncTables *enc;
};

struct MixPlan {
    int n_in, n_out, nfchans, in_lfe;
    // mix[o][c]: weight (-1, 0, +1) of input plane c in output plane o
    int8_t mix[6][6];
    // Surround planes that liba52 mixes at level slev into MONO / STEREO / 3F (bit c = input plane c), 0 elsewhere: when a
    // frame's slev is 0 liba52 neither transforms nor mixes them (L52/parse.c:900-913, downmix.c:494-583), which is not a
    // linear mix at level 0 as far as their overlap tails go - see XformLaunch::mix_pending.
    uint8_t surr_mask;
    // ... and the OUTPUT planes that then come out without the bias when the block takes the per-channel path: for 2/1 and
    // 2/2 to STEREO and 3/1 and 3/2 to 3F a52_downmix returns at `if (slev == 0) break;` (downmix.c:530-583) before the
    // mixer that would have added it, and a52_downmix_coeff had told a52_block not to add it in those channels' transforms
    uint8_t nobias_mask;
};

struct XformLaunch {
    const float *coef;
    const uint8_t *blksw;
    float *delay;
    float *pcm;
    int n_streams, frames;
    int blocks = 0;         // 0: frames x 6 blocks per stream; else exactly this many blocks (a52_imdct_512/256 hooks: 1)
    float bias;
    MixPlan plan;
    const int32_t *slot;    // optional per-stream state slot indices (device), see ac3mi_set_state_slots
    int delay_stride;
    // s16 output instead of float planes: [n_streams][frames][6][256][n_out] interleaved in WAVE channel order, what the
    // reference's MapTab converters make of a52_samples() at bias 384 (src/AC3ASM.asm; s16_channel_map).  pcm is unused then.
    int16_t *pcm16 = nullptr;
    int s16_flags = 0;      // liba52 output flags of the planes (selects the channel order)
    // liba52's overlap bookkeeping around frames whose surround mix level is 0 (ac3mi_set_mix_state): per frame "slev is 0"
    // from the decode front end, per output chain the surround planes' share of the overlap tail that is held back
    // ([streams or slots][n_out][128], laid out like `delay`) and its flags ([streams or slots][6]: bit 0 liba52's
    // `downmixed`, bit 1 share pending).  All three null: a plain linear mix.
    const uint8_t *zs = nullptr;
    float *mix_pending = nullptr;
    int32_t *mix_flags = nullptr;
};

// a52_downmix()/a52_downmix_init() semantics as a plane-mixing matrix; returns <0 if
// `output` is not a configuration liba52 grants for `acmod`.
int build_mix_plan(int acmod, int lfeon, int output, MixPlan *plan);

hipError_t launch_xform(const DeviceTables &tab, const XformLaunch &L, hipStream_t stream);

struct DecodeLaunch {
    const uint8_t *frames;
    int frame_bytes, frame_stride, n_streams, frames_per_stream;
    int req_flags, acmod, lfeon, dynrng_on;
    float level;
    float *coef;            // [S][F][6][n_in][256]
    uint8_t *blksw;         // [S][F][6][nfchans]
    uint32_t *status;       // [S][F]
    uint16_t *lfsr;         // [S]
    uint8_t *tap_exp;
    int8_t *tap_bap;
    uint8_t *zs = nullptr;              // optional [S][F]: 1 = the frame's surround channels are mixed at level 0 (see XformLaunch)
    float *dyn_out = nullptr;           // [S][F][6][2] range factors of the stream's dynamic-range words (NaN: none)
    const float *dyn_in = nullptr;      // [S][F][6][2] replacements (NaN: keep)
    const int32_t *slot;
    // few long streams: counting pass + LFSR prefix + one wavefront per frame (decode.hip, MODE 1/2)
    int frame_parallel;
    uint32_t *frame_draws;  // [S][F] workspace
    uint16_t *frame_lfsr;   // [S][F] workspace
    // split front end (parse kernel + one wavefront per audio block): workspaces, all or none
    int split = 0;
    void *ws_desc = nullptr;        // [S][F][6] BlkDesc (80 bytes)
    uint8_t *ws_rows = nullptr;     // [S][F][6][7][512]
    float *ws_cplco = nullptr;      // [S][F][6][90]
    uint32_t *ws_fpos = nullptr;    // [S][F]
    // split front end, one-frame streams, identity routing: mantissas + transform in one kernel (decode_mx.hip) - no
    // coefficient planes in HBM (coef unused), the caller does not launch the transform
    const XformLaunch *fuse = nullptr;
};
hipError_t launch_decode(const DeviceTables &tab, const DecodeLaunch &L, hipStream_t stream);
// decode_wg.hip: one workgroup per stream; X == nullptr: coefficient planes (+ taps) to HBM as launch_decode does;
// else the transform is fused in (identity routing only: returns hipErrorInvalidValue for a mixing plan)
hipError_t launch_decode_wg(const DeviceTables &tab, const DecodeLaunch &L, const XformLaunch *X, int grid_cap, hipStream_t stream);
void build_dec_tables(DecTables *t, uint16_t *lfsr_seq /*[65535]*/, uint16_t *lfsr_idx /*[65536]*/);

struct EncodeLaunch {
    EncConfig cfg;
    const int16_t *pcm;         // [S][F][1536][nch]
    int16_t *last;              // [S][nch][256]
    int32_t *csnr;              // [S]
    uint8_t *frames;            // [S][F][stride]
    int frame_stride, n_streams, frames_per_stream;
    uint8_t chmap[8];
    int32_t *ws_mdct;           // [S][F][6][nch][256]
    bool mdct_full_rows = false; // the rows are a stage tap: all 256 bins are stored, not only the coded ones
    uint8_t *ws_expo;           // [S][F][6][nch][256]
    int8_t *ws_shift;           // [S][F][6][nch]
    uint8_t *ws_eexp;           // [S][F][6][nch][256] encoded exponents
    int16_t *ws_emask;          // [S][F][6][nch][50]  masking curves minus the floor
    uint8_t *ws_strat;          // [S][F][6][nch]
    int32_t *ws_ebits;          // [S][F][nch]
    int32_t *ws_snr;            // [S][F][2] search results between the parts of the split pack kernel (may be null)
    uint32_t *ws_memo;          // [S][F][8] tabulated fit verdicts (may be null: the replay then costs everything itself)
    uint8_t *tap_eexp, *tap_bap, *tap_strat;
    int32_t *tap_snr;
    const int32_t *slot;
    int pack_mode = 0;          // ac3mi_set_encode_mode
    const uint32_t *search_hint = nullptr;     // transcode: per frame, an offset 16 csnroffst + fsnroffst near which to start costing (stride in dwords)
    int search_hint_stride = 0;
};
hipError_t launch_encode(const DeviceTables &tab, const EncodeLaunch &E, hipStream_t stream);
hipError_t launch_enc_history(const EncodeLaunch &E, hipStream_t stream);
void build_enc_tables(EncTables *t);
int enc_config(int freq, int bitrate, int channels, EncConfig *c);   // 0 = rejected (AC3_encode_init returns 0)

hipError_t launch_convert_s16(const float *planes, int16_t *out, int flags, size_t n_blocks, hipStream_t stream);
int s16_channel_map(int flags, int map[6]);

void build_host_tables(float *window256, float2 *tw_long /*[8][16]*/, float2 *tw_short /*[8][16]*/);

}  // namespace ac3mi

struct ac3mi_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // second stream: the byte-stream layer's PCIe copies beside the kernels (stream.hip)
    hipStream_t stream2;
    int encode_mode;        // ac3mi_set_encode_mode
    ac3mi::DeviceTables tab;
    // decode workspace (coefficient planes + block-switch flags between the two kernels)
    float *ws_coef;
    uint8_t *ws_blksw;
    size_t ws_coef_bytes, ws_blksw_bytes;
    // encode workspace (MDCT coefficients, exponents, block exponents between the two kernels)
    void *ws_enc;
    size_t ws_enc_bytes;
    // transcode workspace (float PCM and s16 PCM between the decoder and the encoder)
    void *ws_tc;
    size_t ws_tc_bytes;
    // optional state-slot indirection for the next batch calls (ac3mi_set_state_slots)
    const int32_t *slots;
    // optional liba52-exact overlap state around frames with surround level 0 (ac3mi_set_mix_state)
    float *mix_pending;
    int32_t *mix_flags;
    long long tile_frames;  // workspace bound: batches above this many frames go through in tiles of whole streams (0 = never)
    int decode_mode;        // ac3mi_set_decode_mode
    uint32_t *ws_draws;     // [S][F] draw counts + [S][F] u16 frame-start LFSR states (decode, frame-parallel)
    size_t ws_draws_bytes;
    void *ws_split;         // descriptors, rows, coupling coordinates, generator positions between the split front end's kernels
    size_t ws_split_bytes;
    std::string err;
};
