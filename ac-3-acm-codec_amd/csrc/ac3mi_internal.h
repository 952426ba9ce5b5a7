// ac3mi_internal.h — host-side declarations shared by the translation units of libac3mi.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/ac3mi.h"

namespace ac3mi {

// Device-resident constant tables, built on the host in double precision.
struct DeviceTables {
    float2 *tw_long;    // [8][16]  merged lane twiddles, long block
    float2 *tw_short;   // [8][16]  merged lane twiddles, short block
    float *window;      // [256]    KBD alpha=5 window (L52/imdct.c:364-372)
};

struct MixPlan {
    int n_in, n_out, nfchans, in_lfe;
    // mix[o][c]: weight (-1, 0, +1) of input plane c in output plane o
    int8_t mix[6][6];
};

struct XformLaunch {
    const float *coef;
    const uint8_t *blksw;
    float *delay;
    float *pcm;
    int n_streams, frames;
    float bias;
    MixPlan plan;
};

// a52_downmix()/a52_downmix_init() semantics as a plane-mixing matrix; returns <0 if
// `output` is not a configuration liba52 grants for `acmod`.
int build_mix_plan(int acmod, int lfeon, int output, MixPlan *plan);

hipError_t launch_xform(const DeviceTables &tab, const XformLaunch &L, hipStream_t stream);

void build_host_tables(float *window256, float2 *tw_long /*[8][16]*/, float2 *tw_short /*[8][16]*/);

}  // namespace ac3mi

struct ac3mi_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    ac3mi::DeviceTables tab;
    std::string err;
};
