// profiles/hbm_calibrate.hip — measurement aid (not part of libac3mi.so): what this MI355X delivers for plain
// streaming copies, next to the access pattern of ac3mi::xform_kernel, and what one SIMD issues in scalar and
// vector instructions.  Built by profiles/Makefile, run on the GPU box (profiles/run_r02.sh); the output is
// committed as profiles/r02_hbm_calibration.txt.
//
//   float4 grid-stride copy   the probe MI355X_MICROARCH.md:36 quotes at 6.29 TB/s (read + write bytes / time)
//   float2 / 1 KiB-row copy   8 B per lane, 64 B per 8-lane group, rows of 1 KiB: the transform's addressing
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_f4_stride(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// four independent 16-byte loads in flight per lane before the stores
__global__ __launch_bounds__(256) void copy_f4_stride_x4(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        const float4 v0 = a[i], v1 = a[i + step], v2 = a[i + 2 * step], v3 = a[i + 3 * step];
        b[i] = v0; b[i + step] = v1; b[i + 2 * step] = v2; b[i + 3 * step] = v3;
    }
    for (; i < n; i += step) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy_f4_flat(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy_f2_stride(const float2 *__restrict__ a, float2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy_f4_nt(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v;
        v.x = __builtin_nontemporal_load(&a[i].x); v.y = __builtin_nontemporal_load(&a[i].y);
        v.z = __builtin_nontemporal_load(&a[i].z); v.w = __builtin_nontemporal_load(&a[i].w);
        __builtin_nontemporal_store(v.x, &b[i].x); __builtin_nontemporal_store(v.y, &b[i].y);
        __builtin_nontemporal_store(v.z, &b[i].z); __builtin_nontemporal_store(v.w, &b[i].w);
    }
}
__global__ __launch_bounds__(256) void read_f4(const float4 *__restrict__ a, float *out, size_t n)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 1234.5678f) out[0] = s;
}
__global__ __launch_bounds__(256) void write_f4(float4 *__restrict__ b, size_t n)
{
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
// the transform's addressing: an 8-lane group owns a chain = one 1 KiB plane out of every 6 KiB block (6 chains per
// stream-block), lane l8 moves float2 number l8 + 8 n (n < 16) of the plane; 36 planes per chain (6 blocks x ... ) are
// walked in sequence.  planes = number of 1 KiB planes, chains = planes / blocks_per_chain.
__global__ __launch_bounds__(256) void copy_rows_group8(const float2 *__restrict__ a, float2 *__restrict__ b, int n_chains, int blocks,
                                                        int chains_per_stream)
{
    const int group = blockIdx.x * 32 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    if (group >= n_chains) return;
    const int s = group / chains_per_stream, o = group - s * chains_per_stream;
    const size_t base = ((size_t)s * blocks * chains_per_stream + o) * 128;       // float2 units; plane = 128 float2
    for (int blk = 0; blk < blocks; blk++) {
        const float2 *p = a + base + (size_t)blk * chains_per_stream * 128 + l8;
        float2 *q = b + base + (size_t)blk * chains_per_stream * 128 + l8;
        float2 v[16];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = p[8 * n];
#pragma unroll
        for (int n = 0; n < 16; n++) q[8 * n] = v[n];
    }
}
// same bytes, but a WAVEFRONT owns a plane at a time: 16 B per lane, 1 KiB contiguous per instruction
__global__ __launch_bounds__(256) void copy_rows_wave(const float4 *__restrict__ a, float4 *__restrict__ b, int n_streams, int blocks,
                                                      int chains_per_stream)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_streams) return;
    const size_t base = (size_t)w * blocks * chains_per_stream * 64;                // float4 units; plane = 64 float4
    for (int blk = 0; blk < blocks; blk++) {
        float4 v[6];
        for (int c = 0; c < chains_per_stream; c++) v[c] = a[base + ((size_t)blk * chains_per_stream + c) * 64 + lane];
        for (int c = 0; c < chains_per_stream; c++) b[base + ((size_t)blk * chains_per_stream + c) * 64 + lane] = v[c];
    }
}

// variants of the transform's addressing -------------------------------------------------------------------------
// (a) 16 bytes per lane: lane l8 moves float4 number l8 + 8 n (n < 8) of its plane: 128 contiguous bytes per group
__global__ __launch_bounds__(256) void copy_rows_group8_f4(const float4 *__restrict__ a, float4 *__restrict__ b, int n_chains, int blocks, int cps)
{
    extern __shared__ float pad_[];
    const int group = blockIdx.x * 32 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    if (group >= n_chains) return;
    const int s = group / cps, o = group - s * cps;
    const size_t base = ((size_t)s * blocks * cps + o) * 64;
    for (int blk = 0; blk < blocks; blk++) {
        const float4 *p = a + base + (size_t)blk * cps * 64 + l8;
        float4 *q = b + base + (size_t)blk * cps * 64 + l8;
        float4 v[8];
#pragma unroll
        for (int n = 0; n < 8; n++) v[n] = p[8 * n];
#pragma unroll
        for (int n = 0; n < 8; n++) q[8 * n] = v[n];
    }
}
// (b) as the 8-byte baseline, but the next block's loads are issued before this block's stores
__global__ __launch_bounds__(256) void copy_rows_group8_pf(const float2 *__restrict__ a, float2 *__restrict__ b, int n_chains, int blocks, int cps)
{
    extern __shared__ float pad_[];
    const int group = blockIdx.x * 32 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    if (group >= n_chains) return;
    const int s = group / cps, o = group - s * cps;
    const size_t base = ((size_t)s * blocks * cps + o) * 128;
    float2 v[16], w[16];
#pragma unroll
    for (int n = 0; n < 16; n++) v[n] = a[base + l8 + 8 * n];
    for (int blk = 0; blk < blocks; blk++) {
        const int nb = blk + 1 < blocks ? blk + 1 : blk;
#pragma unroll
        for (int n = 0; n < 16; n++) w[n] = a[base + (size_t)nb * cps * 128 + l8 + 8 * n];
        float2 *q = b + base + (size_t)blk * cps * 128 + l8;
#pragma unroll
        for (int n = 0; n < 16; n++) q[8 * n] = v[n];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = w[n];
    }
}
// (c) the baseline with dynamic LDS padding (occupancy cap) - same kernel body as copy_rows_group8
__global__ __launch_bounds__(256) void copy_rows_group8_pad(const float2 *__restrict__ a, float2 *__restrict__ b, int n_chains, int blocks, int cps)
{
    extern __shared__ float pad_[];
    const int group = blockIdx.x * 32 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    if (group >= n_chains) return;
    const int s = group / cps, o = group - s * cps;
    const size_t base = ((size_t)s * blocks * cps + o) * 128;
    for (int blk = 0; blk < blocks; blk++) {
        const float2 *p = a + base + (size_t)blk * cps * 128 + l8;
        float2 *q = b + base + (size_t)blk * cps * 128 + l8;
        float2 v[16];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = p[8 * n];
#pragma unroll
        for (int n = 0; n < 16; n++) q[8 * n] = v[n];
    }
}
// (d) a workgroup streams whole stream-blocks: 256 threads x 16 bytes = 4 KB contiguous per instruction, 6 KB per block of a
//     stream, streams of a workgroup back to back (what a transform that redistributes through LDS would read and write)
__global__ __launch_bounds__(256) void copy_streams_flat(const float4 *__restrict__ a, float4 *__restrict__ b, int n_streams, int streams_per_wg, int f4_per_stream)
{
    extern __shared__ float pad_[];
    const int s0 = blockIdx.x * streams_per_wg;
    const int ns = s0 + streams_per_wg <= n_streams ? streams_per_wg : n_streams - s0;
    if (ns <= 0) return;
    const size_t base = (size_t)s0 * f4_per_stream, n = (size_t)ns * f4_per_stream;
    for (size_t i = threadIdx.x; i < n; i += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + 256 * k < n ? a[base + i + 256 * k] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + 256 * k < n) b[base + i + 256 * k] = v[k];
    }
}

// (e) block-parallel: an 8-lane group per (chain, block) row, a workgroup of 8 * cpw * 6 lanes takes cpw chains of one-frame
//     streams (6 blocks each), moves its rows in one go and ends (what a transform that passes the overlap tails through LDS
//     instead of walking a chain would read and write)
__global__ void copy_rows_blockpar(const float2 *__restrict__ a, float2 *__restrict__ b, int n_chains, int cpw)
{
    extern __shared__ float pad_[];
    const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const int chain = blockIdx.x * cpw + g / 6, blk = g - 6 * (g / 6);
    if (g >= 6 * cpw || chain >= n_chains) return;
    const int s = chain / 6, o = chain - 6 * s;
    const size_t base = (((size_t)s * 6 + blk) * 6 + o) * 128 + l8;
    float2 v[16];
#pragma unroll
    for (int n = 0; n < 16; n++) v[n] = a[base + 8 * n];
#pragma unroll
    for (int n = 0; n < 16; n++) b[base + 8 * n] = v[n];
}

// ---- issue probes: 8 wavefronts per SIMD, nothing but register arithmetic --------------------------------------
__global__ __launch_bounds__(256) void salu_probe(uint32_t *out, int iters)
{
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
    if ((s0 ^ s1 ^ s2 ^ s3) == 0x12345678u) out[0] = s0;
}
// 16 VALU + 16 SALU per iteration, interleaved in one wavefront's stream
__global__ __launch_bounds__(256) void mixed_probe(uint32_t *out, int iters)
{
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_add_u32 %4, %4, 1\n s_add_u32 %0, %0, 1\n v_xor_b32 %5, %5, 3\n s_xor_b32 %1, %1, 3\n"
            "v_add_u32 %6, %6, 5\n s_add_u32 %2, %2, 5\n v_xor_b32 %7, %7, 7\n s_xor_b32 %3, %3, 7\n"
            "v_add_u32 %4, %4, 1\n s_add_u32 %0, %0, 1\n v_xor_b32 %5, %5, 3\n s_xor_b32 %1, %1, 3\n"
            "v_add_u32 %6, %6, 5\n s_add_u32 %2, %2, 5\n v_xor_b32 %7, %7, 7\n s_xor_b32 %3, %3, 7\n"
            "v_add_u32 %4, %4, 1\n s_add_u32 %0, %0, 1\n v_xor_b32 %5, %5, 3\n s_xor_b32 %1, %1, 3\n"
            "v_add_u32 %6, %6, 5\n s_add_u32 %2, %2, 5\n v_xor_b32 %7, %7, 7\n s_xor_b32 %3, %3, 7\n"
            "v_add_u32 %4, %4, 1\n s_add_u32 %0, %0, 1\n v_xor_b32 %5, %5, 3\n s_xor_b32 %1, %1, 3\n"
            "v_add_u32 %6, %6, 5\n s_add_u32 %2, %2, 5\n v_xor_b32 %7, %7, 7\n s_xor_b32 %3, %3, 7\n"
            : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "scc");
    }
    if ((s0 ^ s1 ^ s2 ^ s3 ^ a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) out[0] = s0;
}
__global__ __launch_bounds__(256) void valu_probe(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_xor_b32 %3, %3, 7\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) out[0] = a0;
}

template <class F> static double time_ms(F launch, int reps = 10)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    launch();
    launch();
    CHK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CHK(hipEventRecord(e0, 0));
        launch();
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    const size_t frames = 65536, bytes = frames * 36 * 1024;           // 2.4 GB: the bench's coefficient (or PCM) array
    float4 *a, *b;
    float *d_out;
    CHK(hipMalloc(&a, 2 * bytes));
    CHK(hipMalloc(&b, 2 * bytes));
    CHK(hipMalloc(&d_out, 64));
    CHK(hipMemset(a, 1, 2 * bytes));
    CHK(hipMemset(b, 0, 2 * bytes));
    printf("# copies: GB/s = (bytes read + bytes written) / median time of 10 launches\n");
    for (size_t sz : {bytes / 2, bytes, 2 * bytes}) {
        const size_t n4 = sz / 16;
        for (int wg_per_cu : {4, 8, 16, 32}) {
            const int grid = cus * wg_per_cu;
            double ms = time_ms([&] { hipLaunchKernelGGL(copy_f4_stride, dim3(grid), dim3(256), 0, 0, a, b, n4); });
            printf("copy float4 grid-stride   %5.2f GB/array grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, grid, ms, 2.0 * sz / ms / 1e6);
        }
        double ms = time_ms([&] { hipLaunchKernelGGL(copy_f4_stride_x4, dim3(cus * 8), dim3(256), 0, 0, a, b, n4); });
        printf("copy float4 stride, 4 deep %5.2f GB/array grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, cus * 8, ms, 2.0 * sz / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_f4_flat, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, a, b, n4); });
        printf("copy float4 one per lane  %5.2f GB/array            : %.3f ms  %.0f GB/s\n", sz / 1e9, ms, 2.0 * sz / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_f2_stride, dim3(cus * 16), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, sz / 8); });
        printf("copy float2 grid-stride   %5.2f GB/array grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, cus * 16, ms, 2.0 * sz / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_f4_nt, dim3(cus * 8), dim3(256), 0, 0, a, b, n4); });
        printf("copy float4 non-temporal  %5.2f GB/array grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, cus * 8, ms, 2.0 * sz / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(read_f4, dim3(cus * 8), dim3(256), 0, 0, a, d_out, n4); });
        printf("read  float4 grid-stride  %5.2f GB       grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, cus * 8, ms, 1.0 * sz / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(write_f4, dim3(cus * 8), dim3(256), 0, 0, b, n4); });
        printf("write float4 grid-stride  %5.2f GB       grid %5d : %.3f ms  %.0f GB/s\n", sz / 1e9, cus * 8, ms, 1.0 * sz / ms / 1e6);
    }
    {
        const int chains = (int)frames * 6;
        double ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_group8, dim3((chains + 31) / 32), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, chains, 6, 6); });
        printf("copy 1 KiB rows, 8-lane group per chain (the transform's addressing) 2.42 GB/array : %.3f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_wave, dim3(((int)frames + 3) / 4), dim3(256), 0, 0, a, b, (int)frames, 6, 6); });
        printf("copy 1 KiB rows, wavefront per stream, 16 B per lane                  2.42 GB/array : %.3f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    }
    {
        const int chains = (int)frames * 6;
        const int grid = (chains + 31) / 32;
        printf("# variants of the transform's addressing (2.42 GB per array, 32 chains per workgroup)\n");
        for (int lds : {0, 40960, 53248, 81920}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_group8_pad, dim3(grid), dim3(256), lds, 0, (const float2 *)a, (float2 *)b, chains, 6, 6); });
            printf("8 B/lane baseline, %5d B LDS per workgroup (occupancy cap): %.3f ms  %.0f GB/s\n", lds, ms, 2.0 * bytes / ms / 1e6);
        }
        for (int lds : {0, 53248}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_group8_f4, dim3(grid), dim3(256), lds, 0, a, b, chains, 6, 6); });
            printf("16 B/lane (128 B per group), %5d B LDS: %.3f ms  %.0f GB/s\n", lds, ms, 2.0 * bytes / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_group8_pf, dim3(grid), dim3(256), lds, 0, (const float2 *)a, (float2 *)b, chains, 6, 6); });
            printf("8 B/lane, next block's loads before this block's stores, %5d B LDS: %.3f ms  %.0f GB/s\n", lds, ms, 2.0 * bytes / ms / 1e6);
        }
        // how much a workgroup may stream before the rate drops: contiguous chunks of 4 .. 36 KB per workgroup
        for (int kb : {4, 8, 12, 24, 36}) {
            const int f4 = kb * 64, g2 = (int)(bytes / 16 / f4);
            double ms = time_ms([&] { hipLaunchKernelGGL(copy_streams_flat, dim3(g2), dim3(256), 0, 0, a, b, g2, 1, f4); });
            printf("contiguous %2d KB per workgroup, then the workgroup ends: %.3f ms  %.0f GB/s\n", kb, ms, 2.0 * (double)g2 * f4 * 16 / ms / 1e6);
        }
        for (int cpw : {1, 2, 4, 5, 8}) {
            const int threads = ((8 * 6 * cpw + 63) / 64) * 64, g3 = (chains + cpw - 1) / cpw;
            for (int lds : {0, 20000, 40000}) {
                double ms = time_ms([&] { hipLaunchKernelGGL(copy_rows_blockpar, dim3(g3), dim3(threads), lds, 0, (const float2 *)a, (float2 *)b, chains, cpw); });
                printf("block-parallel rows, %d chain(s) x 6 blocks per workgroup of %3d lanes, %5d B LDS: %.3f ms  %.0f GB/s\n", cpw, threads, lds, ms, 2.0 * bytes / ms / 1e6);
            }
        }
        for (int spw : {1, 2, 4, 8}) {
            for (int lds : {0, 40960}) {
                const int g2 = ((int)frames + spw - 1) / spw;
                double ms = time_ms([&] { hipLaunchKernelGGL(copy_streams_flat, dim3(g2), dim3(256), lds, 0, a, b, (int)frames, spw, 36 * 64); });
                printf("whole streams (36 KB contiguous each), %d per workgroup, %5d B LDS: %.3f ms  %.0f GB/s\n", spw, lds, ms, 2.0 * bytes / ms / 1e6);
            }
        }
    }
    printf("# issue probes: 8 wavefronts per SIMD, 10^9 instructions per second and SIMD\n");
    {
        const int iters = 8192, grid = cus * 8;
        uint32_t *o = (uint32_t *)d_out;
        double ms = time_ms([&] { hipLaunchKernelGGL(valu_probe, dim3(grid), dim3(256), 0, 0, o, iters); }, 5);
        printf("valu only : %.3f ms  %.3f G VALU/s/SIMD\n", ms, 8.0 * iters * 32 / (ms * 1e-3) / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(salu_probe, dim3(grid), dim3(256), 0, 0, o, iters); }, 5);
        printf("salu only : %.3f ms  %.3f G SALU/s/SIMD\n", ms, 8.0 * iters * 32 / (ms * 1e-3) / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mixed_probe, dim3(grid), dim3(256), 0, 0, o, iters); }, 5);
        printf("16 valu + 16 salu interleaved : %.3f ms  %.3f G VALU/s/SIMD + %.3f G SALU/s/SIMD\n", ms, 8.0 * iters * 16 / (ms * 1e-3) / 1e9,
               8.0 * iters * 16 / (ms * 1e-3) / 1e9);
        for (int wg : {1, 2, 4}) {
            ms = time_ms([&] { hipLaunchKernelGGL(salu_probe, dim3(cus * wg), dim3(256), 0, 0, o, iters); }, 5);
            printf("salu only, %d wavefront(s) per SIMD : %.3f ms  %.3f G SALU/s/SIMD\n", wg, ms, (double)wg * iters * 32 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
