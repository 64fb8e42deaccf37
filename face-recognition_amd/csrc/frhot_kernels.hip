// HIP kernels of the recognition hot path for MI355X (gfx950, wave64, MFMA).
//
// Layout conventions (all device memory):
//   activations  fp16 NHWC with a one-pixel zero halo: [B, H+2, W+2, C]; kernels
//                write interiors only, so the halo written at allocation stays 0
//                and 3x3 "same" convolutions need no bounds checks;
//   conv weights fp16 [Cout][3][3][Cin]  == GEMM operand [N][K], K = 9*Cin;
//   fc weights   fp16 [N][K];
//   gallery      fp16 [rows][D] unit rows + int32 key per row.
//
// One GEMM core serves the 3x3 convolutions (implicit GEMM), the two fc layers
// (split-K) and the gallery scan: 64-deep K steps staged global -> LDS with
// 16-byte LDS-DMA loads (two stages), XOR-swizzled 128-byte LDS rows read with
// ds_read_b128, v_mfma_f32_16x16x32_f16 accumulation in fp32.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "frhot_kernels.h"

#ifndef FR_LOOP_MIN_K
#define FR_LOOP_MIN_K 10
#endif
#ifndef FR_WINOGRAD
#define FR_WINOGRAD 0   // 1: build the Winograd prototype of the 28 x 28 x 512 layer (micro-benchmark only)
#endif
#ifndef FR_WABL
#define FR_WABL 0   // ablation builds of the Winograd kernel (wrong results, timing only)
#endif
#ifndef FR_DBG
#define FR_DBG 0   // diagnostic builds only: 1 no conv stores, 2 no patch DMA, 4 no MFMA, 8 in-kernel stamps (patch conv)
#endif

namespace frhot {

#if FR_DBG & 64
// scan diagnosis, per block (wave 0): {entry, loop begin, loop end, exit, cycles inside tile ends, tile ends,
// insertion rounds, cycles before the per-group work, in the per-group preludes, in the insertion loops}
__device__ unsigned long long g_scan_stamp[256 * 16];
__device__ unsigned long long g_scan_tiles[256 * 20];     // wave 0: clock at the end of each of the first 20 tiles
#endif
#if FR_DBG & (8 | 32 | 64 | 256)
__device__ unsigned long long g_stamp[64];
#define FR_STAMP(var)                                                                        \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define FR_STAMP(var) do { } while (0)
#endif

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define FR_DEV __device__ __forceinline__


// ---------------------------------------------------------------- helpers ---
FR_DEV void glds16(const void* g, void* lds_wave_base) {
    // 64 lanes x 16 B: lane i's bytes land at lds_wave_base + 16*i.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same with the non-temporal cache policy: for bytes that exactly one CU reads exactly once.
FR_DEV void glds16_nt(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// LDS accesses written as the instructions themselves.  The compiler orders every LDS access
// it emits behind ALL LDS-DMA in flight (s_waitcnt vmcnt(0)): it cannot tell that a small table
// beside the rings is never a DMA target.  These helpers keep the ring's prefetch in flight.
FR_DEV unsigned lds_offset(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
FR_DEV void lds_max_u32(unsigned addr, unsigned v) {
    asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// min of the first K of N4*4 consecutive words at addr (16-byte aligned); waits for its own reads
template <int N4, int K>
FR_DEV unsigned lds_min_words(unsigned addr) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v[N4];
    if constexpr (N4 == 2)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(addr) : "memory");
    else if constexpr (N4 == 3)
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(addr) : "memory");
    else
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
    static_assert(N4 >= 2 && N4 <= 4 && K <= N4 * 4, "slot words");
    unsigned m = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < N4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i * 4 + j < K) m = v[i][j] < m ? v[i][j] : m;
    return m;
}

FR_DEV uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch),
// so give every XCD a contiguous run of logical tiles.  Bijective for any n.
FR_DEV int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ------------------------------------------------------------- GEMM core ---
// A block tile is RX "streamed" rows (pixels / batch rows / gallery rows) by RY
// "resident" rows (output channels / queries), 64 K-elements per step.  In LDS a
// tile row is 128 bytes = 8 chunks of 16 B; chunk c of row r is stored at chunk
// position c ^ (r & 7).  LDS-DMA writes lane-linear, so the swizzle is applied to
// the per-lane SOURCE address and again to the read address.
// 256 threads = 4 waves arranged WX x WY; each wave owns (RX/WX) x (RY/WY).
// X_IS_A: the streamed operand is the MFMA A operand (its rows land in the
// accumulator registers, the resident operand's rows on the lanes); otherwise
// the resident operand is A.
template <int RX, int RY, int WX, int WY, bool X_IS_A>
struct GemmCore {
    static_assert(WX * WY == 4, "4 waves");
    static constexpr int TX = RX / WX, TY = RY / WY;
    static constexpr int NX = TX / 16, NY = TY / 16;
    static constexpr int XB = RX * 128, YB = RY * 128, SB = XB + YB;
    static constexpr int NPX = RX / 32, NPY = RY / 32;   // 1-KiB pieces per wave
    static_assert(RX % 32 == 0 && RY % 32 == 0, "tile rows");

    float4v acc[NY][NX];

    FR_DEV void zero() {
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[y][x] = float4v{0.f, 0.f, 0.f, 0.f};
    }

    // Row (inside the tile) that lane `lane` of wave `wave` stages in piece i.
    FR_DEV static int stage_row(int wave, int lane, int i) { return 8 * (wave + 4 * i) + (lane >> 3); }
    // Byte offset inside a source row of the 16-B chunk this lane fetches.
    FR_DEV static int stage_chunk_bytes(int lane) { return (((lane & 7) ^ (lane >> 3)) << 4); }

    FR_DEV static void issue(char* stage, int wave, const char* const (&xp)[NPX], size_t xoff,
                             const char* const (&yp)[NPY], size_t yoff) {
#pragma unroll
        for (int i = 0; i < NPX; ++i) glds16(xp[i] + xoff, stage + (wave + 4 * i) * 1024);
#pragma unroll
        for (int i = 0; i < NPY; ++i) glds16(yp[i] + yoff, stage + XB + (wave + 4 * i) * 1024);
    }

    FR_DEV void compute(const char* stage, int lane, int wx, int wy) {
        const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = (((kk * 4 + g0) ^ l7) << 4);
            half8 fx[NX], fy[NY];
#pragma unroll
            for (int x = 0; x < NX; ++x)
                fx[x] = *(const half8*)(stage + (wx * TX + x * 16 + l15) * 128 + coff);
#pragma unroll
            for (int y = 0; y < NY; ++y)
                fy[y] = *(const half8*)(stage + XB + (wy * TY + y * 16 + l15) * 128 + coff);
#pragma unroll
            for (int y = 0; y < NY; ++y)
#pragma unroll
                for (int x = 0; x < NX; ++x) {
                    if (X_IS_A)
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], acc[y][x], 0, 0, 0);
                    else
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fy[y], fx[x], acc[y][x], 0, 0, 0);
                }
        }
    }
};

// ------------------------------------------------------------ conv1_1 ------
// 3 -> 64 channels.  Input fp16 [B,226,226,4] (4th channel zero), weights
// fp16 [64][64] with k = tap*4 + c for k < 36 and zeros after.  K is covered by two
// 16x16x32 MFMAs (taps 0..7, then tap 8 + zero padding).
// Each wave turns 16 consecutive pixels of a row into 16 x 64 outputs per step and
// stores them as two fully coalesced 1-KiB rows-of-pixels through a wave-private LDS
// transpose (the accumulator holds 4 channels x 1 pixel per lane; a store wants 8 lanes
// x 16 B per pixel).
__global__ __launch_bounds__(256) void conv1_1_kernel(const half_t* __restrict__ x0,
                                                      const half_t* __restrict__ w64,
                                                      const float* __restrict__ bias,
                                                      half_t* __restrict__ y, int ntiles) {
    constexpr int P = 226, T = 14;  // padded width, 16-pixel tiles per row
    constexpr int PITCH = 144;      // bytes per pixel row in LDS: 128 + 16 keeps ds_read_b128 aligned
    __shared__ __attribute__((aligned(16))) char tr[4][16 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    half8 a0[4], a1[4];
    float4v bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const half_t* wr = w64 + (ct * 16 + l15) * 64;
        a0[ct] = *(const half8*)(wr + 8 * g);
        a1[ct] = *(const half8*)(wr + 32 + 8 * g);
        bv[ct] = *(const float4v*)(bias + ct * 16 + g * 4);
    }
    const int t0 = 2 * g, t1 = 2 * g + 1;
    const int off0 = ((t0 / 3) * P + (t0 % 3)) * 4, off1 = ((t1 / 3) * P + (t1 % 3)) * 4;
    const int off8 = (2 * P + 2) * 4;
    char* const my = tr[wave];
    char* const wr_at = my + l15 * PITCH + g * 8;               // + ct*32
    const char* const rd_at = my + (lane >> 3) * PITCH + (lane & 7) * 16;   // pixel lane>>3 (+8), chunk lane&7
    const int nw = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        const int b = t / (224 * T), rem = t % (224 * T);
        const int py = rem / T, px0 = (rem % T) * 16;
        const half_t* src = x0 + (((long)b * P + py) * P + px0 + l15) * 4;
        const half4 v0 = *(const half4*)(src + off0);
        const half4 v1 = *(const half4*)(src + off1);
        const half4 v8 = *(const half4*)(src + off8);
        const half_t z = (half_t)0.f;
        const half8 bf0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const half8 bf1 = {v8[0], v8[1], v8[2], v8[3], z, z, z, z};   // weights are zero beyond k = 35
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float4v acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[ct], bf0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ct], bf1, acc, 0, 0, 0);
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (half_t)fmaxf(acc[j] + bv[ct][j], 0.f);
            *(half4*)(wr_at + ct * 32) = o;
        }
        __builtin_amdgcn_wave_barrier();
        const half8 r0 = *(const half8*)rd_at;
        const half8 r1 = *(const half8*)(rd_at + 8 * PITCH);
        __builtin_amdgcn_wave_barrier();
        // 16 output pixels are 2 KiB contiguous in the NHWC row: lane -> 16 B at lane*16 (+1 KiB)
        char* dst = (char*)(y + (((long)b * P + py + 1) * P + px0 + 1) * 64) + lane * 16;
        // 1.6 GB per 256 faces, read back only by the next layer: streamed past the caches
        // (non-temporal), which writes at 5.2 TB/s instead of 3.4 TB/s
        __builtin_nontemporal_store(r0, (half8*)dst);
        __builtin_nontemporal_store(r1, (half8*)(dst + 1024));
    }
}

// ------------------------------------------------------------- fc layers ---
// partial[z][m][n] over K slice z.  Rows of x beyond M are clamped on load and
// dropped on store.
template <int RX, int RY, int WX, int WY>
__global__ __launch_bounds__(256) void fc_splitk_kernel(const half_t* __restrict__ x,
                                                        const half_t* __restrict__ w,
                                                        float* __restrict__ partial, int M, int N, int K,
                                                        int ksteps_per_split) {
    using Core = GemmCore<RX, RY, WX, WY, false>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wx = wave / WY, wy = wave % WY;
    const int n0 = blockIdx.x * RY, m0 = blockIdx.y * RX, z = blockIdx.z;
    const int cb = Core::stage_chunk_bytes(lane);
    const char* xp[Core::NPX];
    const char* yp[Core::NPY];
    const size_t kbase = (size_t)z * ksteps_per_split * 128;
#pragma unroll
    for (int i = 0; i < Core::NPX; ++i) {
        int m = m0 + Core::stage_row(wave, lane, i);
        m = m < M ? m : M - 1;
        xp[i] = (const char*)x + (size_t)m * K * 2 + kbase + cb;
    }
#pragma unroll
    for (int i = 0; i < Core::NPY; ++i)
        yp[i] = (const char*)w + (size_t)(n0 + Core::stage_row(wave, lane, i)) * K * 2 + kbase + cb;

    Core core;
    core.zero();
    Core::issue(smem, wave, xp, 0, yp, 0);
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < ksteps_per_split - 1; ++ks) {
        Core::issue(smem + (cur ^ 1) * Core::SB, wave, xp, (size_t)(ks + 1) * 128, yp, (size_t)(ks + 1) * 128);
        core.compute(smem + cur * Core::SB, lane, wx, wy);
        __syncthreads();
        cur ^= 1;
    }
    core.compute(smem + cur * Core::SB, lane, wx, wy);

    const int l15 = lane & 15, g0 = lane >> 4;
#pragma unroll
    for (int px = 0; px < Core::NX; ++px) {
        const int m = m0 + wx * Core::TX + px * 16 + l15;
        if (m >= M) continue;
        float* prow = partial + ((size_t)z * M + m) * N;
#pragma unroll
        for (int cy = 0; cy < Core::NY; ++cy) {
            const int c = n0 + wy * Core::TY + cy * 16 + g0 * 4;
            *(float4v*)(prow + c) = core.acc[cy][px];
        }
    }
}

__global__ __launch_bounds__(256) void fc_reduce_relu_kernel(const float* __restrict__ partial,
                                                             const float* __restrict__ bias,
                                                             half_t* __restrict__ y, int M, int N, int splitk) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index
    const size_t total4 = (size_t)M * N / 4;
    if (i4 >= total4) return;
    const int n = (int)((i4 * 4) % N);
    float4v acc = *(const float4v*)(bias + n);
    for (int z = 0; z < splitk; ++z) acc += *(const float4v*)(partial + (size_t)z * M * N + i4 * 4);
    half4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (half_t)fmaxf(acc[j], 0.f);
    *(half4*)(y + i4 * 4) = o;
}

FR_DEV float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One block per batch row; N == 4096 (16 values per thread).
__global__ __launch_bounds__(256) void fc_reduce_relu_l2_kernel(const float* __restrict__ partial,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out_f32,
                                                                half_t* __restrict__ out_q, int Dq, int M,
                                                                int N, int splitk) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    float4v v[4];
    float ss = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = (i * 256 + tid) * 4;
        float4v acc = *(const float4v*)(bias + n);
        for (int z = 0; z < splitk; ++z) acc += *(const float4v*)(partial + ((size_t)z * M + m) * N + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = fmaxf(acc[j], 0.f);
            ss += acc[j] * acc[j];
            if (n + j < Dq) ssq += acc[j] * acc[j];
        }
        v[i] = acc;
    }
    const float tot = block_sum_256(ss, red);
    const float totq = block_sum_256(ssq, red);
    const float inv = tot > 0.f ? 1.0f / sqrtf(tot) : 1.0f;
    const float invq = totq > 0.f ? 1.0f / sqrtf(totq) : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = (i * 256 + tid) * 4;
        if (out_f32) *(float4v*)(out_f32 + (size_t)m * N + n) = v[i] * inv;
        if (out_q && n < Dq) {
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (half_t)(v[i][j] * invq);
            *(half4*)(out_q + (size_t)m * Dq + n) = o;
        }
    }
}

// ------------------------------------------------------------ preprocess ---
// Stage 1: gamma LUT + channel order + cv2-style INTER_LINEAR (11-bit fixed point)
// resize + centred zero pad, uint8 in -> uint8 [B,224,224,3] out, plus the per-image
// maximum byte (DeepFace divides by 255 only when it exceeds 1), plus stage 2's fp16
// network input for the images where it does.
// A thread produces four consecutive output pixels (12 bytes, one store).  The two source pixels
// of a row that a bilinear tap pair or a 2x2 box needs are neighbours, six consecutive bytes: one
// 8-byte load brings both (the load path moves a wave's request at the same pace whether a lane asks
// for one byte or eight, and twelve byte loads per output pixel were what bounded this kernel); at the
// end of a row the load is pulled back inside the row and the value shifted down instead.
FR_DEV void load_pixel_pair(const uint8_t* row, int sx, int rowbytes, bool same, unsigned& pa, unsigned& pb) {
    if (rowbytes >= 8) {
        const int o = min(3 * sx, rowbytes - 8);
        unsigned long long v;
        __builtin_memcpy(&v, row + o, 8);
        v >>= 8 * (3 * sx - o);
        pa = (unsigned)v & 0xffffffu;
        pb = same ? pa : (unsigned)(v >> 24) & 0xffffffu;
    } else {                                            // one or two pixels wide
        const uint8_t* p = row + 3 * sx;
        pa = p[0] | (p[1] << 8) | (p[2] << 16);
        pb = same ? pa : (unsigned)(p[3] | (p[4] << 8) | (p[5] << 16));
    }
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ crops,
                                                        const CropMeta* __restrict__ meta,
                                                        const uint8_t* __restrict__ lut_g, int flip,
                                                        uint8_t* __restrict__ out, int* __restrict__ out_max,
                                                        float scale, half_t* __restrict__ x0) {
    __shared__ uint8_t lut[256];
    __shared__ int red[4];
    // Stage 2 in the same pass: the network input of a byte value v in channel c is
    // (v / 255 * 255 - mean[c]) * scale, a table of 3 x 256 fp16 values built with the IEEE operations
    // normalize_pack_kernel uses.  DeepFace divides by 255 only when the image's largest byte exceeds
    // 1, which is known when the whole image is done: this pass assumes it does, and
    // normalize_pack_kernel then redoes the images where it does not (black ones).
    __shared__ half_t norm[3][256];
    const int b = blockIdx.y;
    lut[threadIdx.x] = lut_g[threadIdx.x];
    {
        const float mean[3] = {91.4953f, 103.8827f, 131.0912f};
        const float v = __fmul_rn(__fdiv_rn((float)threadIdx.x, 255.0f), 255.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) norm[c][threadIdx.x] = (half_t)__fmul_rn(__fsub_rn(v, mean[c]), scale);
    }
    // blockIdx.y is uniform, so these become scalar loads
    const long offset = meta[b].offset;
    const int sh = meta[b].h, sw = meta[b].w, stride = meta[b].stride;
    const int dw = meta[b].dw, dh = meta[b].dh, padx = meta[b].padx, pady = meta[b].pady;
    const int area2x = meta[b].area2x;
    const double scale_x = meta[b].scale_x, scale_y = meta[b].scale_y;
    __syncthreads();
    const uint8_t* src = crops + offset;
    const int sh0 = flip ? 16 : 0, sh2 = flip ? 0 : 16;     // bit offset of the source byte behind output channel 0 / 2
    int vmax = 0;
    // 1024 consecutive output pixels per block (49 blocks per image), four per thread; 224 is a
    // multiple of four, so a thread's pixels share a row
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int oy = p / 224, ox = p - oy * 224;
    const int dy = oy - pady;
    unsigned ob[3] = {0u, 0u, 0u};
    half4 xo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xo[j] = half4{norm[0][0], norm[1][0], norm[2][0], (half_t)0.f};     // padding pixels
    if (dy >= 0 && dy < dh) {
        const uint8_t *q0, *q1;
        int b0 = 0, b1 = 0;
        if (area2x) {
            q0 = src + (size_t)(2 * dy) * stride;
            q1 = q0 + stride;
        } else {
            float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy -= (float)sy;
            b0 = (int)rintf((1.0f - fy) * 2048.0f); b1 = (int)rintf(fy * 2048.0f);
            q0 = src + (size_t)min(max(sy, 0), sh - 1) * stride;
            q1 = src + (size_t)min(max(sy + 1, 0), sh - 1) * stride;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dx = ox + j - padx;
            int r0v = 0, r1v = 0, r2v = 0;
            if (dx >= 0 && dx < dw) {
                unsigned p00, p01, p10, p11;
                if (area2x) {
                    load_pixel_pair(q0, 2 * dx, 3 * sw, false, p00, p01);
                    load_pixel_pair(q1, 2 * dx, 3 * sw, false, p10, p11);
                    auto box = [&](int s) {
                        return (lut[(p00 >> s) & 0xff] + lut[(p01 >> s) & 0xff] + lut[(p10 >> s) & 0xff] +
                                lut[(p11 >> s) & 0xff] + 2) >> 2;
                    };
                    r0v = box(sh0); r1v = box(8); r2v = box(sh2);
                } else {
                    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
                    int sx = (int)floorf(fx);
                    fx -= (float)sx;
                    if (sx < 0) { fx = 0.f; sx = 0; }
                    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
                    const int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
                    const bool same = sx + 1 > sw - 1;          // the right tap is clamped onto the left one
                    load_pixel_pair(q0, sx, 3 * sw, same, p00, p01);
                    load_pixel_pair(q1, sx, 3 * sw, same, p10, p11);
                    auto mix = [&](int s) {
                        const int s0 = lut[(p00 >> s) & 0xff] * a0 + lut[(p01 >> s) & 0xff] * a1;
                        const int s1 = lut[(p10 >> s) & 0xff] * a0 + lut[(p11 >> s) & 0xff] * a1;
                        return (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
                    };
                    r0v = mix(sh0); r1v = mix(8); r2v = mix(sh2);
                }
            }
            // bytes 3j, 3j+1, 3j+2 of the thread's 12
            const unsigned long long three = (unsigned long long)(r0v | (r1v << 8) | (r2v << 16)) << (8 * (3 * j % 4));
            ob[3 * j / 4] |= (unsigned)three;
            if (3 * j % 4 > 1) ob[3 * j / 4 + 1] |= (unsigned)(three >> 32);
            vmax = max(vmax, max(r0v, max(r1v, r2v)));
            xo[j] = half4{norm[0][r0v], norm[1][r1v], norm[2][r2v], (half_t)0.f};
        }
    }
    half_t* xd = x0 + (((size_t)b * 226 + oy + 1) * 226 + ox + 1) * 4;     // 8-byte aligned, 16 from the second pixel on
    *(half4*)xd = xo[0];
    *(half8*)(xd + 4) = half8{xo[1][0], xo[1][1], xo[1][2], xo[1][3], xo[2][0], xo[2][1], xo[2][2], xo[2][3]};
    *(half4*)(xd + 12) = xo[3];
    uint8_t* o = out + ((size_t)b * 224 * 224 + p) * 3;         // 12 * (pixel group): dword aligned
    struct Out12 { unsigned a, b, c; };
    *(Out12*)o = Out12{ob[0], ob[1], ob[2]};
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) vmax = max(vmax, __shfl_xor(vmax, o2));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m > 0) atomicMax(out_max + b, m);
    }
}

// Stage 2: float conversion (x/255 when the image maximum exceeds 1, *255, minus the
// VGGFace2 channel means), input scale, fp16, into the interior of [B,226,226,4].
// only_undivided: resize_u8_kernel has written the input of every image under the assumption
// that its maximum exceeds 1; only the images where it does not are done (again) here.
__global__ __launch_bounds__(256) void normalize_pack_kernel(const uint8_t* __restrict__ u8,
                                                             const int* __restrict__ img_max, float scale,
                                                             half_t* __restrict__ x0, int only_undivided) {
    const int b = blockIdx.y;
    const bool div = img_max[b] > 1;
    if (div && only_undivided) return;          // resize_u8_kernel wrote this image's input already
    const float mean[3] = {91.4953f, 103.8827f, 131.0912f};
    // (a launch that only redoes the rare undivided images has a few blocks per image, which loop)
    for (int p = blockIdx.x * 256 + threadIdx.x; p < 224 * 224; p += gridDim.x * 256) {
        const uint8_t* s = u8 + ((size_t)b * 224 * 224 + p) * 3;
        half4 o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)s[c];
            if (div) v = __fdiv_rn(v, 255.0f);
            v = __fmul_rn(v, 255.0f);
            v = __fsub_rn(v, mean[c]);
            o[c] = (half_t)__fmul_rn(v, scale);
        }
        o[3] = (half_t)0.f;
        const int oy = p / 224, ox = p % 224;
        *(half4*)(x0 + (((size_t)b * 226 + oy + 1) * 226 + ox + 1) * 4) = o;
    }
}

// ------------------------------------------------------------- sharpness ---
// The check-in blur gate (app/src/utils/legacy.py:125-129): grey = cv2.cvtColor(BGR2GRAY) on uint8
// (15-bit fixed point: (3735 B + 19235 G + 9798 R + 16384) >> 15), cv2.Laplacian(CV_64F) with its
// default aperture (up + down + left + right - 4 centre, BORDER_REFLECT_101), then the variance
// over all pixels.  Every Laplacian value is an integer in [-1020, 1020], so its sum and sum of
// squares are exact in 64-bit integers: out[b] = {sum, sum of squares} (zero at launch); the host
// finishes the variance.
// A wave walks down a strip of image blockIdx.y, 256 pixels wide (four per lane, one 12-byte load
// per lane and row: the load path moves a wave's request at the same pace whether a lane asks for
// one byte or sixteen) and 16 rows tall, with the grey values of three rows in registers: every
// pixel is read once (plus two halo rows per strip), left and right neighbours come from the
// lanes beside (lanes 0 and 63 read theirs, which is also where the mirrored border falls out of
// the address), nothing goes through LDS.  Images narrower than four pixels take a plain path.
FR_DEV int grey_of(unsigned b, unsigned g, unsigned r) { return (int)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15); }

struct Px12 { unsigned d0, d1, d2; };

FR_DEV Px12 load12(const uint8_t* p) {
    Px12 v;
    __builtin_memcpy(&v, p, 12);
    return v;
}

FR_DEV unsigned load4(const uint8_t* p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Four pixels from 12 bytes, shifted down by `skip` whole pixels (the lane that reaches past the
// row's end loads the row's last 12 bytes instead and drops what belongs to its neighbour).
FR_DEV void grey4(const Px12 v, int skip, int g[4]) {
    const int sh = 24 * skip, ws = sh >> 5, s2 = sh & 31;       // 0, 24, 48, 72 bits = whole dwords + 0, 24, 16, 8 bits
    const unsigned w0 = ws == 0 ? v.d0 : (ws == 1 ? v.d1 : v.d2);
    const unsigned w1 = ws == 0 ? v.d1 : (ws == 1 ? v.d2 : 0u);
    const unsigned w2 = ws == 0 ? v.d2 : 0u;
    const unsigned e0 = (unsigned)((((unsigned long long)w1 << 32) | w0) >> s2);
    const unsigned e1 = (unsigned)((((unsigned long long)w2 << 32) | w1) >> s2);
    const unsigned e2 = w2 >> s2;
    // bytes 0..11 of the shifted run: e0 = bytes 0..3, e1 = bytes 4..7, e2 = bytes 8..11
    g[0] = grey_of(e0 & 0xff, (e0 >> 8) & 0xff, (e0 >> 16) & 0xff);
    g[1] = grey_of(e0 >> 24, e1 & 0xff, (e1 >> 8) & 0xff);
    g[2] = grey_of((e1 >> 16) & 0xff, e1 >> 24, e2 & 0xff);
    g[3] = grey_of((e2 >> 8) & 0xff, (e2 >> 16) & 0xff, e2 >> 24);
}

// One pixel from the dword that holds its three bytes (`sh`: their bit offset inside it).
FR_DEV int grey1(unsigned v, int sh) {
    v >>= sh;
    return grey_of(v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff);
}

__global__ __launch_bounds__(256) void crop_sharpness_kernel(const uint8_t* __restrict__ crops,
                                                             const CropMeta* __restrict__ meta,
                                                             unsigned long long* __restrict__ out) {
    constexpr int SW = 256, BH = 16, R = 8;
    __shared__ long long red[4][2];
    const int b = blockIdx.y;
    const long offset = meta[b].offset;
    const int h = meta[b].h, w = meta[b].w, stride = meta[b].stride;
    const uint8_t* src = crops + offset;
    // OpenCV's BORDER_REFLECT_101 for an index in [-1, n]; further out (never used) clamps
    auto reflect = [](int i, int n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
        return min(max(i, 0), n - 1);
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long sum = 0, ssq = 0;
    if (w < 4) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < h * w; i += gridDim.x * 256) {
            const int y = i / w, x = i - y * w;
            auto at = [&](int yy, int xx) {
                const uint8_t* p = src + (size_t)reflect(yy, h) * stride + 3 * reflect(xx, w);
                return grey_of(p[0], p[1], p[2]);
            };
            const int lap = at(y - 1, x) + at(y + 1, x) + at(y, x - 1) + at(y, x + 1) - 4 * at(y, x);
            sum += lap; ssq += lap * lap;
        }
    } else {
        const int strips = (w + SW - 1) / SW, nitems = strips * ((h + BH - 1) / BH);
        for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
            const int band = item / strips;
            const int x = (item - band * strips) * SW + 4 * lane, y0 = band * BH, y1 = min(y0 + BH, h);
            // this lane's 12 bytes inside a row, pulled back to the row's end where they would overrun it
            const int xs = min(x, w - 4), skip = min(x - xs, 3);
            const int o4 = 3 * xs;
            // the pixel beyond the strip (lane 0: left of its first, lane 63: right of its last)
            const bool edge = lane == 0 || lane == 63;
            const int xe = 3 * reflect(lane == 0 ? x - 1 : x + 4, w), oe = min(xe, 3 * w - 4), she = 8 * (xe - oe);
            const int last = w - 1 - x;                // which of the lane's pixels is the row's last one
            // R rows at a time, their loads issued before the first is used (the two rows that start
            // the band go out together with the first R)
            const Px12 ru = load12(src + (size_t)reflect(y0 - 1, h) * stride + o4);
            const Px12 rm = load12(src + (size_t)y0 * stride + o4);
            const unsigned rme = edge ? load4(src + (size_t)y0 * stride + oe) : 0u;
            int u[4], m[4], me = 0;
            int s = 0, q = 0;
            for (int y = y0; y < y1; y += R) {
                Px12 rd[R];
                unsigned rde[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint8_t* row = src + (size_t)reflect(y + r + 1, h) * stride;   // past the band: clamped, unused
                    rd[r] = load12(row + o4);
                    rde[r] = edge ? load4(row + oe) : 0u;
                }
                if (y == y0) {
                    grey4(ru, skip, u);
                    grey4(rm, skip, m);
                    me = grey1(rme, she);
                }
                int d[R][4], de[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    grey4(rd[r], skip, d[r]);
                    de[r] = grey1(rde[r], she);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    int left = __shfl_up(m[3], 1), right = __shfl_down(m[0], 1);
                    if (lane == 0) left = me;
                    if (lane == 63) right = me;
                    const bool live = y + r < y1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int lf = j ? m[j - 1] : left;
                        // the row's last pixel has the pixel to its left on both sides
                        const int rt = j == last ? lf : (j < 3 ? m[j + 1] : right);
                        const int lap = u[j] + d[r][j] + lf + rt - 4 * m[j];
                        if (live && j <= last) { s += lap; q += lap * lap; }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { u[j] = m[j]; m[j] = d[r][j]; }
                    me = de[r];
                }
            }
            sum += s; ssq += q;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); ssq += __shfl_xor(ssq, o); }
    if (lane == 0) { red[wave][0] = sum; red[wave][1] = ssq; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v != 0) atomicAdd(out + 2 * (size_t)b + threadIdx.x, (unsigned long long)v);
    }
}

// --------------------------------------------------------------- gallery ---
// Gallery rows are kept in HBM in the order the scan streams them: tile by tile (256 rows), and
// inside a tile 64-component chunk by chunk, so that one K-step's operand (256 rows x 64
// components = 32 KiB) is ONE contiguous block and every 1 KiB LDS-DMA piece reads 1 KiB of
// consecutive bytes.  Row-major, a piece gathered eight 128-B lines that lay D*2 bytes apart:
// at D = 4096 the 64-query scan streamed 5.3 TB/s that way and streams 6.6 TB/s this way.
//   component c of row r  ->  half index ((r / 256 * (D/64) + c / 64) * 256 + r % 256) * 64 + c % 64
// Only the kernels of this section and the scan know the layout; rows enter and leave the
// library row-major.
FR_DEV size_t gal_index(int64_t r, int c, int nk) {
    return ((((size_t)(r >> 8) * (size_t)nk + (size_t)(c >> 6)) << 8) + (size_t)(r & 255)) * 64 + (size_t)(c & 63);
}

// One wave per row: normalise an f32 row and store it as fp16.
__global__ __launch_bounds__(256) void gallery_upsert_kernel(const float* __restrict__ vecs,
                                                             const int32_t* __restrict__ keys,
                                                             const int64_t* __restrict__ rows, int n,
                                                             half_t* __restrict__ g, int32_t* __restrict__ gk,
                                                             int D) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* v = vecs + (size_t)i * D;
    // norm from a double-precision sum of squares, rounded to float, then one IEEE division per
    // element: the stored row is then the same whatever the summation order (what a float32
    // cosine collection keeps, rounded once more to fp16)
    double ss = 0.0;
    for (int c = lane; c < D; c += 64) ss += (double)v[c] * (double)v[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    float nrm = (float)sqrt(ss);
    nrm = nrm < 1e-30f ? 1.0f : nrm;
    const int64_t r = rows ? rows[i] : i;
    for (int c = lane; c < D; c += 64) g[gal_index(r, c, D >> 6)] = (half_t)__fdiv_rn(v[c], nrm);
    if (lane == 0 && gk) gk[r] = keys ? keys[i] : 0;
}

__global__ void gallery_set_keys_kernel(const int64_t* __restrict__ rows, int n, int32_t key,
                                        int32_t* __restrict__ gk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gk[rows[i]] = key;
}

FR_DEV int synth_value(uint64_t seed, uint64_t row, int D, int col) {
    const uint64_t h = splitmix64(seed + row * (uint64_t)D + (uint64_t)col);
    return (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) + (int)((h >> 24) & 255) - 510;
}

// Deterministic synthetic gallery: integer values from a hash, exact integer sum of
// squares, one IEEE sqrt + divide + multiply per element, so a numpy restatement
// reproduces every stored half exactly.
__global__ __launch_bounds__(256) void gallery_fill_kernel(half_t* __restrict__ g, int32_t* __restrict__ gk,
                                                           int D, uint64_t seed, int64_t row0, int64_t n,
                                                           int nkeys, int dup_every) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t r = row0 + i;
    const uint64_t src = (dup_every > 0 && r > 0 && (r % dup_every) == dup_every - 1) ? (uint64_t)(r - 1) : (uint64_t)r;
    unsigned long long ss = 0;
    for (int c = lane; c < D; c += 64) {
        const int v = synth_value(seed, src, D, c);
        ss += (unsigned long long)(v * v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = __fsqrt_rn((float)ss);
    const float inv = ss == 0 ? 1.0f : __fdiv_rn(1.0f, nrm);
    for (int c = lane; c < D; c += 64)
        g[gal_index(r, c, D >> 6)] = (half_t)__fmul_rn((float)synth_value(seed, src, D, c), inv);
    // keys: hashed over nkeys values, or (nkeys < 0) runs of -nkeys consecutive rows per key
    if (lane == 0)
        gk[r] = nkeys > 0   ? (int32_t)(splitmix64(seed ^ (src * 0xD1B54A32D192ED03ull)) % (uint64_t)nkeys)
                : nkeys < 0 ? (int32_t)(src / (uint64_t)(-(int64_t)nkeys))
                            : 0;
}

__global__ __launch_bounds__(256) void gallery_gather_kernel(const half_t* __restrict__ g,
                                                             const int32_t* __restrict__ gk,
                                                             const int64_t* __restrict__ rows, int n, int D,
                                                             half_t* __restrict__ out, int32_t* __restrict__ out_keys) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t r = rows[i];
    for (int c = lane; c < D; c += 64) out[(size_t)i * D + c] = g[gal_index(r, c, D >> 6)];
    if (lane == 0 && out_keys) out_keys[i] = gk[r];
}

// Inverse of the gather: rows of `in` ([n,D] fp16, already unit) and their keys go to rows[i].
__global__ __launch_bounds__(256) void gallery_scatter_kernel(const half_t* __restrict__ in,
                                                              const int32_t* __restrict__ in_keys,
                                                              const int64_t* __restrict__ rows, int n, int D,
                                                              half_t* __restrict__ g, int32_t* __restrict__ gk) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t r = rows[i];
    for (int c = lane * 8; c < D; c += 512) *(half8*)(g + gal_index(r, c, D >> 6)) = *(const half8*)(in + (size_t)i * D + c);
    if (lane == 0) gk[r] = in_keys[i];
}

// out[i] = in[index[i]] for n rows of D halfs (queries brought into key order).
__global__ __launch_bounds__(256) void rows_gather_f16_kernel(const half_t* __restrict__ in,
                                                              const int32_t* __restrict__ index, int n, int D,
                                                              half_t* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const size_t src = (size_t)index[i] * D, dst = (size_t)i * D;
    for (int c = lane * 8; c < D; c += 512) *(half8*)(out + dst + c) = *(const half8*)(in + src + c);
}

__global__ __launch_bounds__(256) void f32_to_unit_f16_kernel(const float* __restrict__ q, half_t* __restrict__ o,
                                                              int Bq, int D) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= Bq) return;
    const float* v = q + (size_t)i * D;
    double ss = 0.0;            // as gallery_upsert_kernel
    for (int c = lane; c < D; c += 64) ss += (double)v[c] * (double)v[c];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) ss += __shfl_xor(ss, o2);
    float nrm = (float)sqrt(ss);
    nrm = nrm < 1e-30f ? 1.0f : nrm;
    for (int c = lane; c < D; c += 64) o[(size_t)i * D + c] = (half_t)__fdiv_rn(v[c], nrm);
}

// ------------------------------------------------------------ scan+top-k ---
// max / min over the four lanes l, l^16, l^32, l^48 (the lanes that hold one MFMA output column),
// through the row-swap instructions: no LDS round trip
FR_DEV float lanes4_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
FR_DEV float lanes4_min(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// float -> unsigned with the same order (and back); key 0 is below every float's key
FR_DEV unsigned ordered_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
FR_DEV float ordered_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Unsigned max over aligned groups of N = 8 or 16 adjacent lanes, result in every lane of the
// group: a butterfly of DPP lane permutations (pairs, quads, mirrored halves, mirrored row).
template <int N>
FR_DEV unsigned group_umax(unsigned v) {
#define FR_DPP_MAX(ctrl)                                                                                  \
    { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false); v = v > o ? v : o; }
    FR_DPP_MAX(0xB1)        // quad_perm [1,0,3,2]
    FR_DPP_MAX(0x4E)        // quad_perm [2,3,0,1]
    FR_DPP_MAX(0x141)       // row_half_mirror: lane i <- lane 7 - i of its half row
    if (N == 16) FR_DPP_MAX(0x140)      // row_mirror: lane i <- lane 15 - i of its row
#undef FR_DPP_MAX
    return v;
}

// max(a, b, c) as one instruction.  Written as the instruction itself: through fmaxf the compiler
// first re-canonicalises every MFMA result (one more VALU operation per score).
FR_DEV float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int K>
struct TopK {
    float s[K];
    int r[K];
    FR_DEV void init() {
#pragma unroll
        for (int i = 0; i < K; ++i) { s[i] = -INFINITY; r[i] = -1; }
    }
    // Precondition: (v,row) beats the tail.  `tie` adds the ascending-row order
    // inside equal scores (needed when candidates do not arrive in row order).
    template <bool TIE>
    FR_DEV void push(float v, int row) {
        s[K - 1] = v; r[K - 1] = row;
#pragma unroll
        for (int i = K - 1; i > 0; --i) {
            const bool sw = s[i] > s[i - 1] || (TIE && s[i] == s[i - 1] && (unsigned)r[i] < (unsigned)r[i - 1]);
            const float ts = s[i - 1]; const int tr = r[i - 1];
            s[i - 1] = sw ? s[i] : ts; r[i - 1] = sw ? r[i] : tr;
            s[i] = sw ? ts : s[i];     r[i] = sw ? tr : r[i];
        }
    }
    FR_DEV bool beats_tail(float v, int row) const {
        return v > s[K - 1] || (v == s[K - 1] && (unsigned)row < (unsigned)r[K - 1]);
    }
};

// ------------------------------------------------- 8-wave ring GEMM core ---
// 512 threads = 8 waves arranged WX x WY over a tile of RX streamed rows by RY resident
// rows, 64 K-elements per step.  The streamed operand has a ring of three LDS stages
// (loads issued two steps ahead: HBM latency), the resident operand two stages (one step
// ahead: L2 latency).  Loads are LDS-DMA, waited for with counted s_waitcnt vmcnt so that
// the newest stage stays in flight across the step barrier.
template <int RX, int RY, int WX, int WY, bool X_IS_A>
struct GemmCore8 {
    static_assert(WX * WY == 8, "8 waves");
    static_assert(RX == 256, "the counted waits below assume 4 streamed loads per wave per step");
    static constexpr int TX = RX / WX, TY = RY / WY;
    static constexpr int NX = TX / 16, NY = TY / 16;
    static constexpr int XB = RX * 128, YB = RY * 128;
    static constexpr int XS = 3, YS = 2;
    static constexpr int LDS = XS * XB + YS * YB;
    static constexpr int NPX = RX / 64, NPY = RY / 64;
    static_assert(RY % 64 == 0 && TX % 16 == 0 && TY % 16 == 0, "tile shape");

    float4v acc[NY][NX];

    FR_DEV void zero() {
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[y][x] = float4v{0.f, 0.f, 0.f, 0.f};
    }
    FR_DEV static int stage_row(int wave, int lane, int i) { return 8 * (wave + 8 * i) + (lane >> 3); }
    FR_DEV static int stage_chunk_bytes(int lane) { return (((lane & 7) ^ (lane >> 3)) << 4); }

    FR_DEV static void issue_x(char* xbuf, int wave, const char* const (&xp)[NPX], size_t off) {
#pragma unroll
        for (int i = 0; i < NPX; ++i) glds16(xp[i] + off, xbuf + (wave + 8 * i) * 1024);
    }
    FR_DEV static void issue_y(char* ybuf, int wave, const char* const (&yp)[NPY], size_t off) {
#pragma unroll
        for (int i = 0; i < NPY; ++i) glds16(yp[i] + off, ybuf + (wave + 8 * i) * 1024);
    }
    // Wait until only the newest streamed stage (NPX loads) is still in flight / nothing is.
    FR_DEV static void wait_keep_newest() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    FR_DEV static void wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

    // Software-pipelined form of one K-step: four units u = (kk, xh) of NY * NX/2 MFMAs each
    // (kk = 32-deep half of the step, xh = half of the streamed rows).  The resident fragments
    // of a kk are read once and kept for both of its units; the streamed fragments alternate
    // between two sets, so unit u+1's reads are in flight while unit u's MFMAs issue.
    static constexpr int NXH = NX / 2;
    half8 pfy[2][NY], pfx[2][NXH];
    template <int U>
    FR_DEV void read_unit(const char* xbuf, const char* ybuf, int lane, int wx, int wy) {
        constexpr int kk = U >> 1, xh = U & 1;
        const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
        const int coff = (((kk * 4 + g0) ^ l7) << 4);
        if (xh == 0) {
#pragma unroll
            for (int y = 0; y < NY; ++y)
                pfy[kk][y] = *(const half8*)(ybuf + (wy * TY + y * 16 + l15) * 128 + coff);
        }
#pragma unroll
        for (int x = 0; x < NXH; ++x)
            pfx[xh][x] = *(const half8*)(xbuf + (wx * TX + (xh * NXH + x) * 16 + l15) * 128 + coff);
    }
    template <int U>
    FR_DEV void mma_unit() {
        constexpr int kk = U >> 1, xh = U & 1;
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int x = 0; x < NXH; ++x) {
                if (X_IS_A)
                    acc[y][xh * NXH + x] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(pfx[xh][x], pfy[kk][y], acc[y][xh * NXH + x], 0, 0, 0);
                else
                    acc[y][xh * NXH + x] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(pfy[kk][y], pfx[xh][x], acc[y][xh * NXH + x], 0, 0, 0);
            }
    }

    FR_DEV void compute(const char* xbuf, const char* ybuf, int lane, int wx, int wy) {
        const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = (((kk * 4 + g0) ^ l7) << 4);
            half8 fx[NX], fy[NY];
#pragma unroll
            for (int y = 0; y < NY; ++y)
                fy[y] = *(const half8*)(ybuf + (wy * TY + y * 16 + l15) * 128 + coff);
#pragma unroll
            for (int x = 0; x < NX; ++x)
                fx[x] = *(const half8*)(xbuf + (wx * TX + x * 16 + l15) * 128 + coff);
#pragma unroll
            for (int y = 0; y < NY; ++y)
#pragma unroll
                for (int x = 0; x < NX; ++x) {
                    if (X_IS_A)
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], acc[y][x], 0, 0, 0);
                    else
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fy[y], fx[x], acc[y][x], 0, 0, 0);
                }
        }
    }
};

// 3x3 convolution as a persistent implicit GEMM on the 8-wave ring core.  Block w walks
// output tiles w, w+G, ... (tile = 256 window-major pixels x RY channels, channel tile
// fastest); the (tile, k-step) sequence is one pipeline, so the next tile's operands are
// in flight while this tile's epilogue stores.  Every wave issues the same number of
// vector-memory instructions per step (dead pixels store to `dump`), which lets the end
// of a tile's last step leave its stores in flight behind a counted wait.
template <int HW, int CIN, int COUT, bool POOL, int OPAD, int RY, int WX, int WY>
__global__ __launch_bounds__(512) void conv3x3_ring_kernel(const half_t* __restrict__ x,
                                                           const half_t* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           half_t* __restrict__ y, half_t* __restrict__ dump,
                                                           int batch, int reverse) {
    using Core = GemmCore8<256, RY, WX, WY, false>;
    constexpr int H = HW, W = HW, PW = W + 2, HALF = W / 2;
    constexpr int K = 9 * CIN, CPK = CIN / 64, NK = 9 * CPK;
    constexpr int NT = COUT / RY;
    constexpr int NX = Core::NX, NY = Core::NY, NST = NX * NY / 2;
    static_assert(CIN % 64 == 0 && COUT % RY == 0 && NY % 2 == 0, "channel tiling");
    static_assert(4 + NST <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const ys = smem + Core::XS * Core::XB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave / WY, wy = wave % WY;
    const int l15 = lane & 15, g0 = lane >> 4;
    // pixel indices fit an int: the launcher bounds the input to 4 GiB of 32-bit byte offsets.
    // (As 64-bit values the divisions below cost registers that the pooled variants spilled.)
    const int M = batch * H * W;
    const long ntiles = (long)((M + 255) / 256) * NT;
    const int G = gridDim.x;
    const int wk = xcd_remap(blockIdx.x, G);
    const long my_tiles = wk < ntiles ? (ntiles - wk + G - 1) / G : 0;
    const long S = my_tiles * NK;
    if (S == 0) return;
    const int cb = Core::stage_chunk_bytes(lane);
    const int n0 = (wk % NT) * RY;     // G is a multiple of NT: the channel tile never changes

    // per-lane byte offsets into x and w (both far below 4 GiB for the layers this kernel serves);
    // the uniform part of an address stays in scalar registers
    unsigned xo[Core::NPX], yo[Core::NPY];
    // `reverse` walks the pixel tiles from the last image to the first: a layer then starts on
    // the part of its input that the previous layer wrote last and that the 256 MiB Infinity
    // Cache still holds (layers alternate direction)
    const int nmt = (M + 255) / 256;
    auto set_pixels = [&](long tile) {
        const int mt = (int)(tile / NT);
        const int m0 = (reverse ? nmt - 1 - mt : mt) * 256;
#pragma unroll
        for (int i = 0; i < Core::NPX; ++i) {
            int m = m0 + Core::stage_row(wave, lane, i);
            m = m < M ? m : M - 1;
            const int b = m / (H * W);
            const int rem = m % (H * W);
            const int win = rem >> 2, d = rem & 3;
            const int py = 2 * (win / HALF) + (d >> 1), px = 2 * (win % HALF) + (d & 1);
            xo[i] = (unsigned)(((((long)b * (H + 2) + py) * PW + px) * CIN) * 2 + cb);
        }
    };
    // LDS row r of the weight tile holds output channel n0 + chan(r), permuted so that the
    // accumulators of channel-tile pair (2u, 2u+1) give every lane 8 CONSECUTIVE channels:
    // row = wy*TY + cy*16 + g*4 + j  ->  channel wy*TY + (cy/2)*32 + g*8 + (cy&1)*4 + j.
#pragma unroll
    for (int i = 0; i < Core::NPY; ++i) {
        const int r = Core::stage_row(wave, lane, i);
        const int in = r % Core::TY, cy = in >> 4, g = (in >> 2) & 3, j = in & 3;
        const int ch = (r / Core::TY) * Core::TY + (cy >> 1) * 32 + g * 8 + (cy & 1) * 4 + j;
        yo[i] = (unsigned)(((long)(n0 + ch) * K) * 2 + cb);
    }

    // Bias of this lane's channels, fetched and waited for before the pipeline starts: an
    // ordinary load inside the loop would make the compiler drain the LDS-DMA queue.
    float4v bv[NY];
#pragma unroll
    for (int cy = 0; cy < NY; ++cy) {
        bv[cy] = *(const float4v*)(bias + n0 + wy * Core::TY + (cy >> 1) * 32 + g0 * 8 + (cy & 1) * 4);
        asm volatile("" ::"v"(bv[cy][0]), "v"(bv[cy][1]), "v"(bv[cy][2]), "v"(bv[cy][3]));
    }

    long itile = wk;                 // issue side: tile and tap position of the next streamed step
    int ir = 0, is = 0, icc = 0;
    set_pixels(itile);
    // K order: the nine taps of one 64-channel chunk run back to back (channel chunk outermost),
    // so the shifted re-reads of a pixel row's chunk come within nine steps of each other and
    // the XCD's L2 still holds them (with the taps outermost the re-reads were CIN/64 steps apart
    // and a round's working set exceeded the 4 MiB L2).
    auto next_x = [&](int slot) {
        const char* const base = (const char*)x + ((size_t)(ir * PW + is) * CIN + icc * 64) * 2;
#pragma unroll
        for (int i = 0; i < Core::NPX; ++i) {
            unsigned o = xo[i];
            asm volatile("" : "+v"(o));     // keeps the 64-bit sum out of loop-invariant registers
            glds16(base + o, xs + slot * Core::XB + (wave + 8 * i) * 1024);
        }
        if (++is == 3) {
            is = 0;
            if (++ir == 3) {
                ir = 0;
                if (++icc == CPK) { icc = 0; itile += G; set_pixels(itile < ntiles ? itile : ntiles - 1); }
            }
        }
    };
    int ytap = 0, ycc = 0;
    auto next_y = [&](int slot) {
        const char* const base = (const char*)w + ((size_t)ytap * CIN + ycc * 64) * 2;
#pragma unroll
        for (int i = 0; i < Core::NPY; ++i) {
            unsigned o = yo[i];
            asm volatile("" : "+v"(o));
            glds16(base + o, ys + slot * Core::YB + (wave + 8 * i) * 1024);
        }
        if (++ytap == 9) { ytap = 0; if (++ycc == CPK) ycc = 0; }
    };

    Core core;
    core.zero();
    next_y(0);
    next_x(0);
    if (S > 1) { next_x(1); Core::wait_keep_newest(); } else { Core::wait_all(); }
    __builtin_amdgcn_s_barrier();

    constexpr int HO = POOL ? H / 2 : H, WO = POOL ? W / 2 : W;
    constexpr int OW = WO + 2 * OPAD, OH = HO + 2 * OPAD;
    int xslot = 0, yslot = 0, ks = 0;
    long tile = wk;
    // epilogue of the tile whose last step has just been accumulated
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int mt = (int)(tile / NT);
        const int m0 = (reverse ? nmt - 1 - mt : mt) * 256;
#pragma unroll
        for (int px = 0; px < NX; ++px) {
            const int m = m0 + wx * Core::TX + px * 16 + l15;
            const bool live = m < M;
            const int mm = live ? m : M - 1;
            const int b = mm / (H * W);
            const int rem = mm % (H * W);
            const int win = rem >> 2, d = rem & 3;
            int oy, ox;
            if (POOL) { oy = win / HALF; ox = win % HALF; }
            else { oy = 2 * (win / HALF) + (d >> 1); ox = 2 * (win % HALF) + (d & 1); }
            half_t* orow = y + (((long)b * OH + oy + OPAD) * OW + ox + OPAD) * COUT;
#pragma unroll
            for (int u = 0; u < NY / 2; ++u) {
                const int c = n0 + wy * Core::TY + u * 32 + g0 * 8;
                half8 o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4v v = core.acc[2 * u + h][px];
                    if (POOL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // max over the 2x2 window = the four lanes of a quad (row-local DPP; a
                            // shuffle intrinsic compiles to ds_bpermute here, 512 LDS-path ops per tile)
                            float t = v[j];
                            int q = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true);
                            t = fmaxf(t, __builtin_bit_cast(float, q));
                            q = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true);
                            v[j] = fmaxf(t, __builtin_bit_cast(float, q));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * h + j] = (half_t)fmaxf(v[j] + bv[2 * u + h][j], 0.f);
                }
                // Every wave issues this store (dead pixels go to `dump`): the counted wait
                // below relies on a fixed number of vector-memory ops per step.  Pooled
                // layers store from the quad leader only; its EXEC is never empty.
                half_t* dst = live ? orow + c : dump + lane * 8;
                if (!POOL || (lane & 3) == 0) *(half8*)dst = o;
            }
        }
        core.zero();
        tile += G;
    };
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    auto reads_done = [&]() __attribute__((always_inline)) {
        fence(); __builtin_amdgcn_s_waitcnt(0xC07F); fence();       // lgkmcnt(0)
    };
    // Pipeline across the step barrier: a step's unit 3 is issued after the barrier that opens the
    // next step, behind that step's first fragment reads, so the MFMA pipe has work while the reads
    // are in flight.  A tile's epilogue follows its last unit 3, at the head of the next step; its
    // stores are older than every load issued afterwards, so the counted wait at the step's end
    // (only the newest streamed stage stays in flight) also retires them.  The step's eight LDS-DMA
    // instructions per wave are issued together at its head: the CU's vector-memory path accepts one
    // 1 KiB instruction per ~16 cycles, and spreading them between the MFMA units (two per unit)
    // or staggering the two wave groups measured 3-6 % slower.
#if FR_DBG & 32
    unsigned long long dsum[7] = {0, 0, 0, 0, 0, 0, 0}, tprev;
    FR_STAMP(tprev);
#endif
    for (long s = 0; s < S; ++s) {
        const char* const xbuf = xs + xslot * Core::XB;
        const char* const ybuf = ys + yslot * Core::YB;
        core.template read_unit<0>(xbuf, ybuf, lane, wx, wy);
        fence();
        if (s > 0) {
            core.template mma_unit<3>();
            fence();
            if (ks == NK) { epilogue(); ks = 0; }
        }
        if (s + 1 < S) next_y(yslot ^ 1);
        if (s + 2 < S) next_x(xslot == 0 ? 2 : xslot - 1);
        reads_done();
#if FR_DBG & 32
        unsigned long long ta, tb, tc, td, te, tf;
        FR_STAMP(ta);
#endif
        core.template read_unit<1>(xbuf, ybuf, lane, wx, wy);
        fence();
        core.template mma_unit<0>();
        reads_done();
#if FR_DBG & 32
        FR_STAMP(tb);
#endif
        core.template read_unit<2>(xbuf, ybuf, lane, wx, wy);
        fence();
        core.template mma_unit<1>();
        reads_done();
#if FR_DBG & 32
        FR_STAMP(tc);
#endif
        core.template read_unit<3>(xbuf, ybuf, lane, wx, wy);
        fence();
        core.template mma_unit<2>();
        ++ks;
        reads_done();      // every read of this step's stages has returned: the stages may be refilled
#if FR_DBG & 32
        FR_STAMP(td);
#endif
        if (s + 2 < S) Core::wait_keep_newest();
        else Core::wait_all();
#if FR_DBG & 32
        FR_STAMP(te);
#endif
        __builtin_amdgcn_s_barrier();
        fence();
#if FR_DBG & 32
        FR_STAMP(tf);
        dsum[0] += ta - tprev; dsum[1] += tb - ta; dsum[2] += tc - tb; dsum[3] += td - tc; dsum[4] += te - td;
        dsum[5] += tf - te; dsum[6] += 1; tprev = tf;
#endif
        xslot = xslot == 2 ? 0 : xslot + 1;
        yslot ^= 1;
    }
    core.template mma_unit<3>();
    epilogue();
#if FR_DBG & 32
    if (blockIdx.x == 8 && lane == 0)
        for (int i = 0; i < 7; ++i) g_stamp[wave * 8 + i] = dsum[i];
#endif
}

// 3x3 convolution with the input PATCH resident in LDS (layers whose height is a multiple
// of 16).  A tile is 16x16 output pixels of one image by RY channels.  Its 18x18-pixel input
// patch (one 64-channel chunk: 324 x 128 B, 41 one-KiB pieces) is fetched ONCE and read by
// all nine taps, so the LDS fill traffic per tile is 1.27 x the tile's input instead of 9 x.
// Item = (tile, 64-channel chunk), nine steps (taps) each; while an item computes, the next
// item's patch streams into the other of two patch slots (one piece per wave in taps 0..5).
// Weights: a ring of four RY x 128-B stages fetched three steps ahead, or, when all 9*CIN/64
// stages fit beside the patches (CIN = COUT = 64), loaded once and kept (RES).
//
// Patch image in LDS: pixel P = py*18 + px owns the 128-B row P; 16-B chunk c of that row is
// stored at chunk position c ^ sigma(px,py), sigma = ((px>>1)&3)<<1 | (py&1).  An MFMA B
// fragment covers a 2x8-pixel block (lane&15 -> dy = (l>>1)&1, dx = 2*(l>>2) + (l&1)), and with
// this sigma every 16-lane group of a ds_read_b128 touches 16 different 16-B slots of the
// 256-B bank row for every tap shift.
//
// FUSE1 (the 224 x 224 layer after the 3-channel one): the patch is not fetched but COMPUTED.
// x is then the network input [B,226,226,4] and w1/b1 the first layer's [64][64] weights and
// bias as conv1_1_kernel takes them.  The 324 patch pixels are 21 groups of 16 (2 x 8 blocks,
// and the last two columns); wave v makes groups v, v+8, v+16 of the NEXT tile's patch while
// the current tile computes: the three 8-byte tap loads of a group (its 3 x 3 windows of the
// 20 x 20 x 4 input) are issued in the L phase of taps 1/3/5 and its eight MFMAs
// (K = 36 in two 16x16x32, as in conv1_1_kernel), bias, ReLU, fp16 and four ds_write_b64 into
// the swizzled patch rows run in the M phase of taps 2/4/6.  Patch pixels outside the image
// are the layer's zero padding, not first-layer outputs: they are written as zeros.  The
// values are the ones conv1_1_kernel stores (same MFMA order), so the layer's output is
// bit-identical to the two-kernel path; the 6.4 MB per face between the layers never exists.
template <int HW, int CIN, int COUT, bool POOL, int RY, int WX, int WY, bool FUSE1 = false>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const half_t* __restrict__ x,
                                                            const half_t* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            half_t* __restrict__ y, half_t* __restrict__ dump,
                                                            int batch, int reverse,
                                                            const half_t* __restrict__ w1 = nullptr,
                                                            const float* __restrict__ b1 = nullptr) {
    static_assert(WX * WY == 8 && HW % 16 == 0 && CIN % 64 == 0 && COUT % RY == 0 && RY % 64 == 0, "shape");
    constexpr int PW = 18, NPIECE = 41, PATCH_B = NPIECE * 1024;
    constexpr int CPK = CIN / 64, NKS = 9 * CPK, K = 9 * CIN;
    constexpr int NT = COUT / RY;
    constexpr int TX = 256 / WX, TY = RY / WY, NX = TX / 16, NY = TY / 16;
    constexpr int YB = RY * 128, NPY = RY / 64;
    constexpr bool RES = (NKS * YB + 2 * PATCH_B) <= 160 * 1024;
    constexpr int TPR = HW / 16, TPI = TPR * TPR;   // tiles per row / per image
    constexpr int IW = HW + 2;
    constexpr int NCH = NX * NY / 2;                // epilogue chunks (one 16-B store each) per wave and tile
    // small wave tiles keep the finished tile in a second accumulator set and spread its epilogue
    // over the next tile's MFMA phases; large ones (no registers to spare) run it at the tile's end
    constexpr bool DEFER = NCH <= 4;
    static_assert(NX % 2 == 0 && NY % 2 == 0 && NCH <= 18, "wave tile");
    static_assert(2 * (NPY + 1) + 6 <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    char* const wring = smem + 2 * PATCH_B;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave / WY, wy = wave % WY;
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
    const long ntiles = (long)batch * TPI * NT;
    const int G = gridDim.x;
    const int wk = xcd_remap(blockIdx.x, G);
    const long my_tiles = wk < ntiles ? (ntiles - wk + G - 1) / G : 0;
    if (my_tiles == 0) return;
    const long n_items = my_tiles * CPK;
    const long S = n_items * 9;
    const int n0 = (wk % NT) * RY;       // G is a multiple of NT: the channel tile never changes

    // ---- staging addresses -------------------------------------------------------------
    // patch piece j = min(8*t + wave, 40) in tap t < 6; this lane fetches 16 B of pixel 8j + lane/8
    int poff[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        int j = 8 * t + wave;
        j = j < NPIECE ? j : NPIECE - 1;
        int P = 8 * j + (lane >> 3);
        P = P < PW * PW ? P : PW * PW - 1;
        const int py = P / PW, px = P - py * PW;
        const int sig = (((px >> 1) & 3) << 1) | (py & 1);
        poff[t] = ((py * IW + px) * CIN) * 2 + ((l7 ^ sig) << 4);
    }
    auto item_src = [&](long item) __attribute__((always_inline)) -> const char* {
        const long tile = wk + (item / CPK) * G;
        const int c = (int)(item % CPK);
        const long st = reverse ? (long)batch * TPI - 1 - tile / NT : tile / NT;   // see conv3x3_ring_kernel
        const int b = (int)(st / TPI), rem = (int)(st % TPI);
        const int ty = rem / TPR, tx = rem - ty * TPR;
        if (FUSE1) return (const char*)x + (((long)b * IW + ty * 16) * IW + tx * 16) * 8;      // 4 channels
        return (const char*)x + ((((long)b * IW + ty * 16) * IW + tx * 16) * CIN + c * 64) * 2;
    };
    // which edges of the image the item's tile touches: top, bottom, left, right
    auto item_edges = [&](long item) __attribute__((always_inline)) -> int {
        const long tile = wk + (item / CPK) * G;
        const long st = reverse ? (long)batch * TPI - 1 - tile / NT : tile / NT;
        const int rem = (int)(st % TPI);
        const int ty = rem / TPR, tx = rem - ty * TPR;
        return (ty == 0 ? 1 : 0) | (ty == TPR - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == TPR - 1 ? 8 : 0);
    };
    auto issue_patch = [&](const char* src, int slot, int t) __attribute__((always_inline)) {
        int j = 8 * t + wave;
        j = j < NPIECE ? j : NPIECE - 1;
        glds16(src + poff[t], patch + slot * PATCH_B + j * 1024);
    };
    // weight rows: LDS row r of a stage holds channel n0 + chan(r), permuted as in the ring
    // kernel so that a lane's accumulators of a channel-tile pair are 8 consecutive channels
    const int cb = ((l7 ^ (lane >> 3)) << 4);
    const char* yp[NPY];
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
        const int r = 8 * (wave + 8 * i) + (lane >> 3);
        const int in = r % TY, cy = in >> 4, g = (in >> 2) & 3, j = in & 3;
        const int ch = (r / TY) * TY + (cy >> 1) * 32 + g * 8 + (cy & 1) * 4 + j;
        yp[i] = (const char*)w + ((long)(n0 + ch) * K) * 2 + cb;
    }
    auto issue_w = [&](int stage, int ks) __attribute__((always_inline)) {     // ks = chunk*9 + tap
        const int c = ks / 9, t = ks - c * 9;
        const size_t off = ((size_t)t * CIN + c * 64) * 2;
#pragma unroll
        for (int i = 0; i < NPY; ++i) glds16(yp[i] + off, wring + stage * YB + (wave + 8 * i) * 1024);
    };

    float4v bv[NY];
#pragma unroll
    for (int cy = 0; cy < NY; ++cy) {
        bv[cy] = *(const float4v*)(bias + n0 + wy * TY + (cy >> 1) * 32 + g0 * 8 + (cy & 1) * 4);
        asm volatile("" ::"v"(bv[cy][0]), "v"(bv[cy][1]), "v"(bv[cy][2]), "v"(bv[cy][3]));
    }

    // ---- first-layer producer (FUSE1) ---------------------------------------------------
    constexpr int NG = (PW * PW + 15) / 16;         // 16-pixel groups of a patch
    half8 pa0[4], pa1[4];                           // first-layer weights: A fragments of the two K blocks
    float4v pbv[4];
    int prel[3], pdst[3], pedge[3];                 // per group: source offset, patch row address, edges the pixel lies beyond
    int ptap0 = 0, ptap1 = 0;
    half4 pv0 = {}, pv1 = {}, pv8 = {};             // the group in flight: taps 2g, 2g+1 and 8 of this lane's pixel
    unsigned pkeep = 0;
    if constexpr (FUSE1) {
        static_assert(HW == 224 && CIN == 64 && NT == 1 && CPK == 1, "the fused first layer feeds the 224 x 224 x 64 layer");
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const half_t* wr = w1 + (ct * 16 + l15) * 64;
            pa0[ct] = *(const half8*)(wr + 8 * g0);
            pa1[ct] = *(const half8*)(wr + 32 + 8 * g0);
            pbv[ct] = *(const float4v*)(b1 + ct * 16 + g0 * 4);
        }
        const int t0 = 2 * g0, t1 = 2 * g0 + 1;
        ptap0 = ((t0 / 3) * IW + (t0 % 3)) * 8;
        ptap1 = ((t1 / 3) * IW + (t1 % 3)) * 8;
        // A group's 16 pixels: groups 0..17 are the 2 x 8 blocks of columns 0..15 (the 16 lanes of a k-group then
        // write 16 different 16-byte slots of the 256-byte bank row: pixel parity = px & 1, sigma from px >> 1 and
        // py & 1), groups 18..20 the two last columns, eight rows each (the last one two rows, written twice over).
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int gi = wave + 8 * i;
            gi = gi < NG ? gi : NG - 1;
            int py, px;
            if (gi < 18) { py = 2 * (gi >> 1) + (l15 >> 3); px = 8 * (gi & 1) + (l15 & 7); }
            else { py = 8 * (gi - 18) + (l15 >> 1); px = 16 + (l15 & 1); if (py >= PW) py = 16 + ((l15 >> 1) & 1); }
            const int P = py * PW + px;
            const int sig = (((px >> 1) & 3) << 1) | (py & 1);
            prel[i] = ((py - 1) * IW + (px - 1)) * 8;       // the pixel's 3 x 3 window starts one up and left of it
            pedge[i] = (py == 0 ? 1 : 0) | (py == PW - 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == PW - 1 ? 8 : 0);
            pdst[i] = P * 128 + ((((g0 >> 1) ^ sig) << 4) | ((g0 & 1) << 3));
        }
    }
    // loads of group i of the patch of the tile at `base` (its edges: `edges`)
    auto prod_load = [&](auto ic, const char* base, int edges) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if (i == 2 && wave + 16 >= NG) return;
        const bool outside = (pedge[i] & edges) != 0;
        const char* p = base + (outside ? 0 : prel[i]);     // a padding pixel reads the tile's own corner and is zeroed below
        pv0 = *(const half4*)(p + ptap0);
        pv1 = *(const half4*)(p + ptap1);
        pv8 = *(const half4*)(p + (2 * IW + 2) * 8);
        pkeep = outside ? 0u : 0xffffffffu;
    };
    auto prod_compute = [&](auto ic, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if (i == 2 && wave + 16 >= NG) return;
        const half_t z = (half_t)0.f;
        const half8 bf0 = {pv0[0], pv0[1], pv0[2], pv0[3], pv1[0], pv1[1], pv1[2], pv1[3]};
        const half8 bf1 = {pv8[0], pv8[1], pv8[2], pv8[3], z, z, z, z};     // the weights are zero beyond k = 35
        float4v pacc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa0[ct], bf0, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa1[ct], bf1, pacc[ct], 0, 0, 0);
        char* const row = patch + slot * PATCH_B;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (half_t)fmaxf(pacc[ct][j] + pbv[ct][j], 0.f);
            uint2 u = __builtin_bit_cast(uint2, o);
            u.x &= pkeep; u.y &= pkeep;
            *(uint2*)(row + (pdst[i] ^ (ct << 5))) = u;         // channels 16 ct + 4 g0 ..: chunk (2 ct + g0/2) ^ sigma
        }
    };

    // ---- fragment read addresses --------------------------------------------------------
    // pixel of this lane inside a 2x8 block, block a = wx*NX + xi at rows 2*(a>>1), columns 8*(a&1)
    const int dy = (l15 >> 1) & 1, dx = 2 * (l15 >> 2) + (l15 & 1);
    const int lane_px = ((wx * NX / 2) * 2 * PW + dy * PW + dx) * 128;
    // sigma of (px0 + s, py0 + r): column part per s, row part dy ^ (r & 1)
    int sgx[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) sgx[s3] = (((dx + s3) >> 1) & 3) << 5;
    const int wrow = (wy * TY + l15) * 128;

    // acc: the tile being accumulated; accp: the finished tile before it, whose epilogue
    // (bias, ReLU, pooling, fp16, store) is spread over the MFMA blocks of the current tile so
    // that its vector work hides behind them instead of stalling every wave at the tile seam.
    float4v acc[NY][NX], accp[DEFER ? NY : 1][DEFER ? NX : 1];
    auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NY; ++a)
#pragma unroll
            for (int b2 = 0; b2 < NX; ++b2) acc[a][b2] = float4v{0.f, 0.f, 0.f, 0.f};
    };
    zero();
    constexpr int HO = POOL ? HW / 2 : HW, OW = HO + 2;
    half_t* pend = dump + lane * 8;     // where chunk stores of the pending tile go; dump: no tile pending
    long pend_stride_on = 0;            // 1 once a real tile is pending (0 keeps every chunk inside dump)
    auto tile_out = [&](long tile) __attribute__((always_inline)) -> half_t* {
        const long st = reverse ? (long)batch * TPI - 1 - tile / NT : tile / NT;
        const int b = (int)(st / TPI), rem = (int)(st % TPI);
        const int ty = rem / TPR, tx = rem - ty * TPR;
        const int a0 = wx * NX;
        int oy, ox;
        if (POOL) { oy = ty * 8 + (a0 >> 1); ox = tx * 8 + (l15 >> 2); }
        else { oy = ty * 16 + 2 * (a0 >> 1) + dy; ox = tx * 16 + dx; }
        return y + (((long)b * OW + oy + 1) * OW + ox + 1) * COUT + n0 + wy * TY + g0 * 8;
    };
    auto quad_max = [&](float v) __attribute__((always_inline)) -> float {
        // max over the 2x2 window = the four lanes of a quad (row-local DPP, no LDS traffic)
        int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);   // lanes ^1
        v = fmaxf(v, __builtin_bit_cast(float, t));
        t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);       // lanes ^2
        return fmaxf(v, __builtin_bit_cast(float, t));
    };
    auto epi_chunk = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NCH) {
            constexpr int b2 = j / (NY / 2), u = j % (NY / 2);
            constexpr long boff = POOL ? ((long)(b2 >> 1) * OW + 4 * (b2 & 1)) * COUT
                                       : ((long)2 * (b2 >> 1) * OW + 8 * (b2 & 1)) * COUT;
            half8 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4v v = DEFER ? accp[DEFER ? 2 * u + h : 0][DEFER ? b2 : 0] : acc[2 * u + h][b2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float t2 = POOL ? quad_max(v[jj]) : v[jj];
                    o[4 * h + jj] = (half_t)fmaxf(t2 + bv[2 * u + h][jj], 0.f);
                }
            }
#if !(FR_DBG & 1)
#ifdef FR_NT_STORE2
            if (!POOL || (lane & 3) == 0) __builtin_nontemporal_store(o, (half8*)(pend + (boff + u * 32) * pend_stride_on));
#else
            if (!POOL || (lane & 3) == 0) *(half8*)(pend + (boff + u * 32) * pend_stride_on) = o;
#endif
#else
            { half_t* pp = pend; asm volatile("" ::"v"(o), "v"(pp)); }
#endif
        }
    };

    // ---- ping-pong schedule ------------------------------------------------------------------
    // Waves 0..3 (group A) and 4..7 (group B) share the SIMDs pairwise and run one phase apart,
    // every phase ending in a workgroup barrier: while one group issues the step's LDS-DMA and
    // reads its fragments (L), the other runs its MFMAs (M), so a SIMD's matrix pipe always has
    // exactly one wave feeding it and the other wave's memory work hides behind that.
    //   phase 2s  :  A: L(s)      B: M(s-1)
    //   phase 2s+1:  A: M(s)      B: L(s)
    const bool isA = wave < 4;
    // a phase handles U of the step's two 32-deep K blocks: 16 MFMAs per wave either way
    constexpr int U = (NX * NY >= 16) ? 1 : 2, NU = 2 / U;
    half8 fx[U][NX], fy[U][NY];
    static_assert(!FUSE1 || (RES && NU == 1 && DEFER), "the fused producer's place in the schedule");
    auto phase_L = [&](auto tapc, auto uc, int chunk, long s, int slot, const char* nsrc, int nedges) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value, u = decltype(uc)::value;
        constexpr int r = tap / 3, sx = tap - 3 * r;
        // the step's LDS-DMA issues are spread over its L phases (an issue costs the wave tens of
        // cycles, and an L phase has to fit beside the partner's 16 MFMAs)
#if !(FR_DBG & 16) && !defined(FR_W_IN_M)
        if (u == 0) {
            if (!RES) {
                if (s + 3 < S) issue_w((int)((s + 3) & 3), (int)((s + 3) % NKS));
            }
        }
#endif
#if !(FR_DBG & 2)
        if (u == NU - 1) {
            if constexpr (FUSE1) {
                if (tap == 1) prod_load(std::integral_constant<int, 0>{}, nsrc, nedges);
                if (tap == 3) prod_load(std::integral_constant<int, 1>{}, nsrc, nedges);
                if (tap == 5) prod_load(std::integral_constant<int, 2>{}, nsrc, nedges);
            } else {
                if (tap < 6) issue_patch(nsrc, slot ^ 1, tap);
            }
        }
#endif
        const char* const pb = patch + slot * PATCH_B;
        const char* const wb = wring + (RES ? (chunk * 9 + tap) : (int)(s & 3)) * YB;
#pragma unroll
        for (int k2 = 0; k2 < U; ++k2) {
            const int kk = u * U + k2;
            const int wc = (((kk * 4 + g0) ^ l7) << 4);
#pragma unroll
            for (int a = 0; a < NY; ++a) fy[k2][a] = *(const half8*)(wb + wrow + a * 16 * 128 + wc);
            const int pc = ((kk * 64 + (g0 << 4)) ^ sgx[sx] ^ ((dy ^ (r & 1)) << 4));
#pragma unroll
            for (int b2 = 0; b2 < NX; ++b2) {
                const int blk = (((b2 >> 1) * 2 + r) * PW + 8 * (b2 & 1) + sx) * 128;
                fx[k2][b2] = *(const half8*)(pb + lane_px + blk + pc);
            }
        }
    };
    auto phase_M = [&](auto tapc, auto uc, int chunk, long tile, long s, int slot) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value, u = decltype(uc)::value;
#if !(FR_DBG & 2)
        if constexpr (FUSE1) {
            // (group B's M phase of tap 8 runs beside group A's first reads of the new patch: nothing after tap 7)
            if (tap == 2) prod_compute(std::integral_constant<int, 0>{}, slot ^ 1);
            if (tap == 4) prod_compute(std::integral_constant<int, 1>{}, slot ^ 1);
            if (tap == 6) prod_compute(std::integral_constant<int, 2>{}, slot ^ 1);
        }
#endif
#if defined(FR_W_IN_M)
        if (u == 0 && NU == 2) {
            if (!RES) {
                if (s + 3 < S) issue_w((int)((s + 3) & 3), (int)((s + 3) % NKS));
            }
        }
#endif
#if !(FR_DBG & 4)
#pragma unroll
        for (int k2 = 0; k2 < U; ++k2)
#pragma unroll
            for (int a = 0; a < NY; ++a)
#pragma unroll
                for (int b2 = 0; b2 < NX; ++b2)
                    acc[a][b2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fy[k2][a], fx[k2][b2], acc[a][b2], 0, 0, 0);
#endif
        if constexpr (FUSE1) {
            // the patch rows written above must be in LDS before the barrier that ends this phase
            if (tap == 2 || tap == 4 || tap == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (u != NU - 1) return;
        if (DEFER) {
            if (chunk == 0) {
                epi_chunk(std::integral_constant<int, 2 * tap>{});
                epi_chunk(std::integral_constant<int, 2 * tap + 1>{});
            }
            if (tap == 8 && chunk == CPK - 1) {
#pragma unroll
                for (int a = 0; a < NY; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < NX; ++b2) accp[DEFER ? a : 0][DEFER ? b2 : 0] = acc[a][b2];
                pend = tile_out(tile);
                pend_stride_on = 1;
                zero();
            }
        } else if (tap == 8 && chunk == CPK - 1) {
            pend = tile_out(tile);
            pend_stride_on = 1;
            epi_chunk(std::integral_constant<int, 0>{}); epi_chunk(std::integral_constant<int, 1>{});
            epi_chunk(std::integral_constant<int, 2>{}); epi_chunk(std::integral_constant<int, 3>{});
            epi_chunk(std::integral_constant<int, 4>{}); epi_chunk(std::integral_constant<int, 5>{});
            epi_chunk(std::integral_constant<int, 6>{}); epi_chunk(std::integral_constant<int, 7>{});
            zero();
        }
    };

    // ---- prologue -------------------------------------------------------------------------
    {
        const char* src0 = item_src(0);
        if constexpr (FUSE1) {
            const int e0 = item_edges(0);
            prod_load(std::integral_constant<int, 0>{}, src0, e0); prod_compute(std::integral_constant<int, 0>{}, 0);
            prod_load(std::integral_constant<int, 1>{}, src0, e0); prod_compute(std::integral_constant<int, 1>{}, 0);
            prod_load(std::integral_constant<int, 2>{}, src0, e0); prod_compute(std::integral_constant<int, 2>{}, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t) issue_patch(src0, 0, t);
        }
        if (RES) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) issue_w(ks, ks);
        } else {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) issue_w(ks, ks % NKS);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // Both groups run the SAME instruction stream, one unit after the other as L, barrier, M,
    // barrier; group B passes one extra barrier first and group A one extra at the end, which
    // puts B exactly one phase behind A for the whole kernel.
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    if (!isA) phase_barrier();
    long s = 0;            // global step
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_e = 0, sum_L = 0, sum_B1 = 0, sum_M = 0, sum_B2 = 0,
                       n_units = 0;
    (void)st_a; (void)st_b; (void)st_c; (void)st_d; (void)st_e; (void)sum_L; (void)sum_B1; (void)sum_M; (void)sum_B2;
    (void)n_units;
    unsigned long long sum_L2 = 0, sum_W = 0, sum_B3 = 0, sum_M2 = 0, sum_B4 = 0;
    (void)sum_L2; (void)sum_W; (void)sum_B3; (void)sum_M2; (void)sum_B4;
    for (long item = 0; item < n_items; ++item) {
        const int slot = (int)(item & 1);
        const int chunk = (int)(item % CPK);
        const long tile = wk + (item / CPK) * G;
        // the item after the last re-fetches the last item's patch into the idle slot: every
        // step then issues the same number of loads, which the counted waits rely on
        const char* nsrc = item_src(item + 1 < n_items ? item + 1 : item);
        const int nedges = FUSE1 ? item_edges(item + 1 < n_items ? item + 1 : item) : 0;
        auto step = [&](auto tapc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value;
            FR_STAMP(st_a);
            phase_L(tapc, std::integral_constant<int, 0>{}, chunk, s, slot, nsrc, nedges);
            if (NU == 2) {
                FR_STAMP(st_b);
                phase_barrier();
                FR_STAMP(st_c);
                phase_M(tapc, std::integral_constant<int, 0>{}, chunk, tile, s, slot);
                FR_STAMP(st_d);
                phase_barrier();
                FR_STAMP(st_e);
#if FR_DBG & 8
                sum_L += st_b - st_a; sum_B1 += st_c - st_b; sum_M += st_d - st_c; sum_B2 += st_e - st_d; ++n_units;
#endif
                phase_L(tapc, std::integral_constant<int, NU - 1>{}, chunk, s, slot, nsrc, nedges);
                FR_STAMP(st_a);
#if FR_DBG & 8
                sum_L2 += st_a - st_e;
#endif
            }
            // End of the step's last L phase.  Before the barrier that follows, this wave's pieces
            // of weight stage s+1 (issued in L(s-2)) and everything older (at tap 8: the next
            // patch) must have landed: group A reads them two phases from here, group B's partner
            // one.  What was issued after L(s-2)'s loads stays in flight: e(s-2), l(s-1), e(s-1),
            // l(s), with l(t) = NPY weight pieces + one patch piece in taps 0..5 and e(t) = the
            // epilogue stores of M(t).
            if (RES) {
                if (tap == 8 && !FUSE1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next patch has landed
            } else {
                constexpr int l1 = NPY + ((tap + 8) % 9 < 6 ? 1 : 0), l0 = NPY + (tap < 6 ? 1 : 0);
                if (s + 4 >= S) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (DEFER) {
                    constexpr int e2 = tap < 2 ? 0 : (2 * (tap - 2) < NCH) + (2 * (tap - 2) + 1 < NCH);
                    constexpr int e1 = tap < 1 ? 0 : (2 * (tap - 1) < NCH) + (2 * (tap - 1) + 1 < NCH);
                    if (chunk == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(l1 + l0 + e2 + e1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(l1 + l0) : "memory");
                } else {
                    // stores: all NCH in M(tap 8) of a tile's last item
                    if (tap <= 1 && chunk == 0 && item > 0)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(l1 + l0 + NCH) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(l1 + l0) : "memory");
                }
            }
            FR_STAMP(st_b);
            phase_barrier();
            FR_STAMP(st_c);
            phase_M(tapc, std::integral_constant<int, NU - 1>{}, chunk, tile, s, slot);
            FR_STAMP(st_d);
            phase_barrier();
            FR_STAMP(st_e);
#if FR_DBG & 8
            if (NU == 2) { sum_W += st_b - st_a; sum_B3 += st_c - st_b; sum_M2 += st_d - st_c; sum_B4 += st_e - st_d; }
#endif
            ++s;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
    }
    if (isA) phase_barrier();
#if FR_DBG & 8
    if (blockIdx.x == 0 && lane == 0) {
        g_stamp[wave * 5 + 0] = sum_L; g_stamp[wave * 5 + 1] = sum_B1; g_stamp[wave * 5 + 2] = sum_M;
        g_stamp[wave * 5 + 3] = sum_B2; g_stamp[wave * 5 + 4] = n_units;
    }
    if (blockIdx.x == 0 && lane == 1) {
        g_stamp[40 + wave * 3] = sum_L2 + (sum_W << 32); g_stamp[40 + wave * 3 + 1] = sum_B3 + (sum_M2 << 32);
        g_stamp[40 + wave * 3 + 2] = sum_B4;
    }
#endif
    static_assert(NCH <= 8, "chunk lists");
    if (DEFER) {    // epilogue of the last tile
        epi_chunk(std::integral_constant<int, 0>{}); epi_chunk(std::integral_constant<int, 1>{});
        epi_chunk(std::integral_constant<int, 2>{}); epi_chunk(std::integral_constant<int, 3>{});
    }
}

// Gallery scan.  Tile = 256 gallery rows (MFMA A operand, streamed from HBM through an LDS-DMA
// ring) by RY in {64,128,256} queries (B operand, one query per lane & 15; its 64-deep slices
// come from L2 through a second ring).  The 1-D grid holds nb gallery walkers per query tile;
// walker gb visits gallery tiles gb, gb+nb, ...  Blocks that differ only in their query tile are
// adjacent on one XCD so that the gallery stream of one is an L2 hit for the others.  The
// gallery allocation is padded to whole tiles, so loads never leave it; rows >= nrows are
// masked in the last tile.  Ring depths are chosen per query tile so that the LDS is used up:
// the gallery stream runs XS-1 steps ahead (HBM latency), the query slices YS-1 steps (L2).
// A ping-pong schedule (waves 0..3 and 4..7 one barrier-delimited phase apart) and a software-
// pipelined K-step were both measured slower than this lock-step one for this kernel.
// Top-k: every lane keeps a sorted list per query column; before a tile's 4*NX new scores are
// examined, the four lanes that share a query exchange their k-th best: a score below the best
// of those can never reach the final list.
template <int K, int RY, int WX, int WY, bool USE_KEYS, bool WORK = false>
__global__ __launch_bounds__(512) void scan_topk_kernel(const half_t* __restrict__ gal,
                                                           const int32_t* __restrict__ gkeys, long nrows, int D,
                                                           const half_t* __restrict__ q,
                                                           const int32_t* __restrict__ qkeys, int Bq, int nb,
                                                           float* __restrict__ part_s, long long* __restrict__ part_r,
                                                           unsigned* __restrict__ qbound,
                                                           const int32_t* __restrict__ tile_list, long ntl,
                                                           const ScanWork* __restrict__ work) {
    static_assert(WX * WY == 8 && RY % 64 == 0, "shape");
    constexpr int TX = 256 / WX, TY = RY / WY, NX = TX / 16, NY = TY / 16;
    constexpr int XB = 256 * 128, YB = RY * 128;
    constexpr int XS = RY <= 64 ? 4 : 3, YS = RY <= 128 ? 3 : 2;
    constexpr int DX = XS - 1, DY = YS - 1;
    constexpr int NPX = 4, NPY = RY / 64;
    constexpr int XSPLIT = 1, NXH = NX / XSPLIT, NUN = 2 * XSPLIT;     // 2 halves the fragment registers, ~4 % slower
    // loads that may stay in flight at the end of a step (see the wait below)
    constexpr int INFLIGHT = (DX == 3 && DY == 2) ? 2 * NPX + NPY : (DX == 2 && DY == 2) ? NPX + NPY : NPX;
    static_assert((DX == 3 && DY == 2) || (DX == 2 && DY == 2) || (DX == 2 && DY == 1), "ring depths");
    // Block-level score bounds in LDS for the HBM-bound 64-query tile (see the tile end below):
    // KP words per query, of which K are slots.
    constexpr bool LB = RY <= 64;
    constexpr int KP = (K + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const ys = smem + XS * XB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave / WY, wy = wave % WY;
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
#if FR_DBG & 64
    unsigned long long sc_t0 = 0, sc_t1 = 0, sc_t2 = 0, sc_t3 = 0, sc_sel = 0, sc_ta = 0, sc_tb = 0, sc_n = 0, sc_it = 0;
    unsigned long long sc_tc = 0, sc_td = 0, sc_pre = 0, sc_prel = 0, sc_loop = 0;
    FR_STAMP(sc_t0);
#endif
    // Either a regular grid (nb walkers for each of gridDim.x / nb query tiles, all over the same
    // tiles), or one ScanWork item per block: the query tile, the tile list and the walker number
    // of a search that looks up every key's queries in that key's tiles only.
    // (WORK is a template parameter so that the regular launches carry none of this.)
    int qtiles, q0, gb, bql = Bq;       // bql: queries at and past this one are not this block's
    if (WORK) {
        const ScanWork w = work[blockIdx.x];
        q0 = w.q0; bql = w.qend; gb = w.gb; nb = w.nb; ntl = w.ntl;
        tile_list = w.tile_off >= 0 ? tile_list + w.tile_off : nullptr;
        qtiles = 2;                     // other blocks may read the same tiles: keep them in L2
    } else {
        qtiles = gridDim.x / nb;
        const int L = blockIdx.x, xcd = L & 7, seq = L >> 3;
        gb = (seq / qtiles) * 8 + xcd;
        q0 = (seq % qtiles) * RY;
    }
    const int nk = D / 64;
    // With a tile list the launch scans only the listed 256-row tiles (the tiles that may hold rows
    // of the queries' key), in list order; row numbers stay those of the whole gallery.
    const long ntiles = (WORK && tile_list) ? ntl : (nrows + 255) / 256;
    const long my_tiles = gb < ntiles ? (ntiles - gb + nb - 1) / nb : 0;
    const long S = my_tiles * nk;
    const int cb = ((l7 ^ (lane >> 3)) << 4);

    // Per-lane byte offsets of this lane's 16 B inside the query block (row-major; the host checks
    // that it stays below 4 GiB).  Piece i of a stage is 64 rows further on, a uniform distance that
    // goes into the scalar part of the address, so one register serves all pieces; query rows past
    // Bq are clamped to the last row (their columns are never written out).
    const unsigned xo0 = (unsigned)((8 * wave + (lane >> 3)) * D * 2 + cb);
    // inside a gallery K-step block (256 rows x 128 B, contiguous: see gal_index) the same lane's 16 B
    const unsigned xg0 = (unsigned)((8 * wave + (lane >> 3)) * 128 + cb);
    // the query block's offsets are the gallery tile's plus the query tile's first row (a uniform
    // term): they are rebuilt from xo0 at every use instead of occupying a register of their own
    const unsigned yq0 = (unsigned)q0 * (unsigned)D * 2u;
    const unsigned yomax = (unsigned)(bql - 1) * (unsigned)D * 2u + (unsigned)cb;
    const unsigned pstride = 64u * (unsigned)D * 2u;
    TopK<K> top[NY];
#pragma unroll
    for (int y = 0; y < NY; ++y) top[y].init();

    float4v acc[NY][NX];
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[y][x] = float4v{0.f, 0.f, 0.f, 0.f};

    if (S > 0) {
        long itile = gb; int iks = 0;   // next gallery step to issue
        int yks = 0;                    // next query slice to issue
        auto issue_x = [&](int slot) __attribute__((always_inline)) {
            // (uniform index, scalar load: a vector load here would be waited for with everything in flight)
            const long ptile = (WORK && tile_list) ? (long)tile_list[__builtin_amdgcn_readfirstlane((int)itile)] : itile;
            const char* const base = (const char*)gal + ((size_t)ptile * nk + (size_t)iks) * 32768;
#pragma unroll
            // with one query tile every gallery byte is read by one CU, once: stream it non-temporally
            // (+3..6 % on the HBM-bound shapes); with several, the other tiles' blocks want it in L2
            for (int i = 0; i < NPX; ++i) {
                unsigned o = xg0;
                asm volatile("" : "+v"(o));     // keeps the 64-bit sum out of loop-invariant registers
                if (qtiles == 1) glds16_nt(base + (size_t)i * 8192 + o, xs + slot * XB + (wave + 8 * i) * 1024);
                else glds16(base + (size_t)i * 8192 + o, xs + slot * XB + (wave + 8 * i) * 1024);
            }
            if (++iks == nk) { iks = 0; itile += nb; }
        };
        auto issue_y = [&](int slot) __attribute__((always_inline)) {
            unsigned yb = xo0;
            asm volatile("" : "+v"(yb));        // as above: the clamped offsets are recomputed, not kept
            yb += yq0;
#pragma unroll
            for (int i = 0; i < NPY; ++i) {
                unsigned o = yb + (unsigned)i * pstride;
                o = o < yomax ? o : yomax;
                glds16((const char*)q + (size_t)yks * 128 + o, ys + slot * YB + (wave + 8 * i) * 1024);
            }
            if (++yks == nk) yks = 0;
        };
        auto phase_barrier = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (LB) {
            unsigned* const lb = (unsigned*)(smem + XS * XB + YS * YB);      // [RY][KP]
            for (int i = tid; i < RY * KP; i += 512) lb[i] = 0u;
        }
        // prologue: the first DY query slices and DX gallery steps; everything lands before phase 0
#pragma unroll
        for (int i = 0; i < DY; ++i)
            if (i < S) issue_y(i);
#pragma unroll
        for (int i = 0; i < DX; ++i)
            if (i < S) issue_x(i);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if FR_DBG & 64
        FR_STAMP(sc_t1);
#endif
        // block-level bound per query column group (-inf: none yet); byte offset of the lane's
        // slot row in LDS
        float lbval[LB ? NY : 1];
#pragma unroll
        for (int y = 0; y < (LB ? NY : 1); ++y) lbval[y] = -INFINITY;
        const unsigned lbrow = lds_offset(smem + XS * XB + YS * YB) + (unsigned)((wy * TY + l15) * KP * 4);

        // A step runs as NUN units u = (kk, xh): kk = 32-deep half of the step, xh = 1/XSPLIT of the
        // gallery rows.  The query fragments of a kk are read with its first unit and kept; the large
        // tile splits the rows so that the fragments of a unit fit beside 128 accumulators and the
        // top-k lists.
        half8 fx[NXH], fy[NY];
        int xslot = 0, yslot = 0, ks = 0;
        long tile = gb;
        unsigned tdone = 0;             // tiles this block has finished
        for (long s = 0; s < S; ++s) {
            const char* const xbuf = xs + xslot * XB;
            const char* const ybuf = ys + yslot * YB;
            auto unit = [&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value, kk = u / XSPLIT, xh = u % XSPLIT;
                if (u == 0) {
                    if (s + DY < S) issue_y(yslot == 0 ? YS - 1 : yslot - 1);
                }
                if (u == NUN - 1) {
                    if (s + DX < S) issue_x(xslot == 0 ? XS - 1 : xslot - 1);
                }
                const int coff = (((kk * 4 + g0) ^ l7) << 4);
                if (xh == 0) {
#pragma unroll
                    for (int y = 0; y < NY; ++y) fy[y] = *(const half8*)(ybuf + (wy * TY + y * 16 + l15) * 128 + coff);
                }
#pragma unroll
                for (int x = 0; x < NXH; ++x)
                    fx[x] = *(const half8*)(xbuf + (wx * TX + (xh * NXH + x) * 16 + l15) * 128 + coff);
                // a tile's first MFMAs start from zero instead of from cleared accumulators
                if (kk == 0 && ks == 0) {
                    const float4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int y = 0; y < NY; ++y)
#pragma unroll
                        for (int x = 0; x < NXH; ++x)
                            acc[y][xh * NXH + x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], zero, 0, 0, 0);
                } else {
#pragma unroll
                    for (int y = 0; y < NY; ++y)
#pragma unroll
                        for (int x = 0; x < NXH; ++x)
                            acc[y][xh * NXH + x] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], acc[y][xh * NXH + x], 0, 0, 0);
                }
            };
            unit(std::integral_constant<int, 0>{});
            unit(std::integral_constant<int, 1>{});
            if constexpr (NUN == 4) {
                unit(std::integral_constant<int, 2>{});
                unit(std::integral_constant<int, 3>{});
            }
            if (ks == nk - 1) {
#if FR_DBG & 64
                FR_STAMP(sc_ta);
#endif
                const long ptile = (WORK && tile_list) ? (long)tile_list[__builtin_amdgcn_readfirstlane((int)tile)] : tile;
                const int rbase = (int)(ptile * 256) + wx * TX + g0 * 4;    // rows fit 31 bits (checked at launch)
                // Rows whose key does not match the query's, and rows past the end of the gallery (last
                // tile only), are struck out of the accumulators in place (the next tile starts
                // from zero anyway): the selection then reads the accumulators themselves and needs no copy.
                // Key test: (key ^ kx) <= km as unsigned, with (kx, km) = (query key, 0) for an exact
                // match and (0, 0x7fffffff) for "any live row" (tombstones are negative).
                // Nothing below can change a list unless some lane's best raw score beats its own tail
                // (striking rows out only lowers scores): the usual tile, once the tails carry a good
                // bound, ends here without fetching a key.
                bool some = !USE_KEYS;       // without keys the per-query test below is this same test
#pragma unroll
                for (int y = 0; USE_KEYS && y < NY; ++y) {
                    const float4v (&a)[NX] = acc[y];
                    float mq = max3_raw(a[0][0], a[0][1], a[0][2]);
                    mq = max3_raw(mq, a[0][3], a[0][3]);
#pragma unroll
                    for (int x = 1; x < NX; ++x) {
                        mq = max3_raw(mq, a[x][0], a[x][1]);
                        mq = max3_raw(mq, a[x][2], a[x][3]);
                    }
                    some = some || (LB ? __any(mq > top[y].s[K - 1] && mq >= lbval[LB ? y : 0]) : __any(mq > top[y].s[K - 1]));
                }
                if (some) {
                if (USE_KEYS) {
                    int4 kv[NX];
#pragma unroll
                    for (int x = 0; x < NX; ++x) kv[x] = *(const int4*)(gkeys + rbase + x * 16);   // padded allocation: in bounds
                    // the queries' keys are fetched here, once per tile, rather than kept in registers
                    int mykey[NY];
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        const int qq = q0 + wy * TY + y * 16 + l15;
                        mykey[y] = (qkeys && qq < bql) ? qkeys[qq] : -2;
                    }
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        __builtin_amdgcn_sched_barrier(0);
                        // any other negative query key (the tombstone value included) is no row key:
                        // it is compared with a value that no stored key takes, so it matches nothing
                        const unsigned kx = mykey[y] == -2 ? 0u : (mykey[y] < 0 ? 0x80000000u : (unsigned)mykey[y]);
                        const unsigned km = mykey[y] == -2 ? 0x7fffffffu : 0u;
#pragma unroll
                        for (int x = 0; x < NX; ++x) {
                            const unsigned k4[4] = {(unsigned)kv[x].x, (unsigned)kv[x].y, (unsigned)kv[x].z, (unsigned)kv[x].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[y][x][j] = ((k4[j] ^ kx) <= km) ? acc[y][x][j] : -INFINITY;
                        }
                    }
                }
                if ((ptile + 1) * 256 > nrows) {
                    const int nr = (int)nrows;
#pragma unroll
                    for (int y = 0; y < NY; ++y)
#pragma unroll
                        for (int x = 0; x < NX; ++x)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[y][x][j] = (rbase + x * 16 + j < nr) ? acc[y][x][j] : -INFINITY;
                }
                }   // some
                // Block-level bound (64-query tile).  A lane's list sees 1/(4*WX) of the block's rows,
                // so while the lists fill nearly every score beats its lane's tail and is inserted:
                // on a 1 M-row gallery a quarter of the launch went into those insertions.  The block
                // keeps K slots per query in LDS; each of the 4*WX lists of a query belongs to one slot
                // (list number mod K) and raises it (LDS atomic max) to the best score its lane has
                // seen, this tile included.  The lists cover disjoint rows, so the slots' minimum is a
                // lower bound on the query's K-th best score over everything the BLOCK has scanned:
                // scores below it are not inserted (scores equal to it are: ties are ordered by row).
                // With the barrier between raising and reading, the first tile is already filtered by
                // the K-th best of its own 4*WX lane maxima.  Stale slots only weaken the bound.
                if constexpr (LB) {
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        const float4v (&a)[NX] = acc[y];
                        float mq = max3_raw(a[0][0], a[0][1], a[0][2]);
                        mq = max3_raw(mq, a[0][3], a[0][3]);
#pragma unroll
                        for (int x = 1; x < NX; ++x) {
                            mq = max3_raw(mq, a[x][0], a[x][1]);
                            mq = max3_raw(mq, a[x][2], a[x][3]);
                        }
                        mq = some ? fmaxf(mq, top[y].s[0]) : top[y].s[0];     // keyed scan without candidates: raw scores do not count
                        const unsigned hk = ordered_key(mq);
                        if (hk > ordered_key(-INFINITY))
                            lds_max_u32(lbrow + (unsigned)((y * 16 * KP + (wx * 4 + g0) % K) * 4), hk);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        const unsigned bm = lds_min_words<KP / 4, K>(lbrow + (unsigned)(y * 16 * KP * 4));
                        lbval[y] = bm > 1u ? ordered_value(bm) : -INFINITY;     // 0: some slot is still empty
                    }
                }
#if FR_DBG & 64
                FR_STAMP(sc_tc);
                sc_pre += sc_tc - sc_ta;
#endif
                if (some) {
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    // one query column group at a time: interleaving the groups would need their
                    // temporaries live together, beside 128 accumulators and the lists
                    __builtin_amdgcn_sched_barrier(0);
#if FR_DBG & 64
                    FR_STAMP(sc_tc);
#endif
                    // The common case once the lists' tails carry a good bound: no lane's best of its
                    // 4*NX scores beats its own tail.  One three-input max per two scores.
                    {
                        const float4v (&a)[NX] = acc[y];
                        float mq = max3_raw(a[0][0], a[0][1], a[0][2]);
                        mq = max3_raw(mq, a[0][3], a[0][3]);
#pragma unroll
                        for (int x = 1; x < NX; ++x) {
                            mq = max3_raw(mq, a[x][0], a[x][1]);
                            mq = max3_raw(mq, a[x][2], a[x][3]);
                        }
                        if (LB ? !__any(mq > top[y].s[K - 1] && mq >= lbval[LB ? y : 0]) : !__any(mq > top[y].s[K - 1])) continue;
                    }
                    // Lower bounds on the K-th best score of the query over the four lanes' lists:
                    // (a) any lane's own K-th best; (b) min(best (KA+1)-th of any lane, worst
                    // (KB+1)-th of all lanes): the lane that supplies the first term holds KA+1
                    // scores at or above it and each other lane KB+1, K or more in all.
                    constexpr int KB = (K - 1) / 5, KA = K - 3 * (KB + 1) - 1;
                    static_assert(KA >= KB && KA < K && (KA + 1) + 3 * (KB + 1) >= K, "threshold bound");
                    float thr = lanes4_max(top[y].s[K - 1]);
                    thr = fmaxf(thr, fminf(lanes4_max(top[y].s[KA]), lanes4_min(top[y].s[KB])));
                    if constexpr (LB) thr = fmaxf(thr, lbval[LB ? y : 0]);
                    float4v (&v)[NX] = acc[y];        // struck-out entries may be overwritten: the next tile starts from zero
                    // best and second best of the lane's 4*NX scores, branch-free
                    float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
                    for (int x = 0; x < NX; ++x)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            m2 = fmaxf(m2, fminf(m1, v[x][j]));
                            m1 = fmaxf(m1, v[x][j]);
                        }
                    const bool q1 = m1 >= thr && m1 > top[y].s[K - 1];
                    if (!__any(q1)) continue;                  // the common case late in the scan
                    if (!__any(m2 >= thr && m2 > top[y].s[K - 1])) {
                        // no lane holds two qualifying scores (the usual case otherwise): one
                        // insertion per lane, of its best; the lowest row among equal scores
                        int hit = 0;
#pragma unroll
                        for (int x = NX - 1; x >= 0; --x)
#pragma unroll
                            for (int j = 3; j >= 0; --j)
                                if (v[x][j] == m1) hit = x * 16 + j;
                        if (q1) top[y].template push<false>(m1, (int)rbase + hit);
                        continue;
                    }
                    if constexpr (K < FR_LOOP_MIN_K) {
                        // short lists: one branch per 16-row block, taken only when some lane of the
                        // wave holds a qualifying score among the block's four; inside, the lane's
                        // best of the four is inserted and struck out until none qualifies
#pragma unroll
                        for (int x = 0; x < NX; ++x) {
                            const float mx = fmaxf(fmaxf(v[x][0], v[x][1]), fmaxf(v[x][2], v[x][3]));
                            if (mx >= thr && mx > top[y].s[K - 1]) {
                                // Nearly always one of the lane's four scores qualifies: insert the
                                // best (lowest row first among equal scores, the order the list
                                // expects), strike it out and look again.
                                float c0 = v[x][0], c1 = v[x][1], c2 = v[x][2], c3 = v[x][3], cm = mx;
                                do {
                                    const int jb = c0 == cm ? 0 : c1 == cm ? 1 : c2 == cm ? 2 : 3;
                                    top[y].template push<false>(cm, (int)(rbase + x * 16) + jb);
                                    c0 = jb == 0 ? -INFINITY : c0; c1 = jb == 1 ? -INFINITY : c1;
                                    c2 = jb == 2 ? -INFINITY : c2; c3 = jb == 3 ? -INFINITY : c3;
                                    cm = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3));
                                } while (cm >= thr && cm > top[y].s[K - 1]);
                            }
                        }
                        continue;
                    }
                    float m = m1;
#if FR_DBG & 64
                    FR_STAMP(sc_td);
                    sc_prel += sc_td - sc_tc;
#endif
                    // Long lists, taken while any lane still holds a score that belongs in its list:
                    // each lane inserts its best remaining score (equal scores: lowest row first, the
                    // order the list expects), strikes it out and looks for the next.  The cost follows
                    // the number of real candidates per lane, not the tile's 4*NX scores.
                    while (__any(m >= thr && m > top[y].s[K - 1])) {
#if FR_DBG & 64
                        ++sc_it;
#endif
                        const bool want = m >= thr && m > top[y].s[K - 1];
                        int hit = -1;
#pragma unroll
                        for (int x = NX - 1; x >= 0; --x)
#pragma unroll
                            for (int j = 3; j >= 0; --j)
                                if (v[x][j] == m) hit = x * 16 + j;      // lowest matching (x, j) wins
                        if (want) top[y].template push<false>(m, (int)rbase + hit);
                        float m2 = -INFINITY;
#pragma unroll
                        for (int x = 0; x < NX; ++x)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (x * 16 + j == hit) v[x][j] = -INFINITY;
                                m2 = fmaxf(m2, v[x][j]);
                            }
                        m = want ? m2 : -INFINITY;     // a lane with nothing to insert is done with this tile
                    }
#if FR_DBG & 64
                    FR_STAMP(sc_tc);
                    sc_loop += sc_tc - sc_td;
#endif
                }
                }   // some
                // Cross-block bound.  A lane's list sees 1/(4*WX*nb) of the rows, so its own K-th best
                // is a weak filter and, without more, nearly every tile finds some lane with a score
                // to insert.  qbound holds K slots per query; every wave of every block belongs to
                // one slot (wave number mod K) and raises it to the best score its lanes hold for the query.
                // The slots' minimum t is then a lower bound on the query's K-th best score over the
                // WHOLE gallery seen so far: K different lists each hold a row scoring >= t.  Scores
                // below t can never reach the answer (scores equal to t can: the order among equal
                // scores is by row), so list entries below t become placeholders (row -1) at
                // the largest value below t, which makes the ordinary "beats the tail" test apply the
                // bound.  Stale or lost slot updates only weaken the bound.  Done after tiles 1, 2,
                // 4, 8, 16, ...: the wait for the reads costs about one memory round trip, and the
                // bound moves little once a block has seen a few thousand rows.
                ++tdone;
                if (qbound != nullptr && (tdone & (tdone - 1)) == 0) {
                    unsigned bk[NY], mine[NY];
                    // a "list" here is the union of the four lanes' lists of one query in one wave
                    const unsigned list = (unsigned)gb * WX + wx;
                    const int myslot = (int)(list % K);
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        const int qq = q0 + wy * TY + y * 16 + l15;          // qbound has a row for every query of the tile
                        const unsigned* const sl = qbound + (size_t)qq * K;
                        unsigned m = 0xffffffffu, cur = 0u;
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const unsigned v = __hip_atomic_load(sl + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            m = v < m ? v : m;
                            cur = j == myslot ? v : cur;
                        }
                        bk[y] = m; mine[y] = cur;
                    }
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        // raise the list's slot only when that changes it: all lists of a slot address
                        // one word, and atomics on one word are served one after the other
                        const int qq = q0 + wy * TY + y * 16 + l15;
                        // (a lane without a row yet holds -inf: below every real score's key)
                        const unsigned hk = ordered_key(lanes4_max(top[y].s[0]));
                        // (work-list form: the tile's columns past the key's last query belong to other
                        // keys' queries, whose bounds this block, scoring them without their key, must not touch)
                        if (g0 == 0 && (!WORK || qq < bql) && hk > ordered_key(-INFINITY) && hk > mine[y])
                            (void)__hip_atomic_fetch_max(qbound + (size_t)qq * K + myslot, hk, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                        if (bk[y] > 1u) {                       // 0: some slot is still empty
                            const float below = ordered_value(bk[y] - 1u);
#pragma unroll
                            for (int i = 0; i < K; ++i) {
                                const bool drop = !(top[y].s[i] > below);
                                top[y].r[i] = drop ? -1 : top[y].r[i];
                                top[y].s[i] = drop ? below : top[y].s[i];
                            }
                        }
                    }
                }
                ks = 0; tile += nb;
#if FR_DBG & 64
                FR_STAMP(sc_tb);
                sc_sel += sc_tb - sc_ta; ++sc_n;
#endif
            } else {
                ++ks;
            }
            // Gallery step s+1 and query slice s+1 must have landed before the next step reads them.
            // In issue order the later of the two is followed only by the loads counted in INFLIGHT,
            // which may stay outstanding; near the end of the block's work fewer loads are issued,
            // so everything is waited for.
            if (s + DX < S && s + DY < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            phase_barrier();
            xslot = xslot == XS - 1 ? 0 : xslot + 1;
            yslot = yslot == YS - 1 ? 0 : yslot + 1;
        }
    }

#if FR_DBG & 64
    FR_STAMP(sc_t2);
#endif
    // merge the WX*4 lists of each query through LDS
    constexpr int NL = WX * 4;
    __syncthreads();
    float* ls = (float*)smem;                       // [RY][NL][K]
    int* lr = (int*)(smem + RY * NL * K * 4);
#pragma unroll
    for (int y = 0; y < NY; ++y) {
        const int ql = wy * TY + y * 16 + l15, slot = wx * 4 + g0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ls[(ql * NL + slot) * K + i] = top[y].s[i];
            lr[(ql * NL + slot) * K + i] = top[y].r[i];
        }
    }
    __syncthreads();
    // Each query's NL lists are sorted (score descending, row ascending inside equal scores), so
    // their best K come out of a K-round merge of the list heads: NL adjacent lanes per query,
    // one per list; a round finds the group's best head (max score, then min row) by DPP
    // exchanges inside the group and the lane that holds it moves on to its list's next entry.
    constexpr int QPP = 512 / NL;                   // queries per pass
    static_assert(RY % QPP == 0 && (NL == 8 || NL == 16), "merge groups");
    const int sub = tid % NL, qslot = tid / NL;
#pragma unroll 1
    for (int qb = 0; qb < RY; qb += QPP) {
        const int ql = qb + qslot;
        const float* const hs = ls + (ql * NL + sub) * K;
        const int* const hrw = lr + (ql * NL + sub) * K;
        int pos = 0;
        int r = hrw[0];
        unsigned hk = r >= 0 ? ordered_key(hs[0]) : 0u;
        unsigned hr = r >= 0 ? (unsigned)r : 0xffffffffu;
        const size_t o = ((size_t)gb * Bq + q0 + ql) * K;
        const bool writer = sub == 0 && q0 + ql < bql;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const unsigned kmax = group_umax<NL>(hk);
            const bool c1 = hk == kmax;
            const unsigned rmin = ~group_umax<NL>(~(c1 ? hr : 0xffffffffu));
            if (writer) {
                part_s[o + i] = kmax ? ordered_value(kmax) : -INFINITY;
                part_r[o + i] = kmax ? (long long)rmin : -1;
            }
            if (c1 && hr == rmin && kmax != 0u) {
                ++pos;
                r = pos < K ? hrw[pos < K ? pos : 0] : -1;
                const float v = hs[pos < K ? pos : 0];
                hk = r >= 0 ? ordered_key(v) : 0u;
                hr = r >= 0 ? (unsigned)r : 0xffffffffu;
            }
        }
    }
#if FR_DBG & 64
    FR_STAMP(sc_t3);
    if (tid == 0 && blockIdx.x < 256) {
        unsigned long long* const o = g_scan_stamp + blockIdx.x * 16;
        o[0] = sc_t0; o[1] = sc_t1; o[2] = sc_t2; o[3] = sc_t3; o[4] = sc_sel; o[5] = sc_n; o[6] = sc_it;
        o[7] = sc_pre; o[8] = sc_prel; o[9] = sc_loop;
    }
#endif
}

// ------------------------------------- 64-query scan with a selector wave ---
// LDS accesses of the selection queues, written as the instructions themselves (see lds_offset).
FR_DEV unsigned lds_read_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
FR_DEV unsigned long long lds_read_u64(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
FR_DEV void lds_write_u32(unsigned addr, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
FR_DEV void lds_write_u64(unsigned addr, unsigned long long v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// 16 consecutive words at addr (64-byte aligned)
FR_DEV void lds_read_16u32(unsigned addr, unsigned (&o)[16]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = v0[j]; o[4 + j] = v1[j]; o[8 + j] = v2[j]; o[12 + j] = v3[j]; }
}
// four 8-byte words at addr, addr + stride, ... (stride in bytes, a compile-time constant below 16 KiB)
FR_DEV void lds_read_4u64(unsigned addr, int stride, unsigned long long (&o)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:%5\n\tds_read_b64 %2, %4 offset:%6\n\t"
                 "ds_read_b64 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                 : "v"(addr), "n"(512), "n"(1024), "n"(1536)
                 : "memory");
    (void)stride;
}
// two words, 64 bytes apart (the thresholds of a lane's two query columns), one wait
FR_DEV void lds_read_2u32_64(unsigned addr, unsigned& a, unsigned& b) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}
FR_DEV unsigned lds_add_rtn_u32(unsigned addr, unsigned v) {
    unsigned r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}

// The 64-query tile of the gallery scan (HBM-bound: 64 flop per gallery byte) with the top-k
// selection taken off the streaming waves.  576 threads: waves 0..7 stream the gallery exactly
// as scan_topk_kernel<K, 64, 4, 2> does (256-row tiles through an LDS-DMA ring, MFMA against
// the 64 queries), wave 8 is the SELECTOR: lane l keeps THE sorted top-K list of query l of
// the tile for everything this block scans, in registers.
//   * thr[64] in LDS is a lower bound on each query's K-th best score (the selector's list
//     tail once the list is full, raised by the cross-block bound when there is one), as an
//     ordered key, only ever raised (LDS atomic max).
//   * At a tile end the streaming waves only FILTER: one three-input max per two scores and a
//     ballot; a lane with scores >= thr reserves as many consecutive entries of its query's
//     queue in LDS as it has survivors (one LDS atomic add) and writes them, 8 bytes each:
//     {ordered key of the score, ~row}.  Two queues, by tile parity.
//   * The selector drains the queue of tile t into its lists while the streaming waves run the
//     steps of tile t+1 (a few entries per step, between the same barriers; insertion without
//     a branch), publishes the new thr, and, when the launch has cross-block bounds, exchanges
//     them through device memory: its loads and stores wait in nobody's memory queue but its own.
//   * A block's first tile has no threshold yet, and every block of the launch is at its first
//     tile at the same time, so nothing there may be serial.  The 16 lists of a query (4 lanes
//     x 4 waves) put their best score into a table; the largest of them that at least K reach
//     is a first threshold (K rows score at least that); the survivors are queued as usual and
//     then RANKED by the streaming waves, eight lanes per query (rank = number of the query's
//     entries with a larger key), which leaves every query's start list in order in LDS for
//     the selector to pick up.
//   * A full queue (more than CAP survivors of one query in one tile: ties, adversarial
//     orders) is exact too: lanes that found no room raise a flag, every wave sees it after the
//     barrier, the selector drains at once, and the lanes try again with the new thresholds.
// Every wave executes the same sequence of barriers: the counts (tile ends, the first tile's
// extra ones, two per overflow round) derive from values all nine waves read from LDS
// after a barrier and before anyone writes them again.
// The block's lists come out sorted (score descending, row ascending among equal scores)
// straight from the selector's registers: there is no merge at the end.
// Measured on 1 M x 512 x 64 queries, top-10 (one MI355X, same box, queries and merge as the
// benchmark runs them): 0.205 -> 0.173 ms (0.74 of 8 TB/s); 1 M x 4096 x 64: 1.185 -> 1.165 ms.
template <int K, bool USE_KEYS, bool WORK, int XS, int YS, int CAP>
__global__ __launch_bounds__(576) void scan_sel_kernel(const half_t* __restrict__ gal,
                                                          const int32_t* __restrict__ gkeys, long nrows, int D,
                                                          const half_t* __restrict__ q,
                                                          const int32_t* __restrict__ qkeys, int Bq, int nb,
                                                          float* __restrict__ part_s, long long* __restrict__ part_r,
                                                          unsigned* __restrict__ qbound,
                                                          const int32_t* __restrict__ tile_list, long ntl,
                                                          const ScanWork* __restrict__ work, unsigned long long gen) {
    constexpr int RY = 64, WY = 2, TX = 64, TY = 32, NX = 4, NY = 2;
    constexpr int XB = 256 * 128, YB = RY * 128;
    constexpr int DX = XS - 1, DY = YS - 1, NPX = 4, NPY = 1;
    // loads that may stay in flight at the end of a step: those issued after the later of
    // gallery step s+1 and query slice s+1
    constexpr int IFX = (DX - 1) * (NPX + NPY), IFY = NPX + (DY - 1) * (NPX + NPY);
    constexpr int INFLIGHT = IFX < IFY ? IFX : IFY;
    static_assert(XS >= 3 && YS >= 2 && CAP >= K, "ring depths / queue length");
    constexpr int CTRL = XS * XB + YS * YB;         // thr[64] | cnt[2][64] | overflow tag
    constexpr int QUEUE = CTRL + 1024;              // [2][CAP][64] entries of 8 bytes
    constexpr unsigned LOWEST_KEY = 0x00800000u;    // ordered key of -FLT_MAX: below every score, above the struck-out -inf
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const ys = smem + XS * XB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = (wave & 7) / WY, wy = wave % WY;
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
    const unsigned a_thr = lds_offset(smem + CTRL), a_cnt = a_thr + 256, a_ovf = a_thr + 768;
    const unsigned a_queue = lds_offset(smem + QUEUE);
    // first tile only, in the second queue's memory (unused until the second tile's end):
    const unsigned a_boot = a_queue + CAP * 512;    // [64 queries][16 lists] best score of every list (ordered keys)
    const unsigned a_init = a_boot + 4096;          // [64 queries][K] start lists
    static_assert(4096 + 64 * K * 8 <= CAP * 512 && CAP % 8 == 0 && CAP <= 32, "first-tile tables");

    int qtiles, q0, gb, bql = Bq;       // bql: queries at and past this one are not this block's
    if (WORK) {
        const ScanWork w = work[blockIdx.x];
        q0 = w.q0; bql = w.qend; gb = w.gb; nb = w.nb; ntl = w.ntl;
        tile_list = w.tile_off >= 0 ? tile_list + w.tile_off : nullptr;
        qtiles = 2;                     // other blocks may read the same tiles: keep them in L2
    } else {
        qtiles = gridDim.x / nb;
        const int L = blockIdx.x, xcd = L & 7, seq = L >> 3;
        gb = (seq / qtiles) * 8 + xcd;
        q0 = (seq % qtiles) * RY;
    }
    const int nk = D / 64;
    const long ntiles = (WORK && tile_list) ? ntl : (nrows + 255) / 256;
    const long my_tiles = gb < ntiles ? (ntiles - gb + nb - 1) / nb : 0;
    const long S = my_tiles * nk;

    if (wave == 8) {
        // ------------------------------------------------------------ selector
        // The list is kept as K 64-bit keys, best first: {ordered key of the score, ~row}, so one
        // unsigned comparison orders by score and, among equal scores, by ascending row; 0 = empty.
        __builtin_amdgcn_s_setprio(3);              // short bursts of work that the streaming waves wait for
        unsigned long long key[K];
#pragma unroll
        for (int j = 0; j < K; ++j) key[j] = 0ull;
        const unsigned ql = (unsigned)lane;
#if FR_DBG & 64
        unsigned long long sl_a = 0, sl_b = 0, sl_drain = 0, sl_rounds = 0;
#endif
        if (S > 0) {
            lds_write_u32(a_thr + ql * 4, LOWEST_KEY);
            lds_write_u32(a_cnt + ql * 4, 0u);
            lds_write_u32(a_cnt + 256 + ql * 4, 0u);
            if (lane == 0) lds_write_u32(a_ovf, 0u);
#pragma unroll
            for (int j = 0; j < K; ++j) lds_write_u64(a_init + (ql * K + (unsigned)j) * 8, 0ull);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // the streaming waves' prologue barrier
            int pend = -1;                          // queue being drained beside the next tile's steps
            unsigned n = 0, i = 0;                  // this lane's entries in it; the round (same in every lane)
            unsigned chk = 0, tdone = 0;
            int ks = 0;
            unsigned bound = LOWEST_KEY;            // cross-block bound of this lane's query (ordered key)
            bool xpending = false;                  // the slots were asked for during the previous interval
            unsigned long long xv[K];
#pragma unroll
            for (int j = 0; j < K; ++j) xv[j] = 0ull;
            unsigned long long* const slots = (unsigned long long*)qbound + (size_t)(q0 + (int)ql) * K;
            const int qq = q0 + (int)ql;
            const unsigned myslot = (unsigned)gb % (unsigned)K;
            const unsigned myturn = (unsigned)gb / (unsigned)K, nclass = ((unsigned)nb + K - 1) / (unsigned)K;
            // the slots are asked for early in a tile and used at its last step but one (short tiles: at
            // the next tile's first step, before asking again)
            const int xask = nk >= 4 ? 1 : 0, xuse = nk >= 4 ? nk - 2 : 0;
#ifdef FR_SEL_R
            const int R = FR_SEL_R;
#else
            const int R = (CAP + nk - 1) / nk > 4 ? (CAP + nk - 1) / nk : 4;
#endif
            // One entry of every lane's queue: branch-free insertion.  b[j]: the entry beats list
            // place j (then it beats every later place too); place j takes the entry, its left
            // neighbour, or keeps its own.  Lanes without an entry insert key 0, which beats nothing.
            auto drain_round = [&](unsigned p) __attribute__((always_inline)) {
#if FR_DBG & 64
                ++sl_rounds;
#endif
                const unsigned long long e = lds_read_u64(a_queue + ((p * CAP + i) * 64 + ql) * 8);
                const unsigned long long nkey = i < n ? e : 0ull;
                bool b[K];
#pragma unroll
                for (int j = 0; j < K; ++j) b[j] = nkey > key[j];
#pragma unroll
                for (int j = K - 1; j > 0; --j) {
                    const unsigned long long left = b[j - 1] ? key[j - 1] : nkey;
                    key[j] = b[j] ? left : key[j];
                }
                key[0] = b[0] ? nkey : key[0];
                ++i;
            };
            auto publish = [&]() __attribute__((always_inline)) {
                const unsigned tail = (unsigned)(key[K - 1] >> 32);         // 0 while the list is not full
                lds_max_u32(a_thr + ql * 4, tail > bound ? tail : bound);
            };
            auto queued = [&](unsigned p) __attribute__((always_inline)) {
                const unsigned c = lds_read_u32(a_cnt + p * 256 + ql * 4);
                return c < (unsigned)CAP ? c : (unsigned)CAP;
            };
            for (long s = 0; s < S; ++s) {
                const bool tile_end = ks == nk - 1;
                if (xpending && (ks == xuse)) {
                    // Cross-block bound: K slots per query in device memory; slot j belongs to the blocks
                    // with (block mod K) = j, which scan rows no other class scans, and holds a score that
                    // some row of the class reached for the query: K different rows score at least the
                    // slots' minimum.  A slot is {launch number: 40 bits, leading 24 bits of the score's
                    // ordered key}: what earlier launches left behind reads as empty, so nothing clears the
                    // slots, and a key cut short is a lower score, which keeps the bound a bound.
                    // The blocks of a class take TURNS at writing their slot (one per tile: block c of the
                    // class after tile c, c + classes, ...), with a plain write-through store of the larger
                    // of what the slot held and the block's own best.  No atomics: 256 blocks raising the
                    // same few lines at once held the whole chip's stream up for tens of microseconds.  Two
                    // blocks out of step may write one slot at once and one value is lost: whatever is left
                    // is still a score a row of the class reached.
                    unsigned m = 0xffffffffu, mine = 0u;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const unsigned kj = (xv[j] >> 24) == gen ? (unsigned)(xv[j] & 0xffffffu) : 0u;
                        m = kj < m ? kj : m;
                        mine = (unsigned)j == myslot ? kj : mine;
                    }
                    const unsigned hk = (unsigned)(key[0] >> 40);       // 0: no row yet
                    // (work-list form: columns past the key's last query belong to other keys' queries)
                    if ((!WORK || qq < bql) && tdone % nclass == myturn && hk > mine)
                        __hip_atomic_store(slots + myslot, (gen << 24) | (unsigned long long)hk, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    m <<= 8;
                    bound = m > bound ? m : bound;
                    publish();
                    xpending = false;
                }
                if (pend >= 0) {
                    int rounds = tile_end ? CAP : R;        // a tile's last step completes the drain
#if FR_DBG & 64
                    FR_STAMP(sl_a);
#endif
                    while (rounds-- > 0 && __any(i < n)) drain_round((unsigned)pend);
#if FR_DBG & 64
                    FR_STAMP(sl_b);
                    sl_drain += sl_b - sl_a;
#endif
                    if (!__any(i < n)) {
                        lds_write_u32(a_cnt + (unsigned)pend * 256 + ql * 4, 0u);
                        publish();
                        pend = -1;
                    }
                }
                // Under a full-rate gallery stream a load takes microseconds to come back, and the wait
                // for it stands in front of a barrier every streaming wave waits at: hence the distance
                // between asking and using.
                if (qbound != nullptr && ks == xask && tdone > 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        xv[j] = __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    xpending = true;
                }
                if (tile_end) {
                    const unsigned p = tdone & 1u;
                    if (tdone == 0 && !(FR_DBG & 128)) {    // the streaming waves work out the first thresholds
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_s_barrier();
                    }
                    for (;;) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();       // A: this round's entries are in the queue
                        ++chk;
                        const bool again = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_read_u32(a_ovf)) == chk;
                        if (!again) break;
                        n = queued(p); i = 0;
#if FR_DBG & 64
                        FR_STAMP(sl_a);
#endif
                        while (__any(i < n)) drain_round(p);
#if FR_DBG & 64
                        FR_STAMP(sl_b);
                        sl_drain += sl_b - sl_a;
#endif
                        lds_write_u32(a_cnt + p * 256 + ql * 4, 0u);
                        publish();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();       // B: thresholds are out, the queue is empty
                    }
                    if (tdone == 0 && !(FR_DBG & 128) && chk == 1) {
                        // First tile without an overflow: the streaming waves have ranked its queue (see
                        // there) and left every query's best K entries, in order, in the start-list table.
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
#pragma unroll
                        for (int j = 0; j < K; ++j) key[j] = lds_read_u64(a_init + (ql * K + (unsigned)j) * 8);
                        lds_write_u32(a_cnt + ql * 4, 0u);
                        publish();
                        // the first block of every class writes its best to the class's slot at once: the
                        // others read it early in the second tile
                        const unsigned hk = (unsigned)(key[0] >> 40);
                        if (qbound != nullptr && (!WORK || qq < bql) && myturn == 0 && hk != 0u)
                            __hip_atomic_store(slots + myslot, (gen << 24) | (unsigned long long)hk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        n = 0; i = 0; pend = -1;
                    } else {
                        n = queued(p); i = 0; pend = (int)p;
                    }
                    ++tdone; ks = 0;
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    ++ks;
                }
            }
            if (pend >= 0)
                while (__any(i < n)) drain_round((unsigned)pend);
        }
        if (q0 + (int)ql < bql) {
            const size_t o = ((size_t)gb * Bq + q0 + ql) * K;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const unsigned h = (unsigned)(key[j] >> 32);
                part_s[o + j] = h ? ordered_value(h) : -INFINITY;
                part_r[o + j] = h ? (long long)(~(unsigned)key[j]) : -1;
            }
        }
#if FR_DBG & 64
        FR_STAMP(sl_a);
        if (lane == 0 && blockIdx.x < 256) {
            unsigned long long* const o = g_scan_stamp + blockIdx.x * 16;
            o[3] = sl_a; o[6] = sl_rounds; o[9] = sl_drain;
        }
#endif
        return;
    }

    // ------------------------------------------------------------ streaming waves
    if (S <= 0) return;
#if FR_DBG & 64
    unsigned long long sc_t0 = 0, sc_t1 = 0, sc_t2 = 0, sc_ta = 0, sc_tb = 0, sc_sel = 0, sc_n = 0, sc_boot = 0, sc_ovr = 0;
    FR_STAMP(sc_t0);
#endif
    const int cb = ((l7 ^ (lane >> 3)) << 4);
    const unsigned xo0 = (unsigned)((8 * wave + (lane >> 3)) * D * 2 + cb);
    const unsigned xg0 = (unsigned)((8 * wave + (lane >> 3)) * 128 + cb);
    const unsigned yq0 = (unsigned)q0 * (unsigned)D * 2u;
    const unsigned yomax = (unsigned)(bql - 1) * (unsigned)D * 2u + (unsigned)cb;
    float4v acc[NY][NX];
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[y][x] = float4v{0.f, 0.f, 0.f, 0.f};

    long itile = gb; int iks = 0;   // next gallery step to issue
    int yks = 0;                    // next query slice to issue
    auto issue_x = [&](int slot) __attribute__((always_inline)) {
        const long ptile = (WORK && tile_list) ? (long)tile_list[__builtin_amdgcn_readfirstlane((int)itile)] : itile;
        const char* const base = (const char*)gal + ((size_t)ptile * nk + (size_t)iks) * 32768;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            unsigned o = xg0;
            asm volatile("" : "+v"(o));
            if (qtiles == 1) glds16_nt(base + (size_t)i * 8192 + o, xs + slot * XB + (wave + 8 * i) * 1024);
            else glds16(base + (size_t)i * 8192 + o, xs + slot * XB + (wave + 8 * i) * 1024);
        }
        if (++iks == nk) { iks = 0; itile += nb; }
    };
    auto issue_y = [&](int slot) __attribute__((always_inline)) {
        unsigned o = xo0;
        asm volatile("" : "+v"(o));
        o += yq0;
        o = o < yomax ? o : yomax;
        glds16((const char*)q + (size_t)yks * 128 + o, ys + slot * YB + wave * 1024);
        if (++yks == nk) yks = 0;
    };
#pragma unroll
    for (int i = 0; i < DY; ++i)
        if (i < S) issue_y(i);
#pragma unroll
    for (int i = 0; i < DX; ++i)
        if (i < S) issue_x(i);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if FR_DBG & 64
    FR_STAMP(sc_t1);
#endif

    half8 fx[NX], fy[NY];
    int xslot = 0, yslot = 0, ks = 0;
    long tile = gb;
    unsigned tdone = 0, chk = 0;
    const unsigned qa = (unsigned)((wy * TY + l15) * 4);     // byte offset of query (y = 0) in thr / cnt
    for (long s = 0; s < S; ++s) {
        const char* const xbuf = xs + xslot * XB;
        const char* const ybuf = ys + yslot * YB;
        auto unit = [&](auto uc) __attribute__((always_inline)) {
            constexpr int kk = decltype(uc)::value;
            if (kk == 0) {
                if (s + DY < S) issue_y(yslot == 0 ? YS - 1 : yslot - 1);
            } else {
                if (s + DX < S) issue_x(xslot == 0 ? XS - 1 : xslot - 1);
            }
            const int coff = (((kk * 4 + g0) ^ l7) << 4);
#pragma unroll
            for (int y = 0; y < NY; ++y) fy[y] = *(const half8*)(ybuf + (wy * TY + y * 16 + l15) * 128 + coff);
#pragma unroll
            for (int x = 0; x < NX; ++x) fx[x] = *(const half8*)(xbuf + (wx * TX + x * 16 + l15) * 128 + coff);
            if (kk == 0 && ks == 0) {       // a tile's first MFMAs start from zero
                const float4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int y = 0; y < NY; ++y)
#pragma unroll
                    for (int x = 0; x < NX; ++x)
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], zero, 0, 0, 0);
            } else {
#pragma unroll
                for (int y = 0; y < NY; ++y)
#pragma unroll
                    for (int x = 0; x < NX; ++x)
                        acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[x], fy[y], acc[y][x], 0, 0, 0);
            }
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
        const bool tile_end = ks == nk - 1;
        // Gallery step s+1 and query slice s+1 must have landed before the next step reads them.
        auto wait_ring = [&]() __attribute__((always_inline)) {
            if (s + DX < S && s + DY < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (!tile_end) {
            ++ks;
            wait_ring();
        } else {
#if FR_DBG & 64
            FR_STAMP(sc_ta);
#endif
            const long ptile = (WORK && tile_list) ? (long)tile_list[__builtin_amdgcn_readfirstlane((int)tile)] : tile;
            const int rbase = (int)(ptile * 256) + wx * TX + g0 * 4;    // rows fit 31 bits (checked at launch)
            const unsigned p = tdone & 1u;
#if FR_DBG & 128
            const bool boot = false;        // diagnostic: nothing is ever selected (the streaming floor)
#else
            const bool boot = tdone == 0;
#endif
            float thr[NY];
            static_assert(NY == 2, "two query columns per lane");
            auto read_thr = [&]() __attribute__((always_inline)) {
                unsigned t0, t1;
                lds_read_2u32_64(a_thr + qa, t0, t1);
                thr[0] = ordered_value(t0); thr[1] = ordered_value(t1);
            };
            read_thr();
#if FR_DBG & 128
#pragma unroll
            for (int y = 0; y < NY; ++y) thr[y] = INFINITY;
#endif
            // Nothing is queued unless some lane's best raw score reaches its query's threshold
            // (striking rows out only lowers scores): the usual tile ends here, no key fetched.
            bool some = boot;
#pragma unroll
            for (int y = 0; y < NY; ++y) {
                const float4v (&a)[NX] = acc[y];
                float mq = max3_raw(a[0][0], a[0][1], a[0][2]);
                mq = max3_raw(mq, a[0][3], a[0][3]);
#pragma unroll
                for (int x = 1; x < NX; ++x) {
                    mq = max3_raw(mq, a[x][0], a[x][1]);
                    mq = max3_raw(mq, a[x][2], a[x][3]);
                }
                some = some || __any(mq >= thr[y]);
            }
            if (some) {
                if (USE_KEYS) {
                    // Key test: (key ^ kx) <= km as unsigned, with (kx, km) = (query key, 0) for an exact
                    // match and (0, 0x7fffffff) for "any live row" (tombstones are negative).
                    int4 kv[NX];
#pragma unroll
                    for (int x = 0; x < NX; ++x) kv[x] = *(const int4*)(gkeys + rbase + x * 16);   // padded allocation: in bounds
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        const int qq = q0 + wy * TY + y * 16 + l15;
                        const int mykey = (qkeys && qq < bql) ? qkeys[qq] : -2;
                        const unsigned kx = mykey == -2 ? 0u : (mykey < 0 ? 0x80000000u : (unsigned)mykey);
                        const unsigned km = mykey == -2 ? 0x7fffffffu : 0u;
#pragma unroll
                        for (int x = 0; x < NX; ++x) {
                            const unsigned k4[4] = {(unsigned)kv[x].x, (unsigned)kv[x].y, (unsigned)kv[x].z, (unsigned)kv[x].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[y][x][j] = ((k4[j] ^ kx) <= km) ? acc[y][x][j] : -INFINITY;
                        }
                    }
                }
                if ((ptile + 1) * 256 > nrows) {
                    const int nr = (int)nrows;
#pragma unroll
                    for (int y = 0; y < NY; ++y)
#pragma unroll
                        for (int x = 0; x < NX; ++x)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[y][x][j] = (rbase + x * 16 + j < nr) ? acc[y][x][j] : -INFINITY;
                }
            }
            wait_ring();
#if FR_DBG & 64
            unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0;
            FR_STAMP(sb0);
#endif
            if (boot) {
                // A block's first tile: the 16 lists of a query (4 lanes x 4 waves) each put their best
                // score into a table (the second queue's memory, unused until the second tile's end);
                // after a barrier every list counts how many of the 16 reach its own: the largest
                // score that at least K do is reached by K different rows, a first threshold.
                unsigned mk[NY];
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    const float4v (&v)[NX] = acc[y];
                    float mq = fmaxf(fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[0][2], v[0][3])),
                                     fmaxf(fmaxf(v[1][0], v[1][1]), fmaxf(v[1][2], v[1][3])));
                    mq = fmaxf(mq, fmaxf(fmaxf(fmaxf(v[2][0], v[2][1]), fmaxf(v[2][2], v[2][3])),
                                         fmaxf(fmaxf(v[3][0], v[3][1]), fmaxf(v[3][2], v[3][3]))));
                    mk[y] = mq == mq ? ordered_key(mq) : 0u;          // (a NaN counts as no score)
                    lds_write_u32(a_boot + (qa + (unsigned)(y * 64)) * 16 + (unsigned)((wx * 4 + g0) * 4), mk[y]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#if FR_DBG & 64
                FR_STAMP(sb1);
#endif
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    unsigned k16[16];
                    lds_read_16u32(a_boot + (qa + (unsigned)(y * 64)) * 16, k16);
                    unsigned cnt = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) cnt += k16[j] >= mk[y] ? 1u : 0u;
                    if (cnt >= (unsigned)K && mk[y] >= LOWEST_KEY) lds_max_u32(a_thr + qa + (unsigned)(y * 64), mk[y]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                read_thr();
#if FR_DBG & 64
                FR_STAMP(sb2);
#endif
            }
            for (;;) {
                bool blocked = false;
                if (some) {
#pragma unroll
                    for (int y = 0; y < NY; ++y) {
                        float4v (&v)[NX] = acc[y];
                        const float t = thr[y];
                        // one reservation per lane: as many consecutive entries as it has survivors
                        unsigned c = 0;
#pragma unroll
                        for (int x = 0; x < NX; ++x)
#pragma unroll
                            for (int j = 0; j < 4; ++j) c += v[x][j] >= t ? 1u : 0u;
                        if (!__any(c != 0u)) continue;
                        unsigned off = 0;
                        if (c != 0u) off = lds_add_rtn_u32(a_cnt + p * 256 + qa + (unsigned)(y * 64), c);
                        const unsigned ea = a_queue + p * (CAP * 512) + (qa + (unsigned)(y * 64)) * 2;
#pragma unroll
                        for (int x = 0; x < NX; ++x)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool pred = v[x][j] >= t;
                                if (__any(pred)) {
                                    if (pred) {
                                        if (off < (unsigned)CAP) {
                                            // {ordered key of the score, ~row}: the selector's list entry as it stands
                                            const unsigned row = (unsigned)(rbase + x * 16 + j);
                                            lds_write_u64(ea + off * 512, ((unsigned long long)ordered_key(v[x][j]) << 32) |
                                                                              (unsigned long long)(~row));
                                            v[x][j] = -INFINITY;
                                        } else {
                                            blocked = true;
                                        }
                                        ++off;
                                    }
                                }
                            }
                    }
                }
                if (__any(blocked)) lds_write_u32(a_ovf, chk + 1u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();               // A
                __builtin_amdgcn_sched_barrier(0);
                ++chk;
                const bool again = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_read_u32(a_ovf)) == chk;
                if (!again) break;
#if FR_DBG & 64
                ++sc_ovr;
#endif
                __builtin_amdgcn_s_barrier();               // B: the selector has drained and published
                asm volatile("" ::: "memory");
                read_thr();
            }
#if FR_DBG & 64
            FR_STAMP(sb3);
#endif
            if (boot && chk == 1) {
                // First tile, nothing overflowed: the queue holds every row of the tile that reached the
                // first threshold, at most CAP per query.  Instead of the selector inserting them one by
                // one (the whole chip would wait: every block is at its first tile), they are RANKED in
                // parallel: wave w takes queries 8w..8w+7, eight lanes per query, lane (q, e) the entries
                // e, e+8, ...; an entry's rank is the number of the query's entries with a larger key
                // (keys differ: they hold the row), and ranks below K are the query's start list.
                const unsigned qv = (unsigned)(8 * wave + (lane >> 3));
                const unsigned cq = lds_read_u32(a_cnt + qv * 4);
                const unsigned nq = cq < (unsigned)CAP ? cq : (unsigned)CAP;
                constexpr int NE = CAP / 8;
                unsigned long long mine[NE];
                unsigned rank[NE];
#pragma unroll
                for (int t = 0; t < NE; ++t) {
                    const unsigned j = (unsigned)(lane & 7) + 8u * (unsigned)t;
                    const unsigned long long e = lds_read_u64(a_queue + ((j * 64) + qv) * 8);
                    mine[t] = j < nq ? e : 0ull;
                    rank[t] = 0u;
                }
                for (unsigned i4 = 0; __any(i4 < nq); i4 += 4) {
                    unsigned long long e[4];
                    lds_read_4u64(a_queue + ((i4 * 64) + qv) * 8, 512, e);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned long long eu = i4 + (unsigned)u < nq ? e[u] : 0ull;
#pragma unroll
                        for (int t = 0; t < NE; ++t) rank[t] += eu > mine[t] ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int t = 0; t < NE; ++t)
                    if (mine[t] != 0ull && rank[t] < (unsigned)K)
                        lds_write_u64(a_init + (qv * K + rank[t]) * 8, mine[t]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#if FR_DBG & 64
            FR_STAMP(sc_tb);
            if (tdone == 0) {
                sc_boot = sc_tb - sc_ta;
                if (tid == 0 && blockIdx.x < 256) {
                    unsigned long long* const o = g_scan_stamp + blockIdx.x * 16;
                    o[10] = sb0 - sc_ta; o[11] = sb1 - sb0; o[12] = sb2 - sb1; o[13] = sb3 - sb2; o[14] = sc_tb - sb3;
                }
            }
            sc_sel += sc_tb - sc_ta; ++sc_n;
            if (tid == 0 && blockIdx.x < 256 && tdone < 20) g_scan_tiles[blockIdx.x * 20 + tdone] = sc_tb - sc_t1;
#endif
            ++tdone;
            ks = 0; tile += nb;
        }
        if (!tile_end) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        xslot = xslot == XS - 1 ? 0 : xslot + 1;
        yslot = yslot == YS - 1 ? 0 : yslot + 1;
    }
#if FR_DBG & 64
    FR_STAMP(sc_t2);
    if (tid == 0 && blockIdx.x < 256) {
        unsigned long long* const o = g_scan_stamp + blockIdx.x * 16;
        o[0] = sc_t0; o[1] = sc_t1; o[2] = sc_t2; o[4] = sc_sel; o[5] = sc_n; o[7] = sc_boot; o[8] = sc_ovr;
    }
#endif
}

// Wave-wide unsigned max / min through DPP row shifts and row broadcasts: no LDS traffic, the
// result is uniform.  Lanes without a source keep `identity`.
FR_DEV unsigned wave_umax(unsigned v) {
#define FR_DPP_STEP(ctrl, rows)                                                                          \
    { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false); v = v > o ? v : o; }
    FR_DPP_STEP(0x111, 0xf) FR_DPP_STEP(0x112, 0xf) FR_DPP_STEP(0x114, 0xf) FR_DPP_STEP(0x118, 0xf)   // row_shr 1,2,4,8
    FR_DPP_STEP(0x142, 0xa) FR_DPP_STEP(0x143, 0xc)                                                   // row_bcast 15, 31
#undef FR_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
FR_DEV unsigned wave_umin(unsigned v) { return ~wave_umax(~v); }


// One wave per query: the best kout of nlists candidate lists of kin entries each.  Every list
// arrives SORTED (score descending, row ascending inside equal scores; unused tail entries have
// row -1): the scan's per-block lists and a shard's final lists both are.  So no candidate is
// ever inserted anywhere: lane l owns lists l, l+64, l+128, l+192, keeps the head of each, and a
// round picks the best head of the wave (max score, then min row, by DPP reductions) and advances
// that one list.  The lists are staged once in LDS, each lane reading back only what it wrote, as
// per-lane scratch that can be indexed by a position.  Tie order = ascending row.
template <int K>
__global__ __launch_bounds__(64) void topk_final_kernel(const float* __restrict__ cs,
                                                        const long long* __restrict__ cr, int nlists, int kin,
                                                        int Bq, int kout, long long row_base,
                                                        float* __restrict__ out_s, long long* __restrict__ out_r,
                                                        const int32_t* __restrict__ out_index,
                                                        const int32_t* __restrict__ nlists_q,
                                                        long stride_s, long stride_r) {
    // list p of query qi starts at element p * stride + qi * kin of cs / cr: stride = Bq * kin for
    // lists stored back to back, anything larger for lists that sit inside gathered blocks
    constexpr int NL = 4;                         // lists per lane: nlists <= 256 (checked at launch)
    if (nlists_q) nlists = nlists_q[blockIdx.x];  // this query's own number of lists (may be 0)
    __shared__ float ls[K * 64 * NL];
    __shared__ long long lr[K * 64 * NL];
    const int qi = blockIdx.x, lane = threadIdx.x;
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int p = lane + 64 * t;
        const size_t pl = (size_t)(p < nlists ? p : 0), qa = (size_t)qi * kin;
        const size_t as = pl * (size_t)stride_s + qa, ar = pl * (size_t)stride_r + qa;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool have = p < nlists && j < kin;
            ls[(j * NL + t) * 64 + lane] = have ? cs[as + j] : -INFINITY;
            lr[(j * NL + t) * 64 + lane] = have ? cr[ar + j] : -1;
        }
    }
    // heads: (ordered score key, row); an exhausted or empty list has key 0 and the largest row
    constexpr long long NOROW = 0x7fffffffffffffffLL;
    unsigned hk[NL]; long long hr[NL]; int pos[NL];
    auto load_head = [&](int t) __attribute__((always_inline)) {
        const long long r = pos[t] < K ? lr[(pos[t] * NL + t) * 64 + lane] : -1;
        const float v = pos[t] < K ? ls[(pos[t] * NL + t) * 64 + lane] : -INFINITY;
        hk[t] = r >= 0 ? ordered_key(v) : 0u;
        hr[t] = r >= 0 ? r : NOROW;
    };
#pragma unroll
    for (int t = 0; t < NL; ++t) { pos[t] = 0; load_head(t); }
    for (int o = 0; o < kout; ++o) {
        unsigned bk = hk[0]; long long br = hr[0]; int bt = 0;
#pragma unroll
        for (int t = 1; t < NL; ++t) {
            const bool take = hk[t] > bk || (hk[t] == bk && hr[t] < br);
            bk = take ? hk[t] : bk; br = take ? hr[t] : br; bt = take ? t : bt;
        }
        const unsigned kmax = wave_umax(bk);
        const bool c1 = bk == kmax;
        const unsigned hmin = wave_umin(c1 ? (unsigned)((unsigned long long)br >> 32) : 0xffffffffu);
        const bool c2 = c1 && (unsigned)((unsigned long long)br >> 32) == hmin;
        const unsigned lmin = wave_umin(c2 ? (unsigned)(unsigned long long)br : 0xffffffffu);
        const bool c3 = c2 && (unsigned)(unsigned long long)br == lmin;
        const long long rwin = (long long)(((unsigned long long)hmin << 32) | lmin);
        const bool found = kmax != 0u;
        if (lane == 0) {
            // `out_index` (or the query number itself) is the answer row this query's result goes to
            const size_t qo = out_index ? (size_t)out_index[qi] : (size_t)qi;
            out_s[qo * kout + o] = found ? ordered_value(kmax) : -INFINITY;
            out_r[qo * kout + o] = found ? rwin + row_base : -1;
        }
        if (!found) continue;                                    // uniform: every list is exhausted
        // one lane advances the list that supplied the winner (the lowest lane, should two hold it)
        const unsigned long long m = __ballot(c3);
        const bool me = c3 && lane == (int)__builtin_ctzll(m);
#pragma unroll
        for (int t = 0; t < NL; ++t)
            if (me && bt == t) { ++pos[t]; load_head(t); }
    }
}

#if FR_WINOGRAD
// ------------------------------------------------------ Winograd F(2x2, 3x3) ---
// PROTOTYPE, not part of the library (built with -DFR_WINOGRAD=1 into scripts/kbench only).
// Measured on conv4_2 at 256 faces it runs level with the direct kernel (0.84 against 0.83 ms)
// with outputs within 7.5e-4 of the layer maximum; with all global traffic removed from its
// chunk loop it would take 0.59 ms.  Per 256 MFMAs it moves 96 KiB from L2 (64 KiB of U, 32 KiB
// of patch) and about 220 KiB through LDS, against 32 KiB and 96 KiB for the direct kernel's
// same output: on this part the 2.25 x saved MFMAs are paid for in L2 and LDS bandwidth.
// 3x3 convolution with 2.25 x fewer multiplies: a 4x4 input window d of one channel becomes
// V = Bt d B (16 values), a 3x3 filter g becomes U = G g Gt (16 values, made once at model
// load), the 16 products are summed over the input channels position by position (16
// independent GEMMs over [tiles] x [Cin] x [Cout]) and a 2x2 output block is At M A.
//
// A block item is 64 tile slots (TR rows of HW/2 tiles, rows taken from a flat list over all
// images) by 64 output channels, all 16 positions, walked in chunks of 32 input channels.
// Wave w owns positions 2w and 2w+1 for the whole 64 x 64 item (128 accumulator registers).
//   U   never passes through LDS: every fragment is needed by exactly one wave, and the
//       weights are stored in fragment order [channel block][chunk][position][16-channel
//       group][lane][16 B], so a wave's 8 KiB per chunk are eight coalesced loads;
//   x   the item's input rows of one chunk (4 rows of HW+2 pixels per tile row, 64 B each)
//       arrive by LDS-DMA; within a 16-pixel piece the even pixels come first, so the four
//       pixels a group of four neighbouring tiles reads (two apart) lie in one 256-B row;
//   V   every thread transforms one (tile, 4 channels) window per chunk with packed fp16
//       adds and writes its 16 values into the V stage [position][tile][32 channels]
//       (64-B rows, 16-B chunk c at c ^ ((-(tile >> 2)) & 3): conflict-free ds_read_b128
//       fragments); two V stages, one patch stage.
// Output transform: wave w first folds its two positions into the two column sums it
// contributes to, the eight waves' partial sums meet in LDS (one output column at a time:
// 128 KiB), and every thread finishes 8 channels of one tile: row transform, bias, ReLU,
// optional 2x2 max (a tile IS a pooling window), one 16-B store per pixel.
template <int HW>
struct WinoGeo {
    static constexpr int TW = HW / 2;
    static constexpr int TR = 64 / TW < 8 ? 64 / TW : 8;
    static constexpr int PW = HW + 2;
    static constexpr int PR = (4 * PW + 15) / 16 * 16;
    static constexpr int NPIECE = TR * PR / 16;
    static_assert(TW <= 64 && NPIECE <= 32, "a patch stage is 32 one-KiB pieces");
};

FR_DEV void wino_lds_read_b64(half4& d, unsigned addr, int off) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(off) : "memory");
}

template <int OFF>
FR_DEV void wino_gload(half8& u, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(u) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

__global__ __launch_bounds__(256) void wino_weights_kernel(const half_t* __restrict__ w, half_t* __restrict__ U,
                                                           int cin, int cout) {
    // w [Cout][3][3][Cin] -> U [Cout/64][Cin/32][16][4][64 lanes][8]: G g Gt in float32, rounded once
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)cout * cin * 2;      // 16 positions x Cout x Cin / 8
    if (idx >= total) return;
    const int lane = (int)(idx & 63), ns = (int)((idx >> 6) & 3), pos = (int)((idx >> 8) & 15);
    const long rest = idx >> 12;
    const int nkc = cin / 32;
    const int kc = (int)(rest % nkc), nb = (int)(rest / nkc);
    const int co = nb * 64 + ns * 16 + (lane & 15), ci0 = kc * 32 + 8 * (lane >> 4);
    const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    const int xi = pos >> 2, nu = pos & 3;
    half8 o;
    for (int e = 0; e < 8; ++e) {
        float acc = 0.f;
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s)
                acc += Gm[xi][r] * Gm[nu][s] * (float)w[((long)(co * 3 + r) * 3 + s) * cin + ci0 + e];
        o[e] = (half_t)acc;
    }
    *(half8*)(U + idx * 8) = o;
}

template <int HW, int CIN, int COUT, bool POOL, int OPAD>
__global__ __launch_bounds__(512) void conv3x3_wino_kernel(const half_t* __restrict__ x,
                                                           const half_t* __restrict__ U,
                                                           const float* __restrict__ bias,
                                                           half_t* __restrict__ y, int batch) {
    using Geo = WinoGeo<HW>;
    constexpr int TW = Geo::TW, TR = Geo::TR, PW = Geo::PW, PR = Geo::PR;
    constexpr int NKC = CIN / 32, NB = COUT / 64, HT = HW / 2;
    constexpr int VBUF = 65536, PATCH = 2 * VBUF;
    static_assert(CIN % 64 == 0 && COUT % 64 == 0, "channel tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int RT = batch * HT;
    const int nmb = (RT + TR - 1) / TR;
    const long nitems = (long)nmb * NB;
    const int G = gridDim.x;
    const int wk = xcd_remap(blockIdx.x, G);

    // this thread's transform task: tile slot tm, channels 4*tq .. 4*tq+3 of the chunk
    const int tm = tid >> 3, tq = tid & 7;
    const bool tlive = tm < TR * TW;
    const int trow = tlive ? tm / TW : 0, tcol = tlive ? tm % TW : 0;
    unsigned paddr[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int PP = trow * PR + i * PW + 2 * tcol + 2 * jp;     // even; the odd neighbour is 512 B further
            paddr[i][jp] = lds0 + PATCH + (PP >> 4) * 1024 + ((PP >> 1) & 7) * 64 + (tq >> 1) * 16 + (tq & 1) * 8;
        }
    const unsigned vw = lds0 + tm * 64 + ((((tq >> 1) ^ ((0 - (tm >> 2)) & 3))) << 4) + (tq & 1) * 8;
    const unsigned vr = lds0 + (2 * wave) * 4096 + l15 * 64 + (((g ^ ((0 - (l15 >> 2)) & 3))) << 4);

    float4v acc[2][4][4];       // [position][channel group][tile group]
    half8 ufr[2][4];
    half4 d[4][4];

    // Which items a block walks.  The transformed weights of one 64-channel block are 32 * CIN
    // bytes per position, 16 positions: with COUT = 512 all eight blocks together (8 MiB at
    // CIN = 512) do not fit an XCD's 4 MiB L2, and every U fragment would come from the Infinity
    // Cache.  So each XCD serves TWO channel blocks (its quarter of U stays in its L2) and half
    // of the tile rows; the two items of a tile-row group run side by side on two of its CUs
    // and share the patch in L2.
    const int per = G >> 3, xcd = wk / (per > 0 ? per : 1), local = wk % (per > 0 ? per : 1);
    const bool split = NB == 8 && (G & 7) == 0;
    const int mbh = (nmb + 1) / 2;
    const int mb_lo = split ? (xcd >> 2) * mbh : 0;
    const int mb_hi = split ? (mb_lo + mbh < nmb ? mb_lo + mbh : nmb) : nmb;
    const long first = split ? local : wk, stride = split ? per : G;
    const long count = split ? (long)(mb_hi - mb_lo) * 2 : nitems;
    for (long item = first; item < count; item += stride) {
        const int mb = split ? mb_lo + (int)(item >> 1) : (int)(item / NB);
        const int nb = split ? 2 * (xcd & 3) + (int)(item & 1) : (int)(item % NB);
        const int R0 = mb * TR;
        unsigned xo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = wave + 8 * i;
            const int row = lane >> 2, c = lane & 3;
            const int PP = n * 16 + 2 * (row & 7) + (row >> 3);
            int r = PP / PR, P = PP % PR;
            r = r < TR ? r : TR - 1;
            P = P < 4 * PW ? P : 4 * PW - 1;
            const int ii = P / PW, px = P % PW;
            int R = R0 + r;
            R = R < RT ? R : RT - 1;
            const int b = R / HT, ty = R % HT;
            xo[i] = (unsigned)((((b * (HW + 2) + 2 * ty + ii) * PW + px) * CIN) * 2 + c * 16);
        }
        // this wave's fragments of chunk 0, position 2*wave: 4 KiB per position, 64 KiB per chunk
        const char* const ubase = (const char*)U + ((size_t)nb * NKC * 16 + 2 * wave) * 4096;
        const unsigned uoff = lane * 16;

        auto issue_patch = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned o = xo[i];
                asm volatile("" : "+v"(o));
                glds16((const char*)x + kc * 64 + o, smem + PATCH + (wave + 8 * i) * 1024);
            }
        };
        // Written as the instructions themselves: the compiler's own wait for a loaded value would
        // be vmcnt(0) inside the loop, which also drains the patch DMA issued after it.
        auto load_u = [&](int kc, int p) __attribute__((always_inline)) {
            const char* const sb = ubase + (size_t)kc * 65536 + p * 4096;
            asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                         "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                         "global_load_dwordx4 %3, %4, %5 offset:3072"
                         : "=&v"(ufr[p][0]), "=&v"(ufr[p][1]), "=&v"(ufr[p][2]), "=&v"(ufr[p][3])
                         : "v"(uoff), "s"(sb) : "memory");
        };
        auto load_u1 = [&](int kc, int p, int ns) __attribute__((always_inline)) {
            const char* const sb = ubase + (size_t)kc * 65536 + p * 4096;
            if (ns == 0) wino_gload<0>(ufr[p][0], uoff, sb);
            else if (ns == 1) wino_gload<1024>(ufr[p][1], uoff, sb);
            else if (ns == 2) wino_gload<2048>(ufr[p][2], uoff, sb);
            else wino_gload<3072>(ufr[p][3], uoff, sb);
        };
        auto wait_u = [&](int p) __attribute__((always_inline)) {       // all but the newest eight vector-memory ops
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(ufr[p][0]), "+v"(ufr[p][1]), "+v"(ufr[p][2]), "+v"(ufr[p][3]) :: "memory");
        };
        auto transform_read = [&](int i0 = 0, int i1 = 4) __attribute__((always_inline)) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j & 1) wino_lds_read_b64(d[i][j], paddr[i][j >> 1], 512);
                    else wino_lds_read_b64(d[i][j], paddr[i][j >> 1], 0);
                }
        };
        auto transform_wait = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]),
                           "+v"(d[1][2]), "+v"(d[1][3]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[2][2]), "+v"(d[2][3]),
                           "+v"(d[3][0]), "+v"(d[3][1]), "+v"(d[3][2]), "+v"(d[3][3])
                         :: "memory");
        };
        auto transform_cols = [&](int j0 = 0, int j1 = 4) __attribute__((always_inline)) {       // Bt d, in place
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                const half4 a = d[0][j] - d[2][j], b = d[1][j] + d[2][j], c = d[2][j] - d[1][j], e = d[1][j] - d[3][j];
                d[0][j] = a; d[1][j] = b; d[2][j] = c; d[3][j] = e;
            }
        };
        auto transform_rows = [&](int i0 = 0, int i1 = 4) __attribute__((always_inline)) {       // (Bt d) B, in place
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const half4 a = d[i][0] - d[i][2], b = d[i][1] + d[i][2], c = d[i][2] - d[i][1], e = d[i][1] - d[i][3];
                d[i][0] = a; d[i][1] = b; d[i][2] = c; d[i][3] = e;
            }
        };
        auto transform_write = [&](int buf, int i0 = 0, int i1 = 4) __attribute__((always_inline)) {
            const unsigned a = vw + buf * VBUF;
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(d[i][j]), "n"((i * 4 + j) * 4096) : "memory");
        };
        half8 vf[2][4];
        auto read_v = [&](int buf, int p) __attribute__((always_inline)) {
            const unsigned a = vr + buf * VBUF + p * 4096;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072"
                         : "=&v"(vf[p][0]), "=&v"(vf[p][1]), "=&v"(vf[p][2]), "=&v"(vf[p][3]) : "v"(a) : "memory");
        };
        auto wait_v = [&](int p, int newer) __attribute__((always_inline)) {     // `newer` LDS ops may stay in flight
            if (newer == 0)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[p][0]), "+v"(vf[p][1]), "+v"(vf[p][2]), "+v"(vf[p][3]) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vf[p][0]), "+v"(vf[p][1]), "+v"(vf[p][2]), "+v"(vf[p][3]) :: "memory");
        };
        auto mma = [&](int p, int n0 = 0, int n1 = 4) __attribute__((always_inline)) {
#pragma unroll
            for (int ns = n0; ns < n1; ++ns)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
                    acc[p][ns][ms] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufr[p][ns], vf[p][ms], acc[p][ns][ms], 0, 0, 0);
        };
        auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ns = 0; ns < 4; ++ns)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) acc[p][ns][ms] = float4v{0.f, 0.f, 0.f, 0.f};

#if FR_DBG & 256
        unsigned long long ws[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wt0, wt1;
        FR_STAMP(wt0);
#define FR_WSTAMP(i) do { FR_STAMP(wt1); ws[i] += wt1 - wt0; wt0 = wt1; } while (0)
#else
#define FR_WSTAMP(i) do { } while (0)
#endif
        // prologue: chunk 0 transformed into V stage 0, chunk 1's patch and chunk 0's U in flight
        issue_patch(0);
        load_u(0, 0);
        load_u(0, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // the patch
        __builtin_amdgcn_s_barrier();
        fence();
        transform_read();
        transform_wait();
        transform_cols();
        transform_rows();
        if (tlive) transform_write(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        fence();
        issue_patch(NKC > 1 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ufr[0][0]), "+v"(ufr[0][1]), "+v"(ufr[0][2]), "+v"(ufr[0][3]),
                     "+v"(ufr[1][0]), "+v"(ufr[1][1]), "+v"(ufr[1][2]), "+v"(ufr[1][3]) :: "memory");
        __builtin_amdgcn_s_barrier();
        fence();
        // Every chunk issues the same vector-memory operations in the same order (four patch pieces,
        // then four U fragments for each of the two positions), the last chunks with the chunk index
        // held at NKC - 1 (what they fetch is never used): the waits below can then always be "all but
        // the newest eight".
        FR_WSTAMP(0);
        // Vector-memory order of a chunk, per wave: four U fragments of position 0 (one behind each
        // group of four MFMAs that frees its register), after the mid barrier the four patch pieces,
        // then four U fragments of position 1.  The last chunks fetch with the chunk index held at
        // NKC - 1 (never used), so the counted waits are the same in every chunk.
        for (int kc = 0; kc < NKC; ++kc) {
            const int buf = kc & 1;
            const bool more = kc + 1 < NKC;
            const int kn = more ? kc + 1 : kc, kp = kc + 2 < NKC ? kc + 2 : NKC - 1;
            read_v(buf, 0);
            if (more && !(FR_WABL & 32)) transform_read(0, 2);
            fence();
            if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[0][2]), "+v"(vf[0][3]) :: "memory");
            else wait_v(0, 0);
            // in flight at most: position 1's fragments of this chunk
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(ufr[0][0]), "+v"(ufr[0][1]), "+v"(ufr[0][2]), "+v"(ufr[0][3]) :: "memory");
            fence();
            FR_WSTAMP(1);
            if (!(FR_WABL & 16)) mma(0, 0, 1);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 0, 0);
            if (!(FR_WABL & 16)) mma(0, 1, 2);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 0, 1);
            if (more && !(FR_WABL & 32)) transform_read(2, 4);
            read_v(buf, 1);
            fence();
            if (!(FR_WABL & 16)) mma(0, 2, 3);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 0, 2);
            if (!(FR_WABL & 16)) mma(0, 3, 4);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 0, 3);
            fence();
            FR_WSTAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]),
                           "+v"(d[1][2]), "+v"(d[1][3]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[2][2]), "+v"(d[2][3]),
                           "+v"(d[3][0]), "+v"(d[3][1]), "+v"(d[3][2]), "+v"(d[3][3]),
                           "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[1][2]), "+v"(vf[1][3])
                         :: "memory");
            fence();
            FR_WSTAMP(3);
            __builtin_amdgcn_s_barrier();   // every wave has read the patch: it may be refilled
            fence();
            FR_WSTAMP(4);
            if (!(FR_WABL & 8)) issue_patch(kp);
            fence();
            wait_u(1);                      // behind position 1's fragments: four of the next chunk, the patch
            fence();
            FR_WSTAMP(5);
            if (!(FR_WABL & 16)) mma(1, 0, 1);
            if (more) if (!(FR_WABL & 1)) transform_cols(0, 2);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 1, 0);
            if (!(FR_WABL & 16)) mma(1, 1, 2);
            if (more) if (!(FR_WABL & 1)) transform_cols(2, 4);
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 1, 1);
            if (!(FR_WABL & 16)) mma(1, 2, 3);
            if (more) { if (!(FR_WABL & 1)) transform_rows(0, 2); if (tlive && !(FR_WABL & 2)) transform_write(buf ^ 1, 0, 2); }
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 1, 2);
            if (!(FR_WABL & 16)) mma(1, 3, 4);
            if (more) { if (!(FR_WABL & 1)) transform_rows(2, 4); if (tlive && !(FR_WABL & 2)) transform_write(buf ^ 1, 2, 4); }
            fence();
            if (!(FR_WABL & 4)) load_u1(kn, 1, 3);
            fence();
            FR_WSTAMP(6);
            // all but position 1's new fragments: the patch has landed; this wave's V writes are done
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            FR_WSTAMP(7);
            __builtin_amdgcn_s_barrier();
            fence();
            FR_WSTAMP(8);
        }
        // the loads of the last chunk land in registers that are free from here on: wait for them
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ufr[0][0]), "+v"(ufr[0][1]), "+v"(ufr[0][2]), "+v"(ufr[0][3]),
                     "+v"(ufr[1][0]), "+v"(ufr[1][1]), "+v"(ufr[1][2]), "+v"(ufr[1][3]) :: "memory");
        fence();

        // output transform
        const int T = tid >> 3, o8 = tid & 7;
        const int er = T / TW, et = T % TW;
        const int ER = R0 + er;
        const bool elive = T < TR * TW && ER < RT;
        const int eb = elive ? ER / HT : 0, ety = elive ? ER % HT : 0;
        const float4v b0 = *(const float4v*)(bias + nb * 64 + o8 * 8), b1 = *(const float4v*)(bias + nb * 64 + o8 * 8 + 4);
        constexpr int HO = POOL ? HT : HW, OW = HO + 2 * OPAD;
        float pooled[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int ns = 0; ns < 4; ++ns)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) {
                    const float4v a0 = acc[0][ns][ms], a1 = acc[1][ns][ms];
                    float4v z;
                    if ((wave & 1) == 0) z = j == 0 ? a0 + a1 : a1;
                    else z = j == 0 ? a0 : -a0 - a1;
                    *(float4v*)(smem + wave * 16384 + (ms * 16 + l15) * 256 + (((ns * 4 + g) ^ l15) << 4)) = z;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float zs[4][8];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int ch = ((2 * o8 + hh) ^ (T & 15)) << 4;
                    const float4v u0 = *(const float4v*)(smem + (2 * xi) * 16384 + T * 256 + ch);
                    const float4v u1 = *(const float4v*)(smem + (2 * xi + 1) * 16384 + T * 256 + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) zs[xi][hh * 4 + e] = u0[e] + u1[e];
                }
            }
            half8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float bb = e < 4 ? b0[e] : b1[e - 4];
                const float y0 = fmaxf(zs[0][e] + zs[1][e] + zs[2][e] + bb, 0.f);
                const float y1 = fmaxf(zs[1][e] - zs[2][e] - zs[3][e] + bb, 0.f);
                if (POOL) {
                    const float m = fmaxf(y0, y1);
                    pooled[e] = j == 0 ? m : fmaxf(pooled[e], m);
                    o0[e] = (half_t)pooled[e];
                } else {
                    o0[e] = (half_t)y0;
                    o1[e] = (half_t)y1;
                }
            }
            if (elive) {
                if (POOL) {
                    if (j == 1)
                        *(half8*)(y + (((long)eb * OW + ety + OPAD) * OW + et + OPAD) * COUT + nb * 64 + o8 * 8) = o0;
                } else {
                    half_t* const p0 = y + (((long)eb * OW + 2 * ety + OPAD) * OW + 2 * et + j + OPAD) * COUT + nb * 64 + o8 * 8;
                    *(half8*)p0 = o0;
                    *(half8*)(p0 + (long)OW * COUT) = o1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        FR_WSTAMP(9);
#if FR_DBG & 256
        if (blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 5)) {
            ws[10] += 1;
            for (int i = 0; i < 11; ++i) g_stamp[(wave ? 16 : 0) + i] += ws[i];
        }
#endif
    }
}

#endif  // FR_WINOGRAD

// ---------------------------------------------------------------- launch ---
#define FR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

template <int HW, int CIN, int COUT, bool POOL, int RY, int WX, int WY, bool FUSE1 = false>
static hipError_t launch_patch_inst(const void* x, const void* w, const float* bias, void* y, void* dump, int B,
                                    int reverse, hipStream_t s, bool set_attr_only, const void* w1 = nullptr,
                                    const float* b1 = nullptr, int blocks = 0) {
    auto kern = conv3x3_patch_kernel<HW, CIN, COUT, POOL, RY, WX, WY, FUSE1>;
    constexpr int patch2 = 2 * 41 * 1024, nks = 9 * CIN / 64, yb = RY * 128;
    constexpr bool res = nks * yb + patch2 <= 160 * 1024;
    constexpr int lds = patch2 + (res ? nks : 4) * yb;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (set_attr_only)
        return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    constexpr int NT = COUT / RY;
    const long ntiles = (long)B * (HW / 16) * (HW / 16) * NT;
    long grid = blocks > 0 ? blocks / NT * NT : 256;     // one walker per CU; a multiple of COUT / RY
    if (grid > ntiles) grid = ntiles;      // ntiles is a multiple of NT
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, (const half_t*)x, (const half_t*)w, bias,
                       (half_t*)y, (half_t*)dump, B, reverse, (const half_t*)w1, b1);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

template <int HW, int CIN, int COUT, bool POOL, int OPAD, int RY, int WX, int WY>
static hipError_t launch_conv_inst(const void* x, const void* w, const float* bias, void* y, void* dump, int B,
                                   int reverse, hipStream_t s, bool set_attr_only, int blocks = 0) {
    using Core = GemmCore8<256, RY, WX, WY, false>;
    auto kern = conv3x3_ring_kernel<HW, CIN, COUT, POOL, OPAD, RY, WX, WY>;
    constexpr int lds = Core::LDS;
    if (set_attr_only)
        return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long M = (long)B * HW * HW;
    // the kernel addresses its input with 32-bit byte offsets per lane
    if ((unsigned long long)B * (HW + 2) * (HW + 2) * CIN * 2ull >= (1ull << 32)) return hipErrorInvalidValue;
    const long ntiles = ((M + 255) / 256) * (COUT / RY);
    long grid = blocks > 0 ? blocks / (COUT / RY) * (COUT / RY) : 256;     // one walker per CU; a multiple of COUT / RY
    if (grid > ntiles) grid = (ntiles + 1) & ~1L;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, (const half_t*)x, (const half_t*)w, bias,
                       (half_t*)y, (half_t*)dump, B, reverse);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

static hipError_t conv_dispatch(const ConvLayerDesc& d, const void* x, const void* w, const float* bias, void* y,
                                void* dump, int B, hipStream_t s, bool attr) {
#define FR_PATCH(HW, CIN, COUT, POOL, RY, WX, WY)                                                            \
    if (d.hw == HW && d.cin == CIN && d.cout == COUT && d.pool == (POOL ? 1 : 0) && d.opad == 1)              \
        return launch_patch_inst<HW, CIN, COUT, POOL, RY, WX, WY>(x, w, bias, y, dump, B, d.reverse, s, attr, nullptr, nullptr, d.blocks);
    FR_PATCH(224, 64, 64, true, 64, 4, 2)
    FR_PATCH(112, 64, 128, false, 128, 4, 2)
    FR_PATCH(112, 128, 128, true, 128, 4, 2)
#undef FR_PATCH
    // Ring layers: 256 pixels x 256 channels per tile.  A small batch leaves CUs without a tile
    // (the 14 x 14 layers have 0.77 pixel tiles per face); when at most half of them would get one,
    // the 128-channel tiling (twice the tiles, about 0.6 of the time each) is the faster launch:
    // 32 faces, 14 x 14 layers 95 -> 62 us each; at most 32 tiles (a face or a few) take 64-channel
    // tiles (1 face: the stack 0.45 -> 0.33 ms).  All tilings give bit-identical outputs: the K
    // order of every output element is the same.  (Replaying a small batch's launches from a
    // captured graph was measured too: 0.03 ms of 0.6 per call, not kept.)
#define FR_CONV(HW, CIN, COUT, POOL, OPAD)                                                                   \
    if (d.hw == HW && d.cin == CIN && d.cout == COUT && d.pool == (POOL ? 1 : 0) && d.opad == OPAD) {         \
        if (attr) {                                                                                           \
            hipError_t e = launch_conv_inst<HW, CIN, COUT, POOL, OPAD, 128, 4, 2>(x, w, bias, y, dump, B, d.reverse, s, attr); \
            if (e == hipSuccess)                                                                              \
                e = launch_conv_inst<HW, CIN, COUT, POOL, OPAD, 64, 4, 2>(x, w, bias, y, dump, B, d.reverse, s, attr); \
            if (e != hipSuccess) return e;                                                                    \
        } else if (((long)B * HW * HW + 255) / 256 * (COUT / 256) <= 32) {                                    \
            return launch_conv_inst<HW, CIN, COUT, POOL, OPAD, 64, 4, 2>(x, w, bias, y, dump, B, d.reverse, s, attr); \
        } else if (((long)B * HW * HW + 255) / 256 * (COUT / 256) <= 128) {                                   \
            return launch_conv_inst<HW, CIN, COUT, POOL, OPAD, 128, 4, 2>(x, w, bias, y, dump, B, d.reverse, s, attr); \
        }                                                                                                     \
        return launch_conv_inst<HW, CIN, COUT, POOL, OPAD, 256, 2, 4>(x, w, bias, y, dump, B, d.reverse, s, attr, d.blocks);     \
    }
    FR_CONV(56, 128, 256, false, 1)
    FR_CONV(56, 256, 256, false, 1)
    FR_CONV(56, 256, 256, true, 1)
    FR_CONV(28, 256, 512, false, 1)
    FR_CONV(28, 512, 512, false, 1)
    FR_CONV(28, 512, 512, true, 1)
    FR_CONV(14, 512, 512, false, 1)
    FR_CONV(14, 512, 512, true, 0)
#undef FR_CONV
    return hipErrorInvalidValue;
}

#if FR_WINOGRAD
template <int HW, int CIN, int COUT, bool POOL, int OPAD>
static hipError_t launch_wino_inst(const void* x, const void* U, const float* bias, void* y, int B, hipStream_t s,
                                   bool set_attr_only) {
    auto kern = conv3x3_wino_kernel<HW, CIN, COUT, POOL, OPAD>;
    constexpr int lds = 2 * 65536 + 32768;
    if (set_attr_only)
        return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if ((unsigned long long)B * (HW + 2) * (HW + 2) * CIN * 2ull >= (1ull << 31)) return hipErrorInvalidValue;
    using Geo = WinoGeo<HW>;
    const long items = (((long)B * (HW / 2) + Geo::TR - 1) / Geo::TR) * (COUT / 64);
    long grid = 256;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, (const half_t*)x, (const half_t*)U, bias,
                       (half_t*)y, B);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

static hipError_t wino_dispatch(const ConvLayerDesc& d, const void* x, const void* U, const float* bias, void* y, int B,
                                hipStream_t s, bool attr) {
#define FR_WINO(HW, CIN, COUT, POOL, OPAD)                                                          \
    if (d.hw == HW && d.cin == CIN && d.cout == COUT && d.pool == (POOL ? 1 : 0) && d.opad == OPAD) \
        return launch_wino_inst<HW, CIN, COUT, POOL, OPAD>(x, U, bias, y, B, s, attr);
    FR_WINO(28, 512, 512, false, 1)
#undef FR_WINO
    return hipErrorInvalidValue;
}

// U: 16 * cout * cin fp16 (see wino_weights_kernel)
hipError_t launch_wino_weights(const void* w, void* U, int cin, int cout, hipStream_t s) {
    if (!w || !U || cin % 32 || cout % 64) return hipErrorInvalidValue;
    const long total = (long)cout * cin * 2;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const half_t*)w,
                       (half_t*)U, cin, cout);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_conv3x3_wino(const ConvLayerDesc& d, const void* x, const void* U, const float* bias, void* y, int B,
                               hipStream_t s) {
    if (B <= 0 || !x || !U || !bias || !y) return hipErrorInvalidValue;
    return wino_dispatch(d, x, U, bias, y, B, s, false);
}

#endif  // FR_WINOGRAD

// `dump` is >= 1 KiB of scratch that dead lanes of a partial tile store to.
hipError_t launch_conv3x3(const ConvLayerDesc& d, const void* x, const void* w, const float* bias, void* y,
                          void* dump, int B, hipStream_t s) {
    if (B <= 0 || !x || !w || !bias || !y || !dump) return hipErrorInvalidValue;
    return conv_dispatch(d, x, w, bias, y, dump, B, s, false);
}

hipError_t launch_conv1_fused(const void* x0, const void* w64, const float* bias1, const void* w, const float* bias,
                              void* y, void* dump, int B, int reverse, hipStream_t s, int blocks) {
    if (B <= 0 || !x0 || !w64 || !bias1 || !w || !bias || !y || !dump || blocks < 0 || blocks > 256) return hipErrorInvalidValue;
    return launch_patch_inst<224, 64, 64, true, 64, 4, 2, true>(x0, w, bias, y, dump, B, reverse, s, false, w64, bias1,
                                                                blocks);
}

hipError_t launch_conv1_1(const void* x0, const void* w64, const float* bias, void* y, int B, hipStream_t s) {
    if (B <= 0 || !x0 || !w64 || !bias || !y) return hipErrorInvalidValue;
    const long ntiles = (long)B * 224 * 14;
    if (ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    const int blocks = (int)((ntiles + 3) / 4 < 4096 ? (ntiles + 3) / 4 : 4096);
    hipLaunchKernelGGL(conv1_1_kernel, dim3(blocks), dim3(256), 0, s, (const half_t*)x0, (const half_t*)w64, bias,
                       (half_t*)y, (int)ntiles);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

using FcCore = GemmCore<256, 128, 2, 2, false>;

hipError_t launch_fc_splitk(const void* x, const void* w, float* partial, int M, int N, int K, int splitk,
                            hipStream_t s) {
    if (M <= 0 || N <= 0 || N % 128 || splitk <= 0 || K % (64 * splitk) || !x || !w || !partial)
        return hipErrorInvalidValue;
    const int kper = K / 64 / splitk;
    if (M <= 64) {
        // a small batch: 64-row tiles, so that the step's loads are mostly weights (the 256-row tile
        // re-reads 32 KiB of clamped batch rows beside every 16 KiB of weights) and three blocks
        // share a CU.  Every output element sums its K slice in the same order either way.
        using Small = GemmCore<64, 128, 1, 4, false>;
        dim3 grid(N / 128, 1, splitk);
        hipLaunchKernelGGL((fc_splitk_kernel<64, 128, 1, 4>), grid, dim3(256), 2 * Small::SB, s, (const half_t*)x,
                           (const half_t*)w, partial, M, N, K, kper);
        FR_CHECK_LAUNCH();
        return hipSuccess;
    }
    if (M <= 256) {
        // up to a batch of 256: 128-row tiles, two blocks per CU (64 KiB of LDS each) instead of one of 96 KiB.  The
        // kernel's two-stage ring leaves one stage in flight, so a second block per CU is what hides the weights' HBM
        // latency: fc6 + fc7 of 256 faces 0.147 -> 0.134 ms, same bits (the K order of an output element is the same).
        using Mid = GemmCore<128, 128, 2, 2, false>;
        dim3 grid(N / 128, (M + 127) / 128, splitk);
        hipLaunchKernelGGL((fc_splitk_kernel<128, 128, 2, 2>), grid, dim3(256), 2 * Mid::SB, s, (const half_t*)x,
                           (const half_t*)w, partial, M, N, K, kper);
        FR_CHECK_LAUNCH();
        return hipSuccess;
    }
    dim3 grid(N / 128, (M + 255) / 256, splitk);
    hipLaunchKernelGGL((fc_splitk_kernel<256, 128, 2, 2>), grid, dim3(256), 2 * FcCore::SB, s, (const half_t*)x,
                       (const half_t*)w, partial, M, N, K, kper);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_fc_reduce_relu(const float* partial, const float* bias, void* y, int M, int N, int splitk,
                                 hipStream_t s) {
    if (M <= 0 || N <= 0 || N % 4 || !partial || !bias || !y) return hipErrorInvalidValue;
    const size_t total4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(fc_reduce_relu_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, partial, bias,
                       (half_t*)y, M, N, splitk);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_fc_reduce_relu_l2(const float* partial, const float* bias, float* out_f32, void* out_q, int Dq,
                                    int M, int N, int splitk, hipStream_t s) {
    if (M <= 0 || N != 4096 || Dq <= 0 || Dq > N || Dq % 4 || !partial || !bias) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fc_reduce_relu_l2_kernel, dim3(M), dim3(256), 0, s, partial, bias, out_f32, (half_t*)out_q, Dq,
                       M, N, splitk);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_resize_u8(const uint8_t* crops, const CropMeta* meta, const uint8_t* lut, int flip, uint8_t* out,
                            int* out_max, float scale, void* x0, int B, hipStream_t s) {
    if (B <= 0 || !crops || !meta || !lut || !out || !out_max || !x0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(out_max, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(resize_u8_kernel, dim3(49, B), dim3(256), 0, s, crops, meta, lut, flip, out, out_max,
                       scale, (half_t*)x0);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_crop_sharpness(const uint8_t* crops, const CropMeta* meta, int max_tiles, int64_t* out, int B,
                                 hipStream_t s) {
    if (B <= 0 || max_tiles <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)B * 16, s);
    if (e != hipSuccess) return e;
    // a block takes four strips at a time; enough blocks for a single large image
    const int blocks = (max_tiles + 3) / 4, want = B >= 32 ? 64 : 2048 / B;
    crop_sharpness_kernel<<<dim3(blocks < want ? blocks : want, B), 256, 0, s>>>(crops, meta,
                                                                                (unsigned long long*)out);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_normalize_pack(const uint8_t* u8, const int* img_max, float scale, void* x0, int B, hipStream_t s,
                                 bool only_undivided) {
    if (B <= 0 || !u8 || !img_max || !x0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(normalize_pack_kernel, dim3(only_undivided ? 4 : 196, B), dim3(256), 0, s, u8, img_max, scale,
                       (half_t*)x0, only_undivided ? 1 : 0);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_gallery_upsert(const float* vecs, const int32_t* keys, const int64_t* rows, int n, void* g,
                                 int32_t* gk, int D, hipStream_t s) {
    if (n <= 0 || !vecs || !g || D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_upsert_kernel, dim3((n + 3) / 4), dim3(256), 0, s, vecs, keys, rows, n, (half_t*)g, gk,
                       D);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_gallery_set_keys(const int64_t* rows, int n, int32_t key, int32_t* gk, hipStream_t s) {
    if (n <= 0 || !rows || !gk) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_set_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rows, n, key, gk);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_gallery_fill(void* g, int32_t* gk, int D, uint64_t seed, int64_t row0, int64_t n, int nkeys,
                               int dup_every, hipStream_t s) {
    if (n <= 0 || !g || !gk || D % 64 || row0 < 0) return hipErrorInvalidValue;
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (half_t*)g, gk, D, seed, row0, n,
                       nkeys, dup_every);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_gallery_gather(const void* g, const int32_t* gk, const int64_t* rows, int n, int D, void* out,
                                 int32_t* out_keys, hipStream_t s) {
    if (n <= 0 || !g || !rows || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const half_t*)g, gk, rows, n, D,
                       (half_t*)out, out_keys);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_gallery_scatter(const void* in, const int32_t* in_keys, const int64_t* rows, int n, int D, void* g,
                                  int32_t* gk, hipStream_t s) {
    if (n <= 0 || !in || !in_keys || !rows || !g || !gk || D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_scatter_kernel, dim3((n + 3) / 4), dim3(256), 0, s, (const half_t*)in, in_keys, rows, n,
                       D, (half_t*)g, gk);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_f32_to_unit_f16(const float* q, void* o, int Bq, int D, hipStream_t s) {
    if (Bq <= 0 || !q || !o || D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f32_to_unit_f16_kernel, dim3((Bq + 3) / 4), dim3(256), 0, s, q, (half_t*)o, Bq, D);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

#if FR_DBG & 256
hipError_t debug_stamps_read(unsigned long long* out64) { return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_stamp), 64 * 8); }
hipError_t debug_stamps_clear() {
    unsigned long long z[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof z);
}
#endif
#if FR_DBG & 64
hipError_t scan_debug_stamps(unsigned long long* out) {      // 256 x 16 values of the last scan launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_stamp), sizeof(unsigned long long) * 256 * 16);
}
hipError_t scan_debug_tiles(unsigned long long* out) {       // 256 x 20: selector kernel, tile end times
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_tiles), sizeof(unsigned long long) * 256 * 20);
}
#endif

int scan_max_blocks() { return 256; }

static int scan_klist(int k) { return k <= 5 ? 5 : (k <= 10 ? 10 : 16); }
int scan_list_len(int k) { return scan_klist(k); }

static int scan_query_tile(int Bq, int kl) {
    const int cap = kl == 5 ? 256 : (kl == 10 ? 128 : 64);
    const int want = Bq <= 64 ? 64 : (Bq <= 128 ? 128 : 256);
    return want < cap ? want : cap;
}

// Gallery walkers per query tile before the clamp to the gallery's tile count.
static int scan_walkers(int qtiles) {
    const int nb = (scan_max_blocks() / qtiles) & ~7;
    return nb < 8 ? 8 : nb;
}

size_t scan_part_entries(int Bq, int k) {
    const int kl = scan_klist(k);
    const int ry = scan_query_tile(Bq, kl);
    return (size_t)scan_walkers((Bq + ry - 1) / ry) * (size_t)Bq * (size_t)kl;
}

// Scratch of the search entry point that allocates nothing: the per-block candidate lists of the
// scan (rows, then scores) and the cross-block score bounds, one copy per GPU that loads this
// code object, held until the process ends.  kScanStaticEntries covers every request whose
// walkers x queries x list length stays at 256 blocks of 256 top-5 lists (up to 8192 queries of
// top-5, 4096 of top-10, 2048 of top-16); the bounds cover 8192 queries.  Host code never passes
// the array itself to a kernel or a memset: its name is a host-side stand-in, the device address
// comes from hipGetSymbolAddress, once per device, below.
constexpr size_t kScanStaticEntries = 256 * 256 * 5;
constexpr size_t kScanStaticBoundBytes = 8192 * 16 * sizeof(unsigned long long);
constexpr size_t kScanStaticBytes = kScanStaticEntries * (sizeof(int64_t) + sizeof(float)) + kScanStaticBoundBytes;
static __device__ __align__(256) unsigned char scratch[kScanStaticBytes];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];          // device address of `scratch` on each GPU (0: not resolved yet)
// Who used a GPU's array last, and an event behind that use: another owner may take the array over
// only once the event has completed (two owners' kernels would otherwise overlap on it).
static const void* g_scratch_owner[kMaxDevices];
static hipEvent_t g_scratch_done[kMaxDevices];
static std::mutex g_scratch_mu;

// Resolves the address of `scratch` on the CURRENT device (once per device).
static hipError_t scan_static_resolve() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev]) return hipSuccess;
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p || ((uintptr_t)p & 255u)) return hipErrorInvalidSymbol;
    e = hipEventCreateWithFlags(&g_scratch_done[dev], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    g_scratch_ptr[dev] = p;
    return hipSuccess;
}

hipError_t scan_static_claim(int device, const void* owner) {
    if (device < 0 || device >= kMaxDevices || !owner) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) return hipErrorNotInitialized;
    if (g_scratch_owner[device] && g_scratch_owner[device] != owner) {
        const hipError_t e = hipEventQuery(g_scratch_done[device]);
        if (e != hipSuccess) {                    // hipErrorNotReady: the other owner's work is still running
            (void)hipGetLastError();              // (not an error of a later launch on this thread)
            return e;
        }
    }
    g_scratch_owner[device] = owner;
    return hipSuccess;
}

hipError_t scan_static_mark(int device, const void* owner, hipStream_t s) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_owner[device] != owner) return hipErrorInvalidValue;
    return hipEventRecord(g_scratch_done[device], s);
}

hipError_t scan_static_scratch(int device, ScanStaticScratch* out) {
    if (!out || device < 0 || device >= kMaxDevices) return hipErrorInvalidValue;
    void* p;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        p = g_scratch_ptr[device];
    }
    if (!p) return hipErrorNotInitialized;        // kernels_init has not run with this device current
    unsigned char* const b = (unsigned char*)p;
    out->part_r = (int64_t*)b;
    out->part_s = (float*)(b + kScanStaticEntries * sizeof(int64_t));
    out->qbound = (unsigned*)(b + kScanStaticEntries * (sizeof(int64_t) + sizeof(float)));
    out->entries = kScanStaticEntries;
    out->bound_bytes = kScanStaticBoundBytes;
    return hipSuccess;
}

template <int K, int RY, int WX, int WY, bool USE_KEYS, bool WORK = false>
static hipError_t scan_inst(const void* g, const int32_t* gk, long nrows, int D, const void* q, const int32_t* qkeys,
                            int Bq, int nb, float* part_s, int64_t* part_r, unsigned* qbound, hipStream_t s,
                            bool attr_only, const int32_t* tile_list = nullptr, long ntl = 0,
                            const ScanWork* work = nullptr, int nwork = 0) {
    constexpr int xs = RY <= 64 ? 4 : 3, ys = RY <= 128 ? 3 : 2;
    constexpr int bounds = RY <= 64 ? RY * ((K + 3) & ~3) * 4 : 0;        // block-level score bounds
    constexpr int ring = xs * 256 * 128 + ys * RY * 128 + bounds;
    constexpr int merge = RY * WX * 4 * K * 8;
    constexpr int lds = ring > merge ? ring : merge;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = scan_topk_kernel<K, RY, WX, WY, USE_KEYS, WORK>;
    if (attr_only) return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int qtiles = (Bq + RY - 1) / RY;
    hipLaunchKernelGGL(kern, dim3((unsigned)(work ? nwork : nb * qtiles)), dim3(512), lds, s, (const half_t*)g, gk,
                       nrows, D, (const half_t*)q, qkeys, Bq, nb, part_s, (long long*)part_r, qbound, tile_list, ntl,
                       work);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

// The 64-query tile with a selector wave (scan_sel_kernel): rings of 3 gallery and 3 query stages,
// queues of 32 entries.
static int g_sel_variant = 1;            // 0: the kernel without a selector wave (A/B runs of the micro-benchmark)
void scan_set_sel_variant(int v) { g_sel_variant = v; }

// Launch number that tags the cross-block bound slots: 40 bits, never 0, different in every launch of
// the process (ONE counter for all instantiations: two launches on one slot array must never share it).
static unsigned long long scan_next_launch_number() {
    static std::atomic<unsigned long long> launches{0};
    unsigned long long g;
    do g = (launches.fetch_add(1) + 1) & ((1ull << 40) - 1); while (g == 0);
    return g;
}

template <int K, bool USE_KEYS, bool WORK, int XS, int YS, int CAP>
static hipError_t scan_sel_inst(const void* g, const int32_t* gk, long nrows, int D, const void* q,
                                const int32_t* qkeys, int Bq, int nb, float* part_s, int64_t* part_r,
                                unsigned* qbound, hipStream_t s, bool attr_only, const int32_t* tile_list = nullptr,
                                long ntl = 0, const ScanWork* work = nullptr, int nwork = 0) {
    constexpr int lds = XS * 256 * 128 + YS * 64 * 128 + 1024 + 2 * CAP * 64 * 8;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = scan_sel_kernel<K, USE_KEYS, WORK, XS, YS, CAP>;
    if (attr_only) return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int qtiles = (Bq + 63) / 64;
    const unsigned long long gen = scan_next_launch_number();
    hipLaunchKernelGGL(kern, dim3((unsigned)(work ? nwork : nb * qtiles)), dim3(576), lds, s, (const half_t*)g, gk,
                       nrows, D, (const half_t*)q, qkeys, Bq, nb, part_s, (long long*)part_r, qbound, tile_list, ntl,
                       work, gen);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

template <int K, bool USE_KEYS, bool WORK>
static hipError_t scan_sel_variants(const void* g, const int32_t* gk, long nrows, int D, const void* q,
                                    const int32_t* qkeys, int Bq, int nb, float* ps, int64_t* pr, unsigned* qb,
                                    hipStream_t s, bool attr, const int32_t* tile_list = nullptr, long ntl = 0,
                                    const ScanWork* work = nullptr, int nwork = 0) {
    if (attr) {
        hipError_t e = scan_sel_inst<K, USE_KEYS, WORK, 3, 3, 32>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, true);
        return e;
    }
    return scan_sel_inst<K, USE_KEYS, WORK, 3, 3, 32>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, false, tile_list,
                                                      ntl, work, nwork);
}

static hipError_t scan_dispatch(int kl, int ry, bool keys, const void* g, const int32_t* gk, long nrows, int D,
                                const void* q, const int32_t* qkeys, int Bq, int nb, float* ps, int64_t* pr,
                                unsigned* qb, hipStream_t s, bool attr) {
#define FR_SCANSEL(K)                                                                                         \
    if (kl == K && ry == 64 && (attr || g_sel_variant != 0)) {                                                \
        const hipError_t e =                                                                                  \
            keys ? scan_sel_variants<K, true, false>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, attr)  \
                 : scan_sel_variants<K, false, false>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, attr); \
        if (!attr || e != hipSuccess) return e;                                                               \
    }
    FR_SCANSEL(5)
    FR_SCANSEL(10)
    FR_SCANSEL(16)
#undef FR_SCANSEL
#define FR_SCAN(K, RY, WX, WY)                                                                              \
    if (kl == K && ry == RY)                                                                                  \
        return keys ? scan_inst<K, RY, WX, WY, true>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, attr)  \
                    : scan_inst<K, RY, WX, WY, false>(g, gk, nrows, D, q, qkeys, Bq, nb, ps, pr, qb, s, attr);
    FR_SCAN(5, 64, 4, 2)
    FR_SCAN(5, 128, 4, 2)
    FR_SCAN(5, 256, 2, 4)
    FR_SCAN(10, 64, 4, 2)
    FR_SCAN(10, 128, 4, 2)
    FR_SCAN(16, 64, 4, 2)
#undef FR_SCAN
    return hipErrorInvalidValue;
}

// `use_keys`: apply the per-row key test (needed when query keys are given or the gallery
// may hold dead rows).  The gallery buffers must be padded to a multiple of 256 rows.
static int g_bound_min_tiles = -1;       // < 0: the built-in rule below
void scan_set_bound_min_tiles(int t) { g_bound_min_tiles = t; }      // tuning hook of the micro-benchmark

size_t scan_bound_bytes(int Bq) { return (((size_t)Bq + 255) & ~(size_t)255) * 16 * sizeof(unsigned long long); }

// `qbound`: scan_bound_bytes(Bq) bytes of device scratch for the cross-block score bounds, zeroed once
// when allocated (the kernels without a selector wave clear it here, on the stream, before they run; the
// selector kernel's slots carry the launch's number and are never cleared), or null to scan without them.
hipError_t launch_scan_topk(const void* g, const int32_t* gk, int64_t nrows, int D, const void* q,
                            const int32_t* qkeys, int use_keys, int Bq, int k, float* part_s, int64_t* part_r,
                            unsigned* qbound, int* nblk_out, hipStream_t s) {
    if (nrows <= 0 || nrows > 0x7fffff00L || D < 64 || D % 64 || Bq <= 0 || k <= 0 || k > 16 || !g || !q ||
        !part_s || !part_r || !nblk_out || (use_keys && !gk))
        return hipErrorInvalidValue;
    // the kernel addresses the query block and a gallery tile with 32-bit lane offsets
    if ((uint64_t)Bq * (uint64_t)D * 2 >= (1ull << 32) || (uint64_t)D * 512 >= (1ull << 32)) return hipErrorInvalidValue;
    const int kl = scan_klist(k);
    const int ry = scan_query_tile(Bq, kl);
    const int qtiles = (Bq + ry - 1) / ry;
    const long ntiles = (nrows + 255) / 256;
    int nb = scan_walkers(qtiles);
    const long nt8 = (ntiles + 7) & ~7L;
    if (nb > nt8) nb = (int)nt8;
    *nblk_out = nb;
    // The bounds pay once a block walks enough tiles to amortise their reads (and the clearing launch):
    // measured break-even near 20 tiles for the HBM-bound 64-query tile and below 8 for the wider,
    // MFMA-bound tiles, whose tile-end selection the bounds shorten most.
    // (the 64-query tile's selector wave exchanges them beside the stream and its slots need no clearing,
    // but every exchange still disturbs all blocks' streams: same break-even)
    const bool sel = ry == 64 && g_sel_variant != 0;
    const int min_tiles = g_bound_min_tiles >= 0 ? g_bound_min_tiles : (sel ? 3 : ry >= 128 ? 8 : 24);
    if (ntiles / nb < min_tiles) qbound = nullptr;
    if (qbound && !sel) {
        hipError_t e = hipMemsetAsync(qbound, 0, scan_bound_bytes(Bq), s);
        if (e != hipSuccess) return e;
    }
    return scan_dispatch(kl, ry, use_keys != 0, g, gk, (long)nrows, D, q, qkeys, Bq, nb, part_s, part_r, qbound, s,
                         false);
}

// One launch over a list of work items (device memory): block b scans, for the 64-query tile that
// starts at work[b].q0 of the Bq queries, tiles gb, gb + nb, ... of its tile list.  The queries'
// keys are tested on every row.  Candidate lists land at part[(gb * Bq + query) * scan_klist(k)],
// so part_s / part_r hold (largest nb) * Bq lists; `qbound`: null, or scan_bound_bytes(Bq + 64) bytes
// (a key's first query tile starts at the key's first query, not at a multiple of 64, so the last
// tile's columns reach up to 63 queries past the end and their bounds are read like any other's).
hipError_t launch_scan_topk_work(const void* g, const int32_t* gk, int64_t nrows, int D, const void* q,
                                 const int32_t* qkeys, int Bq, int k, float* part_s, int64_t* part_r,
                                 unsigned* qbound, const int32_t* tile_lists, const ScanWork* work, int nwork,
                                 hipStream_t s) {
    if (nrows <= 0 || nrows > 0x7fffff00L || D < 64 || D % 64 || Bq <= 0 || k <= 0 || k > 16 || !g || !gk || !q ||
        !qkeys || !part_s || !part_r || !work || nwork <= 0)
        return hipErrorInvalidValue;
    if ((uint64_t)Bq * (uint64_t)D * 2 >= (1ull << 32) || (uint64_t)D * 512 >= (1ull << 32)) return hipErrorInvalidValue;
    const int kl = scan_klist(k);
    if (qbound && g_sel_variant == 0) {
        hipError_t e = hipMemsetAsync(qbound, 0, scan_bound_bytes(Bq + 64), s);
        if (e != hipSuccess) return e;
    }
#define FR_SCANW(K)                                                                                           \
    if (kl == K)                                                                                              \
        return g_sel_variant != 0                                                                             \
                   ? scan_sel_variants<K, true, true>(g, gk, (long)nrows, D, q, qkeys, Bq, 1, part_s, part_r, qbound, \
                                                      s, false, tile_lists, 0, work, nwork)                   \
                   : scan_inst<K, 64, 4, 2, true, true>(g, gk, (long)nrows, D, q, qkeys, Bq, 1, part_s, part_r,       \
                                                        qbound, s, false, tile_lists, 0, work, nwork);
    FR_SCANW(5)
    FR_SCANW(10)
    FR_SCANW(16)
#undef FR_SCANW
    return hipErrorInvalidValue;
}

hipError_t launch_rows_gather_f16(const void* in, const int32_t* index, int n, int D, void* out, hipStream_t s) {
    if (n <= 0 || D <= 0 || D % 8 || !in || !index || !out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_gather_f16_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const half_t*)in, index,
                       n, D, (half_t*)out);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_topk_final(const float* cs, const int64_t* cr, int nlists, int kin, int Bq, int kout,
                             int64_t row_base, float* out_s, int64_t* out_r, hipStream_t s, const int32_t* out_index,
                             const int32_t* nlists_q, int64_t stride_s, int64_t stride_r) {
    if (nlists <= 0 || nlists > 256 || kin <= 0 || kin > 16 || Bq <= 0 || kout <= 0 || kout > kin || !cs || !cr ||
        !out_s || !out_r)
        return hipErrorInvalidValue;
    const int64_t dense = (int64_t)Bq * kin;
    if (stride_s == 0) stride_s = dense;
    if (stride_r == 0) stride_r = dense;
    if (stride_s < dense || stride_r < dense) return hipErrorInvalidValue;
    const int kl = scan_klist(kin);
    if (kl == 5)
        hipLaunchKernelGGL((topk_final_kernel<5>), dim3(Bq), dim3(64), 0, s, cs, (const long long*)cr, nlists, kin, Bq,
                           kout, (long long)row_base, out_s, (long long*)out_r, out_index, nlists_q, (long)stride_s,
                           (long)stride_r);
    else if (kl == 10)
        hipLaunchKernelGGL((topk_final_kernel<10>), dim3(Bq), dim3(64), 0, s, cs, (const long long*)cr, nlists, kin,
                           Bq, kout, (long long)row_base, out_s, (long long*)out_r, out_index, nlists_q, (long)stride_s,
                           (long)stride_r);
    else
        hipLaunchKernelGGL((topk_final_kernel<16>), dim3(Bq), dim3(64), 0, s, cs, (const long long*)cr, nlists, kin,
                           Bq, kout, (long long)row_base, out_s, (long long*)out_r, out_index, nlists_q, (long)stride_s,
                           (long)stride_r);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t kernels_init() {
    {
        const hipError_t es = scan_static_resolve();
        if (es != hipSuccess) return es;
    }
    static const ConvLayerDesc layers[] = {
        {224, 64, 64, 1, 1},  {112, 64, 128, 0, 1}, {112, 128, 128, 1, 1}, {56, 128, 256, 0, 1},
        {56, 256, 256, 0, 1}, {56, 256, 256, 1, 1}, {28, 256, 512, 0, 1},  {28, 512, 512, 0, 1},
        {28, 512, 512, 1, 1}, {14, 512, 512, 0, 1}, {14, 512, 512, 1, 0},
    };
    for (const auto& d : layers) {
        hipError_t e = conv_dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true);
        if (e != hipSuccess) return e;
#if FR_WINOGRAD
        e = wino_dispatch(d, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true);
        if (e != hipSuccess && e != hipErrorInvalidValue) return e;      // invalid value: no Winograd form of this layer
#endif
    }
    hipError_t e = launch_patch_inst<224, 64, 64, true, 64, 4, 2, true>(nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0,
                                                                        nullptr, true);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fc_splitk_kernel<256, 128, 2, 2>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FcCore::SB);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fc_splitk_kernel<128, 128, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * GemmCore<128, 128, 2, 2, false>::SB);
    if (e != hipSuccess) return e;
    static const int combos[][2] = {{5, 64}, {5, 128}, {5, 256}, {10, 64}, {10, 128}, {16, 64}};
    for (const auto& c : combos)
        for (int keys = 0; keys < 2; ++keys) {
            e = scan_dispatch(c[0], c[1], keys != 0, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr,
                              nullptr, nullptr, true);
            if (e != hipSuccess) return e;
        }
    // the work-list instantiations (search by key)
    e = scan_sel_variants<5, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    if (e == hipSuccess)
        e = scan_sel_variants<10, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    if (e == hipSuccess)
        e = scan_sel_variants<16, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    if (e != hipSuccess) return e;
    e = scan_inst<5, 64, 4, 2, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    if (e == hipSuccess)
        e = scan_inst<10, 64, 4, 2, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    if (e == hipSuccess)
        e = scan_inst<16, 64, 4, 2, true, true>(nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, true);
    return e;
}

}  // namespace frhot
