// Calibration probe (developer tool): ONE wave per SIMD (256 threads per CU) with a 128 x 128
// accumulator tile per wave (8 x 8 blocks of 16 x 16, 256 accumulator registers pinned in AGPRs
// through inline asm), fragments double-buffered per 32-deep half step, operands streamed into
// a 3 + 2 stage LDS ring by LDS-DMA as the convolution kernels do (16 one-KiB pieces per wave
// and 64-deep step).  Prints TFLOP/s for several DMA sources so that the 8-wave kernels
// (two waves per SIMD, 128 x 64 wave tiles, 0.375 fragment reads per MFMA) can be judged
// against what this layout (0.25 reads per MFMA) reaches on the same chip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void mfma_a(float4v& c, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// SRC 0: 1 MiB source shared by all blocks (L2 hits); 1: every block streams its own data out of
// a buffer far larger than the caches; 2: 2 MiB window per XCD; 3: 128 MiB window (Infinity Cache)
// NDMA = pieces per wave per step (16 = what a 256 x 256 x 64 step needs)
// RPM = fragment reads issued per MFMA slot while interleaving (1: one read after each MFMA)
template <int NDMA, int SRC, int BAR, int READS = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe88(
    const _Float16* __restrict__ src, float* __restrict__ out, int steps, const char* __restrict__ big, size_t big_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave >> 1, wy = wave & 1;
    for (int i = tid; i < 10240; i += 256) ((half8*)smem)[i] = ((const half8*)src)[i & 4095];
    __syncthreads();
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
    // ring: X stages at 0, 32 K, 64 K; Y stages at 96 K, 128 K; a stage = 256 rows x 128 B
    const char* xrow = smem + (wx * 128 + l15) * 128;
    const char* yrow = smem + 98304 + (wy * 128 + l15) * 128;
    int off[2];
    off[0] = ((g0 ^ l7) << 4);
    off[1] = (((4 + g0) ^ l7) << 4);

    float4v acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};

    const size_t per_step = (size_t)NDMA * 4 * 1024;
    const char* gsrc = (const char*)src + (size_t)(blockIdx.x % 64) * 1024 + lane * 16;
    const char* bsrc = big + ((size_t)blockIdx.x * 2048 * per_step) % big_bytes + wave * NDMA * 1024 + lane * 16;
    auto dma = [&](int st, int i) {
        const char* g;
        if (SRC == 0) g = gsrc + ((st + i) & 15) * 65536;
        else if (SRC == 1) g = bsrc + (((size_t)st * per_step) % (big_bytes / 2)) + i * 1024;
        else if (SRC == 2)
            g = big + (size_t)(blockIdx.x & 7) * (2u << 20) +
                (((size_t)st * per_step + (size_t)(blockIdx.x >> 3) * 65536 + wave * NDMA * 1024 + i * 1024) % (2u << 20)) + lane * 16;
        else if (SRC == 3)
            g = big + (((size_t)st * per_step + (size_t)blockIdx.x * (512u << 10) + wave * NDMA * 1024 + i * 1024) % (128u << 20)) + lane * 16;
        else {
            // as 2, but a piece's eight 128-B rows lie SRC bytes apart (1024: the pixel rows of a
            // 512-channel activation; 9216: the rows of a 512-input-channel 3x3 weight matrix)
            const size_t pb = ((size_t)st * per_step + (size_t)(blockIdx.x >> 3) * 65536 + wave * NDMA * 1024 + i * 1024);
            g = big + (size_t)(blockIdx.x & 7) * (2u << 20) + ((pb + (size_t)(lane >> 3) * SRC) % (2u << 20)) + (lane & 7) * 16;
        }
        // Y pieces first (the other Y stage, needed half a step from now), then the X pieces of the
        // stage two steps ahead: the counted wait leaves exactly those in flight
        char* dst = i < NDMA / 2 ? smem + 98304 + ((st + 1) & 1) * 32768 + (wave * (NDMA / 2) + i) * 1024
                                 : smem + ((st + 2) % 3) * 32768 + (wave * (NDMA / 2) + i - NDMA / 2) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    half8 fa[2][8], fb[2][8];
    // fragments of (step st, half kk) -> set kk
    auto read_a0 = [&](int xs, int kk, int i) { fa[kk][i] = *(const half8*)(xrow + xs * 32768 + i * 2048 + off[kk]); };
    auto read_b0 = [&](int ys, int kk, int i) { fb[kk][i] = *(const half8*)(yrow + ys * 32768 + i * 2048 + off[kk]); };
#pragma unroll
    for (int i = 0; i < 8; ++i) { read_a0(0, 0, i); read_b0(0, 0, i); read_a0(0, 1, i); read_b0(0, 1, i); }
    auto read_a = [&](int xs, int kk, int i) { if (READS) read_a0(xs, kk, i); };
    auto read_b = [&](int ys, int kk, int i) { if (READS) read_b0(ys, kk, i); };
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");

    int xs = 0, ys = 0;
    for (int st = 0; st < steps; ++st) {
        FENCE();
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma(st, i);
        FENCE();
        // half 0: MFMAs on set 0, reads of half 1 (same stages) between them
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                mfma_a(acc[a][b], fa[0][a], fb[0][b]);
                if (a < 2) {            // 16 reads behind the first 16 MFMAs
                    FENCE();
                    const int r = a * 8 + b;
                    if (r < 8) read_b(ys, 1, r); else read_a(xs, 1, r - 8);
                    FENCE();
                }
            }
        }
        FENCE();
        const int nxs = xs == 2 ? 0 : xs + 1, nys = ys ^ 1;
        // half 1: MFMAs on set 1; the next step's stages must have landed before its reads
        if (BAR == 0) {
            // wait + barrier first, then MFMAs with the next step's reads between them
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA / 2) : "memory");
            __builtin_amdgcn_s_barrier();
            FENCE();
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    mfma_a(acc[a][b], fa[1][a], fb[1][b]);
                    if (a < 2) {
                        FENCE();
                        const int r = a * 8 + b;
                        if (r < 8) read_b(nys, 0, r); else read_a(nxs, 0, r - 8);
                        FENCE();
                    }
                }
        } else {
            // first half of the MFMAs, then wait + barrier, then the rest with the reads between them
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                if (a == BAR) {
                    FENCE();
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA / 2) : "memory");
                    __builtin_amdgcn_s_barrier();
                    FENCE();
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    mfma_a(acc[a][b], fa[1][a], fb[1][b]);
                    if (a >= BAR && a < BAR + 2) {
                        FENCE();
                        const int r = (a - BAR) * 8 + b;
                        // set 0's A fragments 0..a and all of its B fragments are dead by now
                        if (r < 8) read_b(nys, 0, r); else read_a(nxs, 0, r - 8);
                        FENCE();
                    }
                }
            }
        }
        FENCE();
        xs = nxs; ys = nys;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (blockIdx.x == 8 && tid == 0) { out[256 * 256] = (float)(c1 - c0); out[256 * 256 + 1] = (float)(r1 - r0); }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 256 + tid] = s;
}

// fp16 values uniform in [-1, 1): operand bits that toggle as real activations and weights do
// (a constant fill lets the chip hold a clock it cannot hold on real data)
__global__ void fill_random(_Float16* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        p[i] = (_Float16)((float)(int)(h & 0xffff) / 32768.0f - 1.0f);
    }
}

template <int NDMA, int SRC, int BAR, int READS = 1>
static void run(const _Float16* src, float* out, const char* big, size_t big_bytes) {
    const int steps = 2048;
    auto k = probe88<NDMA, SRC, BAR, READS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 163840, 0, src, out, steps, big, big_bytes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double fl = 256.0 * 4 * steps * 128.0 * 16 * 16 * 32 * 2;
    float st[2];
    CK(hipMemcpy(st, out + 256 * 256, 8, hipMemcpyDeviceToHost));
    printf("1 wave/SIMD 8x8: dma/wave/step=%d src=%d bar=%d reads=%d: %7.3f ms  %7.1f TFLOP/s  DMA %.0f GB/s  (%.0f cycles/step at %.0f MHz)\n",
           NDMA, SRC, BAR, READS, ms, fl / ms / 1e9, 256.0 * 4 * steps * NDMA * 1024 / ms / 1e6, st[0] / steps, st[0] / st[1] * 100.0);
}

typedef float float16v __attribute__((ext_vector_type(16)));

// MFMA issue only (no LDS, no DMA): SHAPE 0 = 64 x v_mfma_f32_16x16x32_f16 per 32-deep half,
// SHAPE 1 = 32 x v_mfma_f32_32x32x16_f16 (same flops, half the operand-register reads per flop)
template <int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe_shape(
    const _Float16* __restrict__ src, float* __restrict__ out, int steps) {
    const int tid = threadIdx.x;
    half8 fa[16], fb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        fa[i] = ((const half8*)src)[(tid * 16 + i) & 4095];
        fb[i] = ((const half8*)src)[(tid * 16 + i + 2048) & 4095];
    }
    unsigned long long c0, r0, c1, r1;
    float s = 0.f;
    if (SHAPE == 0) {
        float4v acc[8][8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(fa[kk * 8 + a]), "v"(fb[kk * 8 + b]));
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) s += acc[a][b][0] + acc[a][b][3];
    } else {
        float16v acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(fa[kq * 4 + a]), "v"(fb[kq * 4 + b]));
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][15];
    }
    if (blockIdx.x == 8 && tid == 0) { out[256 * 256] = (float)(c1 - c0); out[256 * 256 + 1] = (float)(r1 - r0); }
    out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE>
static void run_shape(const _Float16* src, float* out) {
    const int steps = 2048;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe_shape<SHAPE>, dim3(256), dim3(256), 0, 0, src, out, steps);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    float st[2];
    CK(hipMemcpy(st, out + 256 * 256, 8, hipMemcpyDeviceToHost));
    const double fl = 256.0 * 4 * steps * 128.0 * 16 * 16 * 32 * 2;
    printf("MFMA only, %s: %7.3f ms  %7.1f TFLOP/s  (%.0f cycles/step at %.0f MHz)\n",
           SHAPE ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9, st[0] / steps, st[0] / st[1] * 100.0);
}

int main() {
    std::vector<_Float16> h(1 << 16);
    unsigned s = 12345;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = (_Float16)(((int)(s >> 9) % 2001 - 1000) / 1000.0f);
    }
    _Float16* src;
    float* out;
    CK(hipMalloc((void**)&src, (1 << 21)));
    CK(hipMemset(src, 0, 1 << 21));
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&out, 256 * 256 * 4 + 64));
    size_t big_bytes = (size_t)8 << 30;
    char* big;
    CK(hipMalloc((void**)&big, big_bytes));
    if (getenv("PROBE_CONST")) CK(hipMemset(big, 1, big_bytes));
    else { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (_Float16*)big, big_bytes / 2); CK(hipDeviceSynchronize()); }
    run_shape<0>(src, out);
    run_shape<1>(src, out);
    run_shape<0>(src, out);
    run_shape<1>(src, out);
    run<16, 2, 4>(src, out, big, big_bytes);
    run<16, 2, 4, 0>(src, out, big, big_bytes);
    run<8, 2, 4>(src, out, big, big_bytes);
    run<0, 2, 4>(src, out, big, big_bytes);
    run<0, 2, 4, 0>(src, out, big, big_bytes);
    run<16, 1024, 4>(src, out, big, big_bytes);
    run<16, 3, 4>(src, out, big, big_bytes);
    run<16, 1, 4>(src, out, big, big_bytes);
    return 0;
}
