// Calibration probe (developer tool): how fast can 8 waves per CU run blocks of
// {R x ds_read_b128, M x v_mfma_f32_16x16x32_f16} under different instruction orders?
// Prints achieved TFLOP/s per variant so kernel schedules can be judged against a ceiling
// measured on the same chip and clock behaviour.
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

// NA x NB accumulator tile per wave: NA + NB fragment reads and NA*NB MFMAs per block.
// MODE 0: reads, wait, MFMAs (serial per wave)
// MODE 1: two fragment sets, reads of the next block issued before the MFMAs of this one
// MODE 2: as 1, waves 4..7 start half a block late
// MODE 3: as 1 with s_setprio 1 around the MFMAs
template <int NA, int NB, int MODE>
__global__ __launch_bounds__(512) void probe(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill 64 KiB of LDS with data
    for (int i = tid; i < 4096; i += 512) ((half8*)smem)[i] = ((const half8*)src)[i + blockIdx.x % 4 * 4096];
    __syncthreads();
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
    // conflict-free XOR-swizzled 128-B rows, as the real kernels read them
    const char* base = smem + (wave * 16 + l15) * 128;
    int off[2];
    off[0] = ((g0 ^ l7) << 4);
    off[1] = (((4 + g0) ^ l7) << 4);
    float4v acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
    half8 fa0[NA], fb0[NB], fa1[NA], fb1[NB];
    auto load = [&](int it, half8 (&fa)[NA], half8 (&fb)[NB]) {
        const char* p = base + ((it & 3) * 8192) + off[it & 1];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[a] = *(const half8*)(p + a * 2048);
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b] = *(const half8*)(p + 32768 + b * 2048);
    };
    auto mma = [&](const half8 (&fa)[NA], const half8 (&fb)[NB]) {
        if (MODE == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            load(it, fa0, fb0);
            mma(fa0, fb0);
        }
    } else if (MODE == 4 || MODE == 5 || MODE == 6) {
        // barrier-paced phases: reads | barrier | MFMAs | barrier; MODE 4: waves 4..7 run one
        // barrier behind (ping-pong); MODE 5: all waves in step; MODE 6: as 4 with s_setprio
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
        if (MODE != 5 && late) bar();
        for (int it = 0; it < iters; ++it) {
            load(it, fa0, fb0);
            bar();
            if (MODE == 6) __builtin_amdgcn_s_setprio(1);
            mma(fa0, fb0);
            if (MODE == 6) __builtin_amdgcn_s_setprio(0);
            bar();
        }
        if (MODE != 5 && !late) bar();
    } else {
        if (MODE == 2 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_sleep(NA * NB / 8);
        load(0, fa0, fb0);
        for (int it = 0; it < iters; it += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load(it + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            load(it + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NA, int NB, int MODE>
static void run(const _Float16* src, float* out, int waves) {
    const int iters = 4096;
    auto k = probe<NA, NB, MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, src, out, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double fl = 256.0 * waves * iters * NA * NB * 16.0 * 16 * 32 * 2;
    printf("NA=%d NB=%d mode=%d waves=%d: %7.3f ms  %7.1f TFLOP/s  (reads/MFMA %.2f)\n", NA, NB, MODE, waves, ms,
           fl / ms / 1e9, (double)(NA + NB) / (NA * NB));
}


// Ping-pong phases with NDMA LDS-DMA pieces (1 KiB each, L2-resident source) per wave and
// iteration, issued either in the load phase (WHERE 0), in the MFMA phase before the MFMAs
// (WHERE 1), or spread between the MFMAs (WHERE 2).
template <int NA, int NB, int NDMA, int WHERE, int SRC, int LOOK = 2>
__global__ __launch_bounds__(512) void probe_dma(const _Float16* __restrict__ src, float* __restrict__ out, int iters,
                                                  const char* __restrict__ big, size_t big_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 512) ((half8*)smem)[i] = ((const half8*)src)[i + blockIdx.x % 4 * 4096];
    __syncthreads();
    const int l15 = lane & 15, g0 = lane >> 4, l7 = lane & 7;
    const char* base = smem + (wave * 16 + l15) * 128;
    int off[2];
    off[0] = ((g0 ^ l7) << 4);
    off[1] = (((4 + g0) ^ l7) << 4);
    float4v acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
    half8 fa[NA], fb[NB];
    const char* gsrc = (const char*)src + (size_t)(blockIdx.x % 64) * 1024 + lane * 16;
    char* dst = smem + 65536 + wave * 4096;     // DMA target region, not read by the MFMA operands
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // SRC 1: every block streams its own 8*NDMA KiB per iteration out of a buffer far larger than the caches
    const size_t per_iter = (size_t)NDMA * 8 * 1024;
    const char* bsrc = big + ((size_t)blockIdx.x * 4096 * per_iter) % big_bytes + wave * NDMA * 1024 + lane * 16;
    auto dma = [&](int it, int i) {
        const char* g;
        if (SRC == 0) g = gsrc + ((it + i) & 15) * 65536;
        else if (SRC == 1) g = bsrc + (((size_t)it * per_iter) % (big_bytes / 2)) + i * 1024;
        else if (SRC == 2)      // 2 MiB window per XCD, walked by its 32 blocks at staggered offsets
            g = big + (size_t)(blockIdx.x & 7) * (2u << 20) +
                (((size_t)it * per_iter + (size_t)(blockIdx.x >> 3) * 65536 + wave * NDMA * 1024 + i * 1024) % (2u << 20)) + lane * 16;
        else                    // 128 MiB window shared by all blocks (Infinity Cache resident, beyond L2)
            g = big + (((size_t)it * per_iter + (size_t)blockIdx.x * (512u << 10) + wave * NDMA * 1024 + i * 1024) % (128u << 20)) + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + (i & 3) * 1024), 16, 0, 0);
    };
    const bool late = wave >= 4;
    if (late) bar();
    for (int it = 0; it < iters; ++it) {
        if (WHERE == 0) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) dma(it, i);
        }
        const char* p = base + ((it & 3) * 8192) + off[it & 1];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[a] = *(const half8*)(p + a * 2048);
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b] = *(const half8*)(p + 32768 + b * 2048);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOOK * NDMA) : "memory");
        bar();
        if (WHERE == 1) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) dma(it, i);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            if (WHERE == 2 && a < NDMA) { __builtin_amdgcn_sched_barrier(0); dma(it, a); __builtin_amdgcn_sched_barrier(0); }
        }
        bar();
    }
    if (!late) bar();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NA, int NB, int NDMA, int WHERE, int SRC, int LOOK = 2>
static void run_dma(const _Float16* src, float* out, const char* big, size_t big_bytes) {
    const int iters = 4096;
    auto k = probe_dma<NA, NB, NDMA, WHERE, SRC, LOOK>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 32768));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536 + 32768, 0, src, out, iters, big, big_bytes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double fl = 256.0 * 8 * iters * NA * NB * 16.0 * 16 * 32 * 2;
    printf("PP NA=%d NB=%d dma/wave/iter=%d where=%d src=%d look=%d: %7.3f ms  %7.1f TFLOP/s  DMA %.0f GB/s\n", NA, NB, NDMA,
           WHERE, SRC, LOOK, ms, fl / ms / 1e9, 256.0 * 8 * iters * NDMA * 1024 / ms / 1e6);
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
    CK(hipMalloc((void**)&out, 256 * 512 * 4));
    size_t big_bytes = (size_t)8 << 30;
    char* big;
    CK(hipMalloc((void**)&big, big_bytes));
    CK(hipMemset(big, 1, big_bytes));
    run_dma<4, 8, 8, 0, 0, 2>(src, out, big, big_bytes);
    run_dma<4, 8, 8, 0, 2, 2>(src, out, big, big_bytes); run_dma<4, 8, 8, 0, 3, 2>(src, out, big, big_bytes);
    run_dma<4, 8, 8, 0, 1, 2>(src, out, big, big_bytes);
    run_dma<4, 8, 4, 0, 2, 2>(src, out, big, big_bytes); run_dma<4, 8, 4, 0, 3, 2>(src, out, big, big_bytes);
    run<2, 4, 0>(src, out, 8); run<2, 4, 1>(src, out, 8); run<2, 4, 2>(src, out, 8); run<2, 4, 3>(src, out, 8);
    run<4, 4, 0>(src, out, 8); run<4, 4, 1>(src, out, 8); run<4, 4, 2>(src, out, 8); run<4, 4, 3>(src, out, 8);
    run<4, 8, 0>(src, out, 8); run<4, 8, 1>(src, out, 8); run<4, 8, 2>(src, out, 8); run<4, 8, 3>(src, out, 8);
    run<4, 4, 0>(src, out, 4); run<4, 4, 1>(src, out, 4); run<4, 8, 1>(src, out, 4);
    run<4, 4, 4>(src, out, 8); run<4, 4, 5>(src, out, 8); run<4, 4, 6>(src, out, 8);
    run<4, 8, 4>(src, out, 8); run<4, 8, 5>(src, out, 8); run<2, 4, 4>(src, out, 8);
    return 0;
}
