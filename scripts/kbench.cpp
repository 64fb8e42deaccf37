// Kernel micro-benchmark for libfrhot's HIP kernels (developer tool, not part of the library).
// Times each conv layer of the VGG-Face stack and the gallery scan at several shapes with HIP
// events, on random data, and prints ms + TFLOP/s or GB/s per kernel.
//   kbench conv [B]            every conv layer at batch B (default 256)
//   kbench fuse [B]            the first two layers as two launches and as one, outputs compared (exit 1 if they differ)
//   kbench stack2 [B]          the conv stack as two halves on two streams, first two layers unfused / fused
//                              (KBENCH_BLOCKS, KBENCH_LAYERS, KBENCH_SYNC, KBENCH_V: see bench_stack2)
//   kbench scan                the scan at the benchmark shapes
//   kbench all
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../face-recognition_amd/csrc/frhot_kernels.h"

using namespace frhot;
#if FR_DBG & (8 | 32)
namespace frhot { extern __device__ unsigned long long g_stamp[64]; }
#endif
#if FR_DBG & 256
namespace frhot { hipError_t debug_stamps_read(unsigned long long* out64); hipError_t debug_stamps_clear(); }
#endif
#if FR_DBG & 64
namespace frhot { hipError_t scan_debug_stamps(unsigned long long* out); hipError_t scan_debug_tiles(unsigned long long* out); }
#include <algorithm>
#endif

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// fp16 bits of a uniform value in [-amp, amp)
static uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

static void* dev_random_f16(size_t n, float amp, uint64_t seed) {
    // a 1 Mi-element random block tiled over the buffer: random enough for clocks and caches
    const size_t blk = 1 << 20;
    std::vector<uint16_t> h(blk);
    uint64_t s = seed;
    for (size_t i = 0; i < blk; ++i) h[i] = f2h(((float)(sm64(s) >> 40) / 8388608.0f - 1.0f) * amp);
    void* d;
    CK(hipMalloc(&d, n * 2));
    for (size_t o = 0; o < n; o += blk) {
        const size_t c = n - o < blk ? n - o : blk;
        CK(hipMemcpy((char*)d + o * 2, h.data(), c * 2, hipMemcpyHostToDevice));
    }
    return d;
}

template <class F>
static float time_ms(F&& f, int warm, int iters, hipStream_t s) {
    for (int i = 0; i < warm; ++i) f();
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / iters;
}

#if FR_WINOGRAD
// Winograd form of a layer against the direct kernel: same input, outputs compared, both timed.
static void bench_wino(int B) {
    static const ConvLayerDesc L[] = {
        {56, 256, 256, 0, 1}, {56, 256, 256, 1, 1}, {28, 256, 512, 0, 1}, {28, 512, 512, 0, 1},
        {28, 512, 512, 1, 1}, {14, 512, 512, 0, 1}, {14, 512, 512, 1, 0},
    };
    hipStream_t s;
    CK(hipStreamCreate(&s));
    void* dump;
    CK(hipMalloc(&dump, 4096));
    for (const ConvLayerDesc& d : L) {
        const size_t xin = (size_t)B * (d.hw + 2) * (d.hw + 2) * d.cin;
        const int ho = d.pool ? d.hw / 2 : d.hw;
        const size_t yout = (size_t)B * (ho + 2 * d.opad) * (ho + 2 * d.opad) * d.cout;
        void* x = dev_random_f16(xin, 1.0f, 10);
        void* w = dev_random_f16((size_t)d.cout * 9 * d.cin, 0.05f, 40);
        std::vector<float> hb(d.cout);
        for (int i = 0; i < d.cout; ++i) hb[i] = 0.01f * (float)((i * 37) % 21 - 10);
        float* bias;
        CK(hipMalloc((void**)&bias, d.cout * 4));
        CK(hipMemcpy(bias, hb.data(), d.cout * 4, hipMemcpyHostToDevice));
        void *y0, *y1, *U;
        CK(hipMalloc(&y0, yout * 2));
        CK(hipMalloc(&y1, yout * 2));
        CK(hipMalloc(&U, (size_t)16 * d.cout * d.cin * 2));
        CK(hipMemset(y0, 0, yout * 2));
        CK(hipMemset(y1, 0, yout * 2));
        CK(launch_wino_weights(w, U, d.cin, d.cout, s));
        hipError_t e = launch_conv3x3_wino(d, x, U, bias, y1, B, s);
        if (e == hipErrorInvalidValue) {
            printf("hw %d cin %d cout %d pool %d: no Winograd form\n", d.hw, d.cin, d.cout, d.pool);
        } else {
            CK(e);
            CK(launch_conv3x3(d, x, w, bias, y0, dump, B, s));
            CK(hipStreamSynchronize(s));
            std::vector<uint16_t> h0(yout), h1(yout);
            CK(hipMemcpy(h0.data(), y0, yout * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), y1, yout * 2, hipMemcpyDeviceToHost));
            double mx = 0, md = 0, sd = 0;
            size_t bad = 0, nz = 0;
            for (size_t i = 0; i < yout; ++i) {
                _Float16 a, b;
                memcpy(&a, &h0[i], 2);
                memcpy(&b, &h1[i], 2);
                const double fa = (double)a, fb = (double)b, df = fabs(fa - fb);
                if (!(df == df)) ++bad;
                if (fabs(fa) > mx) mx = fabs(fa);
                if (df > md) md = df;
                sd += df;
                nz += fa != 0;
            }
            const float t0 = time_ms([&] { CK(launch_conv3x3(d, x, w, bias, y0, dump, B, s)); }, 3, 40, s);
#if FR_DBG & 256
            {
                CK(frhot::debug_stamps_clear());
            }
#endif
            const float t1 = time_ms([&] { CK(launch_conv3x3_wino(d, x, U, bias, y1, B, s)); }, 3, 40, s);
#if FR_DBG & 256
            {
                unsigned long long st[64];
                CK(frhot::debug_stamps_read(st));
                static const char* nm[] = {"prologue", "V0/patch issue+wait", "mma0", "wait patch+V1", "barrier B2", "dma issue+wait U1",
                                           "mma1+transform", "end wait", "barrier B1", "epilogue"};
                for (int wv = 0; wv < 2; ++wv) {
                    const double n = (double)st[wv * 16 + 10];
                    printf("   wave %d, cycles per item (%.0f items):", wv ? 5 : 0, n);
                    double tot = 0;
                    for (int i = 0; i < 10; ++i) { printf("  %s %.0f", nm[i], st[wv * 16 + i] / n); tot += st[wv * 16 + i] / n; }
                    printf("  total %.0f\n", tot);
                }
            }
#endif
            const double fl = 2.0 * B * d.hw * d.hw * 9.0 * d.cin * d.cout;
            printf("hw %3d cin %3d cout %3d pool %d: direct %7.3f ms (%6.1f TF)  winograd %7.3f ms (%6.1f TF-equivalent)  "
                   "max |diff| %.4g of max %.4g (%.2e), mean |diff| %.3g, nonzero %.2f, NaN %zu\n",
                   d.hw, d.cin, d.cout, d.pool, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, md, mx, md / (mx > 0 ? mx : 1),
                   sd / yout, (double)nz / yout, bad);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(y0)); CK(hipFree(y1)); CK(hipFree(U));
    }
    CK(hipFree(dump));
    CK(hipStreamDestroy(s));
}

#endif  // FR_WINOGRAD

static void bench_conv(int B) {
    static const ConvLayerDesc L[] = {
        {224, 64, 64, 1, 1},  {112, 64, 128, 0, 1}, {112, 128, 128, 1, 1}, {56, 128, 256, 0, 1},
        {56, 256, 256, 0, 1}, {56, 256, 256, 1, 1}, {28, 256, 512, 0, 1},  {28, 512, 512, 0, 1},
        {28, 512, 512, 1, 1}, {14, 512, 512, 0, 1}, {14, 512, 512, 1, 0},
    };
    static const char* names[] = {"conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
                                  "conv4_1", "conv4_2", "conv4_3", "conv5_1/2", "conv5_3"};
    hipStream_t s;
    CK(hipStreamCreate(&s));
    void* dump;
    CK(hipMalloc(&dump, 4096));
    double tot_ms = 0, tot_fl = 0;
    {
        // conv1_1
        void* x0 = dev_random_f16((size_t)B * 226 * 226 * 4, 8.0f, 1);
        void* w = dev_random_f16(64 * 64, 0.1f, 2);
        float* bias;
        CK(hipMalloc((void**)&bias, 64 * 4));
        CK(hipMemset(bias, 0, 64 * 4));
        void* y;
        CK(hipMalloc(&y, (size_t)B * 226 * 226 * 64 * 2));
        const float ms = time_ms([&] { CK(launch_conv1_1(x0, w, bias, y, B, s)); }, 3, 40, s);
        const double bytes = (double)B * 224 * 224 * 64 * 2;
        printf("%-10s %8.3f ms  store %7.1f GB/s\n", "conv1_1", ms, bytes / ms / 1e6);
        tot_ms += ms;
        tot_fl += 2.0 * B * 224 * 224 * 27 * 64;
        CK(hipFree(x0)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(y));
    }
    for (int i = 0; i < 11; ++i) {
        const ConvLayerDesc& d = L[i];
        const size_t xin = (size_t)B * (d.hw + 2) * (d.hw + 2) * d.cin;
        const int ho = d.pool ? d.hw / 2 : d.hw;
        const size_t yout = (size_t)B * (ho + 2 * d.opad) * (ho + 2 * d.opad) * d.cout;
        void* x = dev_random_f16(xin, 1.0f, 10 + i);
        void* w = dev_random_f16((size_t)d.cout * 9 * d.cin, 0.05f, 40 + i);
        float* bias;
        CK(hipMalloc((void**)&bias, d.cout * 4));
        CK(hipMemset(bias, 0, d.cout * 4));
        void* y;
        CK(hipMalloc(&y, yout * 2));
        CK(hipMemset(y, 0, yout * 2));
        const float ms = time_ms([&] { CK(launch_conv3x3(d, x, w, bias, y, dump, B, s)); }, 3, 40, s);
        const double fl = 2.0 * B * d.hw * d.hw * 9.0 * d.cin * d.cout;
        const int reps = (i == 9) ? 2 : 1;
        printf("%-10s %8.3f ms  %7.1f TFLOP/s   (hw %d cin %d cout %d pool %d)\n", names[i], ms, fl / ms / 1e9,
               d.hw, d.cin, d.cout, d.pool);
#if FR_DBG & 8
        if (i >= 1 && i < 3) {
            unsigned long long st[64];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(frhot::g_stamp), sizeof st));
            for (int wv = 0; wv < 8; ++wv)
                printf("   wave %d: per unit L %5.0f  barrier %5.0f  M %5.0f  barrier %5.0f cycles (%llu units)\n", wv,
                       (double)st[wv * 5] / st[wv * 5 + 4], (double)st[wv * 5 + 1] / st[wv * 5 + 4],
                       (double)st[wv * 5 + 2] / st[wv * 5 + 4], (double)st[wv * 5 + 3] / st[wv * 5 + 4], st[wv * 5 + 4]);
            for (int wv = 0; wv < 8; ++wv) {
                const double n = (double)st[wv * 5 + 4];
                printf("   wave %d: unit 1   L %5.0f  vmwait %5.0f  barrier %5.0f  M %5.0f  barrier %5.0f\n", wv,
                       (double)(st[40 + wv * 3] & 0xffffffffu) / n, (double)(st[40 + wv * 3] >> 32) / n,
                       (double)(st[40 + wv * 3 + 1] & 0xffffffffu) / n, (double)(st[40 + wv * 3 + 1] >> 32) / n,
                       (double)st[40 + wv * 3 + 2] / n);
            }
        }
#endif
#if FR_DBG & 32
        if (i >= 3) {
            unsigned long long st[64];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(frhot::g_stamp), sizeof st));
            for (int wv = 0; wv < 8; wv += 4) {
                const double n = (double)st[wv * 8 + 6];
                printf("   wave %d per step: head(R0,M3,epi,DMA) %5.0f  u0 %5.0f  u1 %5.0f  u2 %5.0f  vmwait %5.0f  barrier %5.0f  (%.0f steps)\n",
                       wv, st[wv * 8] / n, st[wv * 8 + 1] / n, st[wv * 8 + 2] / n, st[wv * 8 + 3] / n, st[wv * 8 + 4] / n,
                       st[wv * 8 + 5] / n, n);
            }
        }
#endif
        tot_ms += ms * reps;
        tot_fl += fl * reps;
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(y));
    }
    printf("conv stack %8.3f ms  %7.1f TFLOP/s at batch %d\n", tot_ms, tot_fl / tot_ms / 1e9, B);
    CK(hipFree(dump));
    CK(hipStreamDestroy(s));
}

// The first two layers as two launches against the fused one: same input, outputs compared bit for bit, both timed.
static void bench_fuse(int B) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    void* dump;
    CK(hipMalloc(&dump, 4096));
    void* x0 = dev_random_f16((size_t)B * 226 * 226 * 4, 8.0f, 1);
    void* w1 = dev_random_f16(64 * 64, 0.1f, 2);
    void* w2 = dev_random_f16((size_t)64 * 9 * 64, 0.05f, 40);
    if (getenv("KBENCH_IMAGE")) {
        // an image-like input: smooth values, zero fourth channel and halo, first-layer weights zero beyond k = 35
        std::vector<uint16_t> img((size_t)226 * 226 * 4, 0);
        uint64_t sd = 5;
        for (int yy = 1; yy <= 224; ++yy)
            for (int xx = 1; xx <= 224; ++xx)
                for (int c = 0; c < 3; ++c) {
                    const float v = 7.0f * (0.5f + 0.4f * sinf(0.05f * xx + c) * cosf(0.04f * yy)) + 0.3f * ((float)(sm64(sd) >> 40) / 16777216.0f);
                    img[((size_t)yy * 226 + xx) * 4 + c] = f2h(v - 6.0f);
                }
        for (int b = 0; b < B; ++b) CK(hipMemcpy((char*)x0 + (size_t)b * img.size() * 2, img.data(), img.size() * 2, hipMemcpyHostToDevice));
        std::vector<uint16_t> hw(64 * 64);
        CK(hipMemcpy(hw.data(), w1, hw.size() * 2, hipMemcpyDeviceToHost));
        for (int r = 0; r < 64; ++r)
            for (int k = 36; k < 64; ++k) hw[r * 64 + k] = 0;
        CK(hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> hb(128);
    for (int i = 0; i < 128; ++i) hb[i] = 0.05f * (float)((i * 37) % 21 - 10);
    float* bias;
    CK(hipMalloc((void**)&bias, 128 * 4));
    CK(hipMemcpy(bias, hb.data(), 128 * 4, hipMemcpyHostToDevice));
    const size_t mid = (size_t)B * 226 * 226 * 64, yout = (size_t)B * 114 * 114 * 64;
    void *ym, *y0, *y1;
    CK(hipMalloc(&ym, mid * 2));
    CK(hipMalloc(&y0, yout * 2));
    CK(hipMalloc(&y1, yout * 2));
    CK(hipMemset(ym, 0, mid * 2));
    CK(hipMemset(y0, 0, yout * 2));
    CK(hipMemset(y1, 0xff, yout * 2));
    const ConvLayerDesc d{224, 64, 64, 1, 1, 0};
    for (int rev = 0; rev < 2; ++rev) {
        ConvLayerDesc dr = d;
        dr.reverse = rev;
        CK(launch_conv1_1(x0, w1, bias, ym, B, s));
        CK(launch_conv3x3(dr, ym, w2, bias + 64, y0, dump, B, s));
        CK(launch_conv1_fused(x0, w1, bias, w2, bias + 64, y1, dump, B, rev, s));
        CK(hipStreamSynchronize(s));
        std::vector<uint16_t> h0(yout), h1(yout);
        CK(hipMemcpy(h0.data(), y0, yout * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), y1, yout * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, nz = 0, first = yout;
        for (size_t b = 0; b < (size_t)B; ++b)
            for (int yy = 1; yy <= 112; ++yy)
                for (int xx = 1; xx <= 112; ++xx)
                    for (int c = 0; c < 64; ++c) {
                        const size_t i = ((b * 114 + yy) * 114 + xx) * 64 + c;
                        if (h0[i] != h1[i]) { ++diff; if (first == yout) first = i; }
                        nz += h0[i] != 0;
                    }
        printf("fused vs two launches (reverse %d): %zu of %zu interior values differ, %.2f nonzero", rev, diff,
               (size_t)B * 112 * 112 * 64, (double)nz / ((double)B * 112 * 112 * 64));
        if (diff) {
            const size_t c = first % 64, xx = first / 64 % 114, yy = first / 64 / 114 % 114, b = first / 64 / 114 / 114;
            printf("; first at image %zu y %zu x %zu channel %zu", b, yy, xx, c);
        }
        printf("\n");
        if (diff) exit(1);
    }
    const float t1 = time_ms([&] { CK(launch_conv1_1(x0, w1, bias, ym, B, s)); }, 3, 40, s);
    const float t2 = time_ms([&] { CK(launch_conv3x3(d, ym, w2, bias + 64, y0, dump, B, s)); }, 3, 40, s);
    const float t12 = time_ms([&] { CK(launch_conv1_1(x0, w1, bias, ym, B, s)); CK(launch_conv3x3(d, ym, w2, bias + 64, y0, dump, B, s)); }, 3, 40, s);
    const float tf = time_ms([&] { CK(launch_conv1_fused(x0, w1, bias, w2, bias + 64, y1, dump, B, 0, s)); }, 3, 40, s);
    {
        // the library's schedule: the batch as two halves on two streams, each half its own launches
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t fork, join;
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        const int h = B / 2;
        const size_t x0h = (size_t)h * 226 * 226 * 4 * 2, ymh = (size_t)h * 226 * 226 * 64 * 2, yh = (size_t)h * 114 * 114 * 64 * 2;
        for (int fused = 0; fused < 2 && h > 0; ++fused)
            for (int rev = 0; rev < 2; ++rev) {
                ConvLayerDesc dr = d;
                dr.reverse = rev;
                const float t = time_ms([&] {
                    CK(hipEventRecord(fork, s));
                    CK(hipStreamWaitEvent(s2, fork, 0));
                    for (int part = 0; part < 2; ++part) {
                        hipStream_t hs = part ? s2 : s;
                        const char* xi = (const char*)x0 + part * x0h;
                        char* mi = (char*)ym + part * ymh;
                        char* yi = (char*)(fused ? y1 : y0) + part * yh;
                        if (fused) CK(launch_conv1_fused(xi, w1, bias, w2, bias + 64, yi, dump, h, rev, hs));
                        else { CK(launch_conv1_1(xi, w1, bias, mi, h, hs)); CK(launch_conv3x3(dr, mi, w2, bias + 64, yi, dump, h, hs)); }
                    }
                    CK(hipEventRecord(join, s2));
                    CK(hipStreamWaitEvent(s, join, 0));
                }, 3, 40, s);
                printf("two halves on two streams, %s, reverse %d: %.3f ms\n", fused ? "fused" : "two launches", rev, t);
            }
        CK(hipStreamDestroy(s2));
    }
    const float tfr = time_ms([&] { CK(launch_conv1_fused(x0, w1, bias, w2, bias + 64, y1, dump, B, 1, s)); }, 3, 40, s);
    printf("fused, reverse 1: %.3f ms\n", tfr);
    const double fl = 2.0 * B * 224 * 224 * 9.0 * 64 * 64;
    printf("batch %d: conv1_1 %.3f ms, conv1_2 %.3f ms (%.0f TF), back to back %.3f ms; fused %.3f ms (%.0f TF of the second layer)\n", B,
           t1, t2, fl / t2 / 1e9, t12, tf, fl / tf / 1e9);
    CK(hipFree(x0)); CK(hipFree(w1)); CK(hipFree(w2)); CK(hipFree(bias)); CK(hipFree(ym)); CK(hipFree(y0)); CK(hipFree(y1));
    CK(hipFree(dump));
    CK(hipStreamDestroy(s));
}

// The conv stack as the library schedules it: two halves of the batch on two streams, every layer a
// launch per half; the first two layers as two launches or as the fused one.  KBENCH_LAYERS=n stops after n layers.
static void bench_stack2(int B) {
    static const ConvLayerDesc L[] = {
        {224, 64, 64, 1, 1},  {112, 64, 128, 0, 1}, {112, 128, 128, 1, 1}, {56, 128, 256, 0, 1},
        {56, 256, 256, 0, 1}, {56, 256, 256, 1, 1}, {28, 256, 512, 0, 1},  {28, 512, 512, 0, 1},
        {28, 512, 512, 1, 1}, {14, 512, 512, 0, 1}, {14, 512, 512, 0, 1}, {14, 512, 512, 1, 0},
    };
    const int nl = getenv("KBENCH_LAYERS") ? atoi(getenv("KBENCH_LAYERS")) : 12;
    const int nparts = getenv("KBENCH_PARTS") ? atoi(getenv("KBENCH_PARTS")) : 2;      // 2..8 equal parts, one stream each
    if (nparts < 2 || nparts > 8 || B % nparts) { fprintf(stderr, "KBENCH_PARTS must divide the batch\n"); exit(1); }
    hipStream_t s, side[8];
    CK(hipStreamCreate(&s));
    hipEvent_t fork, join[8];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int p = 1; p < nparts; ++p) {
        CK(hipStreamCreateWithFlags(&side[p], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&join[p], hipEventDisableTiming));
    }
    void* dump;
    CK(hipMalloc(&dump, 4096));
    void* x0 = dev_random_f16((size_t)B * 226 * 226 * 4, 8.0f, 1);
    void* w1 = dev_random_f16(64 * 64, 0.1f, 2);
    float* bias;
    CK(hipMalloc((void**)&bias, 512 * 4));
    CK(hipMemset(bias, 0, 512 * 4));
    void* act[13];
    size_t per[13];
    per[0] = (size_t)226 * 226 * 64 * 2;
    void* w[12];
    for (int i = 0; i < 12; ++i) {
        const ConvLayerDesc& d = L[i];
        const int ho = d.pool ? d.hw / 2 : d.hw;
        per[i + 1] = (size_t)(ho + 2 * d.opad) * (ho + 2 * d.opad) * d.cout * 2;
        // uniform with the He standard deviation sqrt(2 / fan_in): activations keep their scale through the stack
        w[i] = dev_random_f16((size_t)d.cout * 9 * d.cin, sqrtf(3.0f * 2.0f / (9.0f * d.cin)), 40 + i);
    }
    for (int i = 0; i < 13; ++i) {
        CK(hipMalloc(&act[i], per[i] * B));
        CK(hipMemset(act[i], 0, per[i] * B));
    }
    const int h = B / nparts;
    // KBENCH_V, variants of the fused schedule: 2 = the second half's first launch waits for the first half's,
    // 4 = one launch for the whole batch before the fork
    const int variant = getenv("KBENCH_V") ? atoi(getenv("KBENCH_V")) : 0;
    hipEvent_t first_done;
    CK(hipEventCreateWithFlags(&first_done, hipEventDisableTiming));
    int tab[12] = {};       // KBENCH_BLOCKS: persistent blocks per launch of conv1_2 (or the fused launch) .. conv5_3, comma separated (0: one per CU)
    {
        const char* e = getenv("KBENCH_BLOCKS");
        for (int k = 0; k < 12 && e && *e; ++k) { tab[k] = atoi(e); e = strchr(e, ','); if (e) ++e; else break; }
    }
    for (int fused = 0; fused < 2; ++fused) {
        const float t = time_ms([&] {
            if (fused && variant == 4) CK(launch_conv1_fused(x0, w1, bias, w[0], bias, act[1], dump, B, 1, s));
            CK(hipEventRecord(fork, s));
            for (int p = 1; p < nparts; ++p) CK(hipStreamWaitEvent(side[p], fork, 0));
            for (int part = 0; part < nparts; ++part) {
                hipStream_t hs = part ? side[part] : s;
                const char* xi = (const char*)x0 + (size_t)part * h * 226 * 226 * 8;
                int first = 0;
                if (fused) {
                    if (variant == 2 && part == 1) CK(hipStreamWaitEvent(hs, first_done, 0));
                    if (variant != 4)
                        CK(launch_conv1_fused(xi, w1, bias, w[0], bias, (char*)act[1] + part * h * per[1], dump, h, 1, hs,
                                              tab[0]));
                    if (variant == 2 && part == 0) CK(hipEventRecord(first_done, hs));
                    first = 1;
                } else {
                    CK(launch_conv1_1(xi, w1, bias, (char*)act[0] + part * h * per[0], h, hs));
                }
                for (int i = first; i < nl; ++i) {
                    ConvLayerDesc d = L[i];
                    d.reverse = (i + 1) & 1;
                    d.blocks = tab[i];
                    CK(launch_conv3x3(d, (const char*)act[i] + part * h * per[i], w[i], bias, (char*)act[i + 1] + part * h * per[i + 1],
                                      dump, h, hs));
                }
            }
            for (int p = 1; p < nparts; ++p) {
                CK(hipEventRecord(join[p], side[p]));
                CK(hipStreamWaitEvent(s, join[p], 0));
            }
            if (getenv("KBENCH_SYNC")) CK(hipStreamSynchronize(s));       // every pass starts on an idle GPU
        }, 3, getenv("KBENCH_ITERS") ? atoi(getenv("KBENCH_ITERS")) : 30, s);       // (30 passes are a burst: clocks settle lower after about a second)
        printf("conv stack (%d layers after the first), %d parts of %d faces on as many streams, %s: %.3f ms\n", nl, nparts, h,
               fused ? "first two layers fused" : "first two layers as two launches", t);
    }
}

static void bench_scan_one(long N, int D, int Bq, int k, int use_keys, hipStream_t s) {
    const size_t padded = ((size_t)N + 255) & ~(size_t)255;
    void* g;
    int32_t* gk;
    CK(hipMalloc(&g, padded * D * 2));
    CK(hipMalloc((void**)&gk, padded * 4));
    CK(launch_gallery_fill(g, gk, D, 0xFACE0000ull + N, 0, N, use_keys ? 4 : 0, 0, s));
    void* q = dev_random_f16((size_t)Bq * D, 0.05f, 7);
    int32_t* qk = nullptr;
    if (use_keys) {
        CK(hipMalloc((void**)&qk, Bq * 4));
        CK(hipMemset(qk, 0, Bq * 4));   // every query asks for key 0
    }
    const int kl = scan_list_len(k);
    float* ps;
    int64_t* pr;
    CK(hipMalloc((void**)&ps, (size_t)scan_max_blocks() * Bq * kl * 4));
    CK(hipMalloc((void**)&pr, (size_t)scan_max_blocks() * Bq * kl * 8));
    unsigned* qb = nullptr;
    if (!getenv("KBENCH_NO_BOUND")) {
        CK(hipMalloc((void**)&qb, scan_bound_bytes(Bq)));
        CK(hipMemset(qb, 0, scan_bound_bytes(Bq)));
    }
    int nblk = 0;
    const float ms = time_ms(
        [&] { CK(launch_scan_topk(g, gk, N, D, q, qk, use_keys, Bq, k, ps, pr, qb, &nblk, s)); }, 3, 40, s);
    const double bytes = (double)N * D * 2 + (use_keys ? (double)N * 4 : 0.0) + (double)Bq * D * 2;
    const double fl = 2.0 * Bq * (double)N * D;
    float* os;
    int64_t* orr;
    CK(hipMalloc((void**)&os, (size_t)Bq * k * 4));
    CK(hipMalloc((void**)&orr, (size_t)Bq * k * 8));
    const float ms_final = time_ms([&] { CK(launch_topk_final(ps, pr, nblk, kl, Bq, k, 0, os, orr, s)); }, 3, 40, s);
    printf("scan N=%-9ld D=%-4d Bq=%-4d k=%-2d keys=%d  %8.3f ms  %7.1f GB/s (%.2f of 8 TB/s)  %7.1f TFLOP/s  nblk %d  final merge %.3f ms\n",
           N, D, Bq, k, use_keys, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, fl / ms / 1e9, nblk, ms_final);
#if FR_DBG & 64
    {
        CK(launch_scan_topk(g, gk, N, D, q, qk, use_keys, Bq, k, ps, pr, qb, &nblk, s));
        CK(hipStreamSynchronize(s));
        static unsigned long long st[256 * 16];
        CK(frhot::scan_debug_stamps(st));
        const int nbk = nblk < 256 ? nblk : 256;
        std::vector<double> col[10];
        for (int b = 0; b < nbk; ++b) {
            const unsigned long long* o = st + b * 16;
            const double v[10] = {(double)(o[1] - o[0]), (double)(o[2] - o[1]), (double)(o[3] - o[2]), (double)o[4],
                                  (double)o[5], (double)o[6], (double)o[7], (double)o[8], (double)o[9], 0.0};
            for (int i = 0; i < 9; ++i) col[i].push_back(v[i]);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
        printf("   [selector kernel: loop end -> selector exit in place of the merge tail; 'before the groups' = first tile's end, "
               "'group preludes' = overflow rounds, 'insertion loops' = selector drain cycles, rounds = selector rounds]\n");
        printf("   stamps (cycles, wave 0; median / max over %d blocks): prologue %.0f / %.0f  loop %.0f / %.0f  merge tail %.0f / %.0f\n"
               "      tile ends: %.0f in all, %.0f / %.0f cycles  = before the groups %.0f + group preludes %.0f + insertion loops %.0f "
               "(%.0f / %.0f rounds)\n",
               nbk, med(col[0]), mx(col[0]), med(col[1]), mx(col[1]), med(col[2]), mx(col[2]), med(col[4]), med(col[3]), mx(col[3]),
               med(col[6]), med(col[7]), med(col[8]), med(col[5]), mx(col[5]));
        static unsigned long long tl[256 * 20];
        CK(frhot::scan_debug_tiles(tl));
        {
            std::vector<double> ph[5];
            for (int b = 0; b < nbk; ++b)
                for (int i = 0; i < 5; ++i) ph[i].push_back((double)st[b * 16 + 10 + i]);
            printf("      first tile's end: test+strike+ring wait %.0f, best-score table -> barrier %.0f, count -> barrier %.0f, "
                   "queue -> barrier -> flag %.0f, ranking -> barrier %.0f\n",
                   med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]));
        }
        printf("      tile durations (median over blocks, wave 0):");
        for (int t = 0; t < 16; ++t) {
            std::vector<double> v;
            for (int b = 0; b < nbk; ++b) v.push_back((double)tl[b * 20 + t] - (t ? (double)tl[b * 20 + t - 1] : 0.0));
            printf(" %.0f", med(v));
        }
        printf("\n");
    }
#endif
    CK(hipFree(os)); CK(hipFree(orr));
    if (qb) CK(hipFree(qb));
    CK(hipFree(g)); CK(hipFree(gk)); CK(hipFree(q)); CK(hipFree(ps)); CK(hipFree(pr));
    if (qk) CK(hipFree(qk));
}

static void bench_scan() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    if (getenv("KBENCH_BOUND_MIN_TILES")) {
        scan_set_bound_min_tiles(atoi(getenv("KBENCH_BOUND_MIN_TILES")));
        for (long n : {1000000L, 2000000L, 4000000L}) {
            bench_scan_one(n, 512, 64, 5, 0, s);
            bench_scan_one(n, 512, 64, 10, 0, s);
            bench_scan_one(n, 512, 256, 5, 0, s);
        }
        CK(hipStreamDestroy(s));
        return;
    }
    if (getenv("KBENCH_D4096")) {         // the reference's descriptor length at serving-sized query counts
        for (int bq : {1, 8, 64, 128, 256}) bench_scan_one(1000000, 4096, bq, 5, 0, s);
        bench_scan_one(1000000, 4096, 64, 5, 1, s);
        CK(hipStreamDestroy(s));
        return;
    }
    if (getenv("KBENCH_SEL")) {           // 64-query tile: without (0) and with the selector wave (1, 2), same process
        const int rounds = atoi(getenv("KBENCH_SEL")) > 0 ? atoi(getenv("KBENCH_SEL")) : 2;
        for (int r = 0; r < rounds; ++r)
            for (int v : {0, 1}) {
                if (getenv("KBENCH_SELV") && !strchr(getenv("KBENCH_SELV"), '0' + v)) continue;
                scan_set_sel_variant(v);
                printf("-- selector variant %d (round %d)\n", v, r);
                bench_scan_one(1000000, 512, 64, 10, 0, s);
                bench_scan_one(1000000, 512, 64, 5, 0, s);
                bench_scan_one(2000000, 512, 64, 10, 0, s);
                bench_scan_one(10000000, 512, 64, 10, 0, s);
                bench_scan_one(1000000, 4096, 64, 5, 0, s);
                bench_scan_one(1000000, 512, 64, 10, 1, s);
            }
        scan_set_sel_variant(1);
        CK(hipStreamDestroy(s));
        return;
    }
    if (getenv("KBENCH_SHORT")) {         // the HBM-bound shape and its neighbours only
        for (long n : {1000000L, 2000000L}) {
            bench_scan_one(n, 512, 64, 10, 0, s);
            bench_scan_one(n, 512, 64, 5, 0, s);
        }
        bench_scan_one(10000000, 512, 64, 10, 0, s);
        bench_scan_one(10000000, 512, 256, 5, 0, s);
        CK(hipStreamDestroy(s));
        return;
    }
    bench_scan_one(1000000, 512, 64, 10, 0, s);
    bench_scan_one(1000000, 512, 64, 5, 0, s);
    bench_scan_one(10000000, 512, 64, 5, 0, s);
    bench_scan_one(10000000, 512, 64, 10, 0, s);
    bench_scan_one(10000000, 512, 64, 16, 0, s);
    bench_scan_one(10000000, 512, 128, 10, 0, s);
    bench_scan_one(10000000, 512, 128, 5, 0, s);
    bench_scan_one(10000000, 512, 256, 5, 0, s);
    bench_scan_one(10000000, 512, 256, 5, 1, s);
    bench_scan_one(1250000, 512, 2048, 5, 0, s);
    bench_scan_one(1000000, 4096, 256, 5, 0, s);
    CK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    CK(kernels_init());
    if (!strcmp(what, "conv") || !strcmp(what, "all")) bench_conv(argc > 2 ? atoi(argv[2]) : 256);
    if (!strcmp(what, "fuse")) bench_fuse(argc > 2 ? atoi(argv[2]) : 256);
    if (!strcmp(what, "stack2")) bench_stack2(argc > 2 ? atoi(argv[2]) : 256);
#if FR_WINOGRAD
    if (!strcmp(what, "wino")) bench_wino(argc > 2 ? atoi(argv[2]) : 256);
#endif
    if (!strcmp(what, "scan") || !strcmp(what, "all")) bench_scan();
    return 0;
}
