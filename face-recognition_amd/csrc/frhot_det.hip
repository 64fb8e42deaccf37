// HIP kernels of the face detector for MI355X (gfx950, wave64, MFMA): the network of the reference's
// DeepFace.extract_faces(detector_backend="yolov8") calls (app/src/utils/legacy.py:60-65,80-86).
// Tensors are fp16 NHWC WITHOUT a halo, addressed as channel slices; weights fp16 [cout][k][k][cin].
#include <hip/hip_runtime.h>
#include <math.h>
#include "frhot_kernels.h"

namespace frhot {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define FR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// ------------------------------------------------------------ face detector ---
// The detector behind the reference's DeepFace.extract_faces(detector_backend="yolov8")
// (app/src/utils/legacy.py:60-65,80-86): a YOLOv8n-pose network (one class, five keypoints) on a
// 640 x 640 letterboxed frame.  Sixty-odd small convolutions (16..256 channels), so one generic
// kernel serves them all: NHWC fp16 tensors addressed as channel slices (a concatenation is two
// producers writing into one tensor), folded batch norm as a bias, SiLU and the bottleneck's
// residual in the epilogue.  A wave computes 32 pixels x up to 64 channels with
// v_mfma_f32_16x16x16_f16, both operands straight from global memory (the weights of a layer
// are a few KiB to 1 MiB and stay in L2; no LDS).
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// KD: input channels per MFMA (32: v_mfma_f32_16x16x32_f16 with 16-byte operand loads, for layers whose
// channel count allows it; 16: the x16 form with 8-byte loads).  PG: groups of 16 pixels per wave
// (4 for the large maps: every weight fragment then feeds four MFMAs; 2 keeps more waves busy on the
// 20 x 20 and 40 x 40 maps of a small batch).  NS: groups of 16 output channels per wave.
// The K loop runs over (tap, channel chunk) steps as ONE software-pipelined sequence: the operands
// of step s+1 are loaded before the MFMAs of step s issue, and nothing in it branches (a tap that
// falls outside the image reads a page of zeros instead of being skipped).
template <int KS, int ST, int KD, int PG, int NS>
__global__ __launch_bounds__(256) void det_conv_kernel(DetConv d, int M) {
    typedef _Float16 frag_t __attribute__((ext_vector_type(KD / 4)));
    constexpr int KL = KD / 4;          // halfs per lane and MFMA
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * (16 * NS);
    const int m0 = (blockIdx.x * 4 + wave) * (16 * PG);
    if (m0 >= M) return;
    const int HWo = d.Ho * d.Wo;
    int pb[PG], py[PG], px[PG];
    bool pv[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int m = m0 + pg * 16 + l15;
        pv[pg] = m < M;
        const int mm = pv[pg] ? m : M - 1;
        pb[pg] = mm / HWo;
        const int rem = mm - pb[pg] * HWo;
        py[pg] = rem / d.Wo;
        px[pg] = rem - py[pg] * d.Wo;
    }
    float4v acc[NS][PG];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) acc[ns][pg] = float4v{0.f, 0.f, 0.f, 0.f};
    const long wrow = (long)KS * KS * d.cin;            // halfs per output channel
    const half_t* const wbase = d.w + (long)(n0 + l15) * wrow + KL * g;
    const half_t* xp[PG];
    auto tap_setup = [&](int tap) __attribute__((always_inline)) {
        const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            const int iy = py[pg] * ST + ky - KS / 2, ix = px[pg] * ST + kx - KS / 2;
            const bool ok = pv[pg] && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            const half_t* real = d.x + (((long)pb[pg] * d.H + iy) * d.W + ix) * d.xc + d.xc0;
            xp[pg] = (ok ? real : d.zeros) + KL * g;
        }
    };
    frag_t fa[2][NS], fb[2][PG];
    int tap = 0, cg = 0;
    auto load = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) fb[buf][pg] = *(const frag_t*)(xp[pg] + cg);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) fa[buf][ns] = *(const frag_t*)(wbase + ns * 16 * wrow + tap * d.cin + cg);
        cg += KD;
        if (cg == d.cin) {
            cg = 0;
            ++tap;
            if (tap < KS * KS) tap_setup(tap);
        }
    };
    auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                if constexpr (KD == 32)
                    acc[ns][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ns], fb[buf][pg], acc[ns][pg], 0, 0, 0);
                else
                    acc[ns][pg] = __builtin_amdgcn_mfma_f32_16x16x16f16(fa[buf][ns], fb[buf][pg], acc[ns][pg], 0, 0, 0);
            }
    };
    const int nsteps = KS * KS * (d.cin / KD);
    tap_setup(0);
    load(0);
    for (int s = 0; s < nsteps; s += 2) {
        if (s + 1 < nsteps) load(1);
        mma(0);
        if (s + 1 < nsteps) {
            if (s + 2 < nsteps) load(0);
            mma(1);
        }
    }
    const half4v zero4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int n = n0 + ns * 16 + 4 * g;
        const float4v bv = *(const float4v*)(d.b + n);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            if (!pv[pg]) continue;
            const long pix = ((long)pb[pg] * d.Ho + py[pg]) * d.Wo + px[pg];
            half4v o;
            half4v rr = zero4;
            if (d.r) rr = *(const half4v*)(d.r + pix * d.rc + d.rc0 + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[ns][pg][j] + bv[j];
                if (d.act) v = v / (1.0f + __expf(-v));
                if (d.r) v += (float)rr[j];          // the bottleneck adds its input to the ACTIVATED map
                o[j] = (half_t)v;
            }
            *(half4v*)(d.y + pix * d.yc + d.yc0 + n) = o;
        }
    }
}

// 5 x 5 / 1 max pool with "same" padding over C channels of a slice, 8 channels per thread
__global__ __launch_bounds__(256) void det_maxpool5_kernel(const half_t* __restrict__ x, int xc, int xc0,
                                                           half_t* __restrict__ y, int yc, int yc0, int C, int H, int W,
                                                           long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C / 8;
    const int ch = (int)(i % c8) * 8;
    const long p = i / c8;
    const int px = (int)(p % W), py = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    half8 m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (half_t)-65504.f;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            const half8 v = *(const half8*)(x + ((b * H + yy) * W + xx) * xc + xc0 + ch);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
    }
    *(half8*)(y + ((b * H + py) * W + px) * yc + yc0 + ch) = m;
}

// nearest-neighbour x2 of a [B,H,W,C] slice into a [B,2H,2W,.] slice
__global__ __launch_bounds__(256) void det_upsample2_kernel(const half_t* __restrict__ x, int xc, int xc0,
                                                            half_t* __restrict__ y, int yc, int yc0, int C, int H, int W,
                                                            long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C / 8;
    const int ch = (int)(i % c8) * 8;
    const long p = i / c8;
    const int W2 = 2 * W, H2 = 2 * H;
    const int px = (int)(p % W2), py = (int)((p / W2) % H2);
    const long b = p / ((long)W2 * H2);
    *(half8*)(y + ((b * H2 + py) * W2 + px) * yc + yc0 + ch) =
        *(const half8*)(x + ((b * H + (py >> 1)) * W + (px >> 1)) * xc + xc0 + ch);
}

// Letterbox of a BGR frame onto the ch x cw canvas (multiples of 32, at most 640): bilinear resize with half-pixel centres
// (float32, rounded to whole levels as a uint8 resize returns them), 114 outside, RGB order,
// / 255, written as 4 channels (RGB and a zero), 8 bytes per pixel.
__global__ __launch_bounds__(256) void det_letterbox_kernel(const uint8_t* __restrict__ imgs,
                                                            const DetImageMeta* __restrict__ meta,
                                                            half_t* __restrict__ out, int ch, int cw) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;       // canvas pixel
    if (p >= ch * cw) return;
    const DetImageMeta m = meta[b];
    const int cy = p / cw, cx = p - cy * cw;
    float rgb[3] = {114.f, 114.f, 114.f};
    const int y = cy - m.top, x = cx - m.left;
    if (y >= 0 && y < m.nh && x >= 0 && x < m.nw) {
        const uint8_t* src = imgs + m.offset;
        float bgr[3];
        if (m.nh == m.h && m.nw == m.w) {
            const uint8_t* q = src + (long)y * m.stride + 3 * x;
            bgr[0] = q[0]; bgr[1] = q[1]; bgr[2] = q[2];
        } else {
            const double fy = ((double)y + 0.5) * m.sy - 0.5, fx = ((double)x + 0.5) * m.sx - 0.5;
            const double y0d = floor(fy), x0d = floor(fx);
            const float ay = (float)(fy - y0d), ax = (float)(fx - x0d);
            const int y0 = (int)y0d, x0 = (int)x0d;
            const int y0c = min(max(y0, 0), m.h - 1), y1c = min(max(y0 + 1, 0), m.h - 1);
            const int x0c = min(max(x0, 0), m.w - 1), x1c = min(max(x0 + 1, 0), m.w - 1);
            const uint8_t* r0 = src + (long)y0c * m.stride;
            const uint8_t* r1 = src + (long)y1c * m.stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t0 = (float)r0[3 * x0c + c] * (1.0f - ax) + (float)r0[3 * x1c + c] * ax;
                const float t1 = (float)r1[3 * x0c + c] * (1.0f - ax) + (float)r1[3 * x1c + c] * ax;
                bgr[c] = floorf(t0 * (1.0f - ay) + t1 * ay + 0.5f);
            }
        }
        rgb[0] = bgr[2]; rgb[1] = bgr[1]; rgb[2] = bgr[0];
    }
    half4v o;
    o[0] = (half_t)(rgb[0] / 255.0f); o[1] = (half_t)(rgb[1] / 255.0f); o[2] = (half_t)(rgb[2] / 255.0f); o[3] = (half_t)0.f;
    *(half4v*)(out + ((long)b * ch * cw + p) * 4) = o;
}

// The first convolution (3 -> 16 channels, 3 x 3, stride 2) on the 4-channel canvas: K = 9 taps x 4
// channels = 36, padded to 64, i.e. two 16x16x32 MFMAs per 16 pixels x 16 channels, the pixel operand
// gathered tap pair by tap pair with 8-byte loads (the same shape as conv1_1 of the descriptor
// network).  As 16 zero-padded input channels through the generic kernel it was the most expensive
// launch of the network (0.18 ms per 32 frames) and the canvas four times as large.
__global__ __launch_bounds__(256) void det_conv0_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w64,
                                                        const float* __restrict__ bias, half_t* __restrict__ y, int M,
                                                        int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const half8 a0 = *(const half8*)(w64 + l15 * 64 + 8 * g);
    const half8 a1 = *(const half8*)(w64 + l15 * 64 + 32 + 8 * g);
    const float4v bv = *(const float4v*)(bias + 4 * g);
    const half4v z4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    const int t0 = 2 * g, t1 = 2 * g + 1;          // this lane's taps of the first K block; tap 8 (g == 0 only) of the second
    const int base = (blockIdx.x * 4 + wave) * 64;
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int m = base + pg * 16 + l15;
        const bool pv = m < M;
        const int mm = pv ? m : M - 1;
        const int b = mm / (Ho * Wo), rem = mm - b * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const half_t* img = x + (long)b * H * W * 4;
        auto tap = [&](int t) __attribute__((always_inline)) -> half4v {
            const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
            const bool ok = iy >= 0 && ix >= 0;                 // (the far edges stay inside: H and W are even)
            const half4v v = *(const half4v*)(img + ((long)(ok ? iy : 0) * W + (ok ? ix : 0)) * 4);
            return ok ? v : z4;
        };
        const half4v v0 = tap(t0), v1 = tap(t1), v8 = tap(8);
        const half_t z = (half_t)0.f;
        const half8 bf0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        half8 bf1 = {v8[0], v8[1], v8[2], v8[3], z, z, z, z};
        if (g != 0) bf1 = half8{z, z, z, z, z, z, z, z};         // weights are zero beyond k = 35 anyway; keeps NaN-free
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf1, acc, 0, 0, 0);
        half4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = acc[j] + bv[j];
            o[j] = (half_t)(v / (1.0f + __expf(-v)));
        }
        if (pv) *(half4v*)(y + (long)mm * 16 + 4 * g) = o;
    }
}

// Head maps of the three levels -> the anchors whose class score exceeds `conf`: xywh on the canvas,
// score, five keypoints (x, y, visibility), anchor number.  One thread per anchor; the survivors
// of an image are appended through one counter per image (their order is made definite on the host).
__global__ __launch_bounds__(256) void det_decode_kernel(DetHead h0, DetHead h1, DetHead h2, float conf,
                                                         DetCand* __restrict__ cand, int* __restrict__ counts, int cap) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int n0 = h0.h * h0.w, n1 = h1.h * h1.w, n2 = h2.h * h2.w;
    if (a >= n0 + n1 + n2) return;
    const DetHead h = a < n0 ? h0 : (a < n0 + n1 ? h1 : h2);
    const int i = a < n0 ? a : (a < n0 + n1 ? a - n0 : a - n0 - n1);
    const int cy = i / h.w, cx = i - cy * h.w;
    const long pix = ((long)b * h.h + cy) * h.w + cx;
    const float logit = (float)h.cls[pix * 16];
    const float score = 1.0f / (1.0f + __expf(-logit));
    if (!(score > conf)) return;
    float dist[4];
    const half_t* bp = h.box + pix * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16], mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = (float)bp[s * 16 + k]; mx = fmaxf(mx, v[k]); }
        float sum = 0.f, ex = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float e = __expf(v[k] - mx); sum += e; ex += e * (float)k; }
        dist[s] = ex / sum;
    }
    const float ax = (float)cx + 0.5f, ay = (float)cy + 0.5f, st = (float)h.stride;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const int slot = atomicAdd(counts + b, 1);
    if (slot >= cap) return;
    DetCand c;
    c.v[0] = (x1 + x2) * 0.5f * st; c.v[1] = (y1 + y2) * 0.5f * st; c.v[2] = (x2 - x1) * st; c.v[3] = (y2 - y1) * st;
    c.v[4] = score;
    const half_t* kp = h.kpt + pix * 16;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        c.v[5 + 3 * k] = ((float)kp[3 * k] * 2.0f + (ax - 0.5f)) * st;
        c.v[6 + 3 * k] = ((float)kp[3 * k + 1] * 2.0f + (ay - 0.5f)) * st;
        c.v[7 + 3 * k] = 1.0f / (1.0f + __expf(-(float)kp[3 * k + 2]));
    }
    c.anchor = a;
    cand[(long)b * cap + slot] = c;
}

// ---- face detector ----
hipError_t launch_det_conv(const DetConv& d, int B, hipStream_t s) {
    if (B <= 0 || !d.x || !d.w || !d.b || !d.y || !d.zeros || d.cin % 16 || d.cout % 16 || d.cin <= 0 || d.cout <= 0 || d.cin > 2048)
        return hipErrorInvalidValue;
    if ((d.xc0 | d.yc0 | d.xc | d.yc) % 4 || (d.r && (d.rc0 | d.rc) % 4)) return hipErrorInvalidValue;   // 8-byte accesses
    const long M = (long)B * d.Ho * d.Wo;
    if (M <= 0 || M > 0x7fffffffL) return hipErrorInvalidValue;
    // 16-byte operand loads need the slices' first channels and widths to be multiples of 8
    const bool k32 = d.cin % 32 == 0 && d.xc0 % 8 == 0 && d.xc % 8 == 0;
    const bool big = M >= 4 * 64 * 256;                 // enough pixels for a block per CU at 64 per wave
    const int ppb = big ? 256 : 128;                    // pixels per block
    const int ns = d.cout >= 64 ? 4 : d.cout / 16;      // 16, 32 or 48 channels: one block column of that width
    if (d.cout > 64 && d.cout % 64) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((M + ppb - 1) / ppb), (unsigned)((d.cout + 16 * ns - 1) / (16 * ns)));
#define FR_DETN(KS, ST, KD, PG)                                                                                       \
    do {                                                                                                              \
        if (ns == 4) hipLaunchKernelGGL((det_conv_kernel<KS, ST, KD, PG, 4>), grid, dim3(256), 0, s, d, (int)M);       \
        else if (ns == 3) hipLaunchKernelGGL((det_conv_kernel<KS, ST, KD, PG, 3>), grid, dim3(256), 0, s, d, (int)M);  \
        else if (ns == 2) hipLaunchKernelGGL((det_conv_kernel<KS, ST, KD, PG, 2>), grid, dim3(256), 0, s, d, (int)M);  \
        else hipLaunchKernelGGL((det_conv_kernel<KS, ST, KD, PG, 1>), grid, dim3(256), 0, s, d, (int)M);               \
    } while (0)
#define FR_DETC(KS, ST)                                  \
    do {                                                 \
        if (k32 && big) FR_DETN(KS, ST, 32, 4);          \
        else if (k32) FR_DETN(KS, ST, 32, 2);            \
        else if (big) FR_DETN(KS, ST, 16, 4);            \
        else FR_DETN(KS, ST, 16, 2);                     \
    } while (0)
    if (d.k == 3 && d.stride == 1) FR_DETC(3, 1);
    else if (d.k == 3 && d.stride == 2) FR_DETC(3, 2);
    else if (d.k == 1 && d.stride == 1) FR_DETC(1, 1);
    else return hipErrorInvalidValue;
#undef FR_DETC
#undef FR_DETN
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

// x [B,H,W,4] -> y [B,H/2,W/2,16]; w64 [16][64] with k = tap * 4 + channel (zero elsewhere)
hipError_t launch_det_conv0(const void* x, const void* w64, const float* bias, void* y, int B, int H, int W, hipStream_t s) {
    if (B <= 0 || !x || !w64 || !bias || !y || H <= 0 || W <= 0 || (H | W) % 2) return hipErrorInvalidValue;
    const long M = (long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(det_conv0_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, (const half_t*)x,
                       (const half_t*)w64, bias, (half_t*)y, (int)M, H, W);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_det_maxpool5(const void* x, int xc, int xc0, void* y, int yc, int yc0, int C, int H, int W, int B,
                               hipStream_t s) {
    if (B <= 0 || !x || !y || C % 8 || (xc | xc0 | yc | yc0) % 8) return hipErrorInvalidValue;
    const long total = (long)B * H * W * (C / 8);
    hipLaunchKernelGGL(det_maxpool5_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const half_t*)x, xc,
                       xc0, (half_t*)y, yc, yc0, C, H, W, total);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_det_upsample2(const void* x, int xc, int xc0, void* y, int yc, int yc0, int C, int H, int W, int B,
                                hipStream_t s) {
    if (B <= 0 || !x || !y || C % 8 || (xc | xc0 | yc | yc0) % 8) return hipErrorInvalidValue;
    const long total = (long)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(det_upsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const half_t*)x, xc,
                       xc0, (half_t*)y, yc, yc0, C, H, W, total);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_det_letterbox(const uint8_t* imgs, const DetImageMeta* meta, void* out4, int ch, int cw, int B, hipStream_t s) {
    if (B <= 0 || !imgs || !meta || !out4 || ch <= 0 || cw <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(det_letterbox_kernel, dim3((unsigned)((ch * cw + 255) / 256), B), dim3(256), 0, s, imgs, meta,
                       (half_t*)out4, ch, cw);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

hipError_t launch_det_decode(const DetHead* heads3, float conf, DetCand* cand, int* counts, int cap, int B, hipStream_t s) {
    if (B <= 0 || !heads3 || !cand || !counts || cap <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return e;
    const int n = heads3[0].h * heads3[0].w + heads3[1].h * heads3[1].w + heads3[2].h * heads3[2].w;
    hipLaunchKernelGGL(det_decode_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, heads3[0], heads3[1], heads3[2], conf,
                       cand, counts, cap);
    FR_CHECK_LAUNCH();
    return hipSuccess;
}

}  // namespace frhot
