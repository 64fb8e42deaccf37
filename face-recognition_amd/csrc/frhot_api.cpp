// C ABI of libfrhot.so (see include/frhot.h): contexts, model loading, the embed
// pipeline, galleries and search.  Host-side only; kernels live in frhot_kernels.hip.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is bound at run time (dlopen)
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <string>
#include <algorithm>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/frhot.h"
#include "frhot_kernels.h"

using namespace frhot;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// No exception may cross the C ABI: every extern "C" body runs inside guarded().
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(FRHOT_ECAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(FRHOT_EHIP, "unexpected failure: %s", e.what());
    } catch (...) {
        return fail(FRHOT_EHIP, "unexpected failure");
    }
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(FRHOT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct LayerSpec { const char* name; int hw, cin, cout, pool, opad; };
// conv1_1 is handled by its own kernel; the rest go through the implicit GEMM.
const LayerSpec kConv[] = {
    {"conv1_1", 224, 3, 64, 0, 1},   {"conv1_2", 224, 64, 64, 1, 1},  {"conv2_1", 112, 64, 128, 0, 1},
    {"conv2_2", 112, 128, 128, 1, 1}, {"conv3_1", 56, 128, 256, 0, 1}, {"conv3_2", 56, 256, 256, 0, 1},
    {"conv3_3", 56, 256, 256, 1, 1},  {"conv4_1", 28, 256, 512, 0, 1}, {"conv4_2", 28, 512, 512, 0, 1},
    {"conv4_3", 28, 512, 512, 1, 1},  {"conv5_1", 14, 512, 512, 0, 1}, {"conv5_2", 14, 512, 512, 0, 1},
    {"conv5_3", 14, 512, 512, 1, 0},
};
constexpr int kNumConv = 13;
constexpr int kFc6K = 7 * 7 * 512, kFcN = 4096;
constexpr int kFc6Split = 14, kFc7Split = 8;

// Weight blob: "FRHOTW01", u32 version, u32 count, then `count` entries of
// {char name[16]; u32 dtype (0 f16, 1 f32); u32 ndim; u32 dims[4]; u64 offset; u64 nbytes}.
struct BlobEntry { char name[16]; uint32_t dtype, ndim, dims[4]; uint64_t offset, nbytes; };
static_assert(sizeof(BlobEntry) == 56, "blob entry layout");

const BlobEntry* blob_find(const uint8_t* blob, size_t nbytes, const char* name) {
    if (nbytes < 16 || memcmp(blob, "FRHOTW01", 8) != 0) return nullptr;
    uint32_t count;
    memcpy(&count, blob + 12, 4);
    if (16 + (size_t)count * sizeof(BlobEntry) > nbytes) return nullptr;
    const BlobEntry* e = (const BlobEntry*)(blob + 16);
    for (uint32_t i = 0; i < count; ++i)
        if (strncmp(e[i].name, name, 16) == 0) {
            if (e[i].offset + e[i].nbytes > nbytes) return nullptr;
            return &e[i];
        }
    return nullptr;
}

size_t act_elems(const LayerSpec& l, int B) {
    const int ho = l.pool ? l.hw / 2 : l.hw;
    const int p = ho + 2 * l.opad;
    return (size_t)B * p * p * l.cout;
}

}  // namespace

// Page-locked staging of one by-key search's index arrays (grown on demand, freed with its owner).
struct IndexPin {
    int32_t* p = nullptr;
    size_t cap = 0;         // int32 entries
    hipEvent_t copied = nullptr;    // recorded behind the upload out of `p`; waited for before `p` is written again
    bool pending = false;
};

struct frhot_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int max_batch = 0;
    std::mutex mu;
    bool model = false;
    float input_scale = 1.0f;
    // weights
    void* w[kNumConv] = {};      // fp16 [Cout][9*Cin]; w[0] is the packed [64][64] conv1_1 matrix
    float* bias[kNumConv] = {};
    void *w6 = nullptr, *w7 = nullptr;
    float *b6 = nullptr, *b7 = nullptr;
    // activations and scratch
    void* x0 = nullptr;          // fp16 [B,226,226,4]
    void* act[kNumConv] = {};
    uint8_t* u8 = nullptr;       // [B,224,224,3]
    int* img_max = nullptr;
    CropMeta* meta_dev = nullptr;
    uint8_t* lut_dev = nullptr;  // [2][256]: identity, gamma 1.5
    void* fc6_out = nullptr;     // fp16 [B,4096]
    float* partial = nullptr;    // split-K partial sums
    // host-call staging
    // Host calls (frhot_embed, frhot_recognise) go through one of two staging slots: a page-locked
    // mirror and a device buffer for the packed crops, a page-locked mirror for the answers, and
    // two events.  A caller holds its slot for the whole call; packing and the upload (copy
    // stream) happen before it takes `mu`, so with two callers in flight the crops of batch i+1
    // are packed and uploaded while batch i computes.
    struct Slot {
        std::mutex m;
        uint8_t* pin = nullptr; size_t pin_cap = 0;
        uint8_t* dev = nullptr; size_t dev_cap = 0;
        uint8_t* res_pin = nullptr;                  // [B,16] f32 scores, then [B,16] i64 rows
        float* emb_pin = nullptr;                    // [B,4096], allocated at the first request for descriptors
        int64_t* sharp_pin = nullptr;                // [B,2] Laplacian sums of the crops
        IndexPin idx;                                // query order / keys / tile lists of a by-key search
        hipEvent_t uploaded = nullptr, uploaded0 = nullptr, done = nullptr;   // everything / the first half of a split batch is on its way
    } slot[2];
    std::atomic<unsigned> next_slot{0};
    std::atomic<int> host_calls{0};                  // host calls (embed / recognise) inside the library right now
    hipStream_t copy_stream = nullptr;
    hipStream_t side_stream = nullptr; hipEvent_t fork = nullptr, join = nullptr;   // second half of a batch's conv stack
    float* emb_dev = nullptr;    // [B,4096]
    void* q_dev = nullptr;       // fp16 [B,4096]
    int32_t* qkeys_dev = nullptr;
    float* res_s = nullptr; int64_t* res_r = nullptr;  // [B,16]
    int64_t* sharp_dev = nullptr;                      // [B,2]
    std::vector<CropMeta> meta_host;
    // the communicator of a row-sharded search and its exchange buffers (frhot_dist_init)
    void* comm = nullptr; int dist_rank = 0, dist_world = 0;
    void* xq = nullptr; size_t xq_cap = 0;           // every rank's queries, fp16 [world*B, D]
    int32_t* xk = nullptr; size_t xk_cap = 0;        // every rank's keys [world*B]
    int64_t* xblk = nullptr; size_t xblk_cap = 0;    // this shard's answer block, then the world's blocks (int64 words)
    // timing: per stage a ring of (begin, end) event pairs; n[i] counts recorded pairs
    bool timing = false;
    static constexpr int kRing = 128;
    hipEvent_t ev_b[5][kRing] = {}, ev_e[5][kRing] = {};
    long ev_n[5] = {};
    int stage_begin(int i) {
        if (!timing) return 0;
        return hipEventRecord(ev_b[i][ev_n[i] % kRing], stream) == hipSuccess ? 0 : 1;
    }
    int stage_end(int i) {
        if (!timing) return 0;
        const int rc = hipEventRecord(ev_e[i][ev_n[i] % kRing], stream) == hipSuccess ? 0 : 1;
        ++ev_n[i];
        return rc;
    }
};

static void dist_release(frhot_ctx* c);

struct frhot_gallery {
    frhot_ctx* ctx = nullptr;
    int D = 0;
    int64_t capacity = 0, nrows = 0;
    bool dense = true;           // every row in [0, nrows) is live: the scan may skip the key test
    void* rows = nullptr;
    int32_t* keys = nullptr;
    float* part_s = nullptr; int64_t* part_r = nullptr; size_t part_cap = 0;
    unsigned* qbound = nullptr; size_t qbound_cap = 0;   // cross-block score bounds of the scan
    // host-call staging
    float* up_vecs = nullptr; int32_t* up_keys = nullptr; int64_t* up_rows = nullptr; int up_cap = 0;
    void* q16 = nullptr; float* q32 = nullptr; int32_t* qk = nullptr; float* out_s = nullptr; int64_t* out_r = nullptr;
    int q_cap = 0;
    // staging of tombstone / read / write_f16 (rows, keys, fp16 rows), grown on demand and kept
    // while small, so that enrol / evict / delete calls allocate nothing
    int64_t* mut_rows = nullptr; int32_t* mut_keys = nullptr; void* mut_f16 = nullptr;
    size_t mut_cap = 0, mut_f16_cap = 0;
    // Which 256-row tiles may hold rows of a key: ascending tile numbers per key, kept on the host
    // under ctx->mu.  A superset is enough (rows that died or changed key stay listed; the scan
    // still tests every row's key), so only writes add to it.  The reference keeps every collection
    // in storage of its own and searches that alone (qdrant_database_FE/app.py:316-337); a search
    // whose queries carry keys does the same here by scanning only its keys' tiles.
    std::unordered_map<int32_t, std::vector<int32_t>> key_tiles;
    bool key_tiles_valid = true;    // false: keys were written on the device (synthetic fill): read them back first
    // Keys strewn over every tile would make the lists as large as keys x tiles and help no search:
    // past key_tiles_limit() entries the lists are dropped for good and searches scan the whole gallery.
    size_t key_tiles_entries = 0;
    bool key_tiles_off = false;
    size_t key_tiles_limit() const { return 8 * (size_t)((capacity + 255) / 256) + 65536; }
    int search_mode = FRHOT_SEARCH_AUTO;
    IndexPin idx;                   // of frhot_search (a call holds ctx->mu until its copies are done)
    int32_t* idx_dev = nullptr; size_t idx_dev_cap = 0;
    void* q16s = nullptr; size_t q16s_cap = 0;      // the queries in key order
    // page-locked mirror of a small frhot_search call's queries and answers (up to kSmallSearch
    // queries): copies out of and into pageable memory cost tens of microseconds each, most of
    // what a search inside one small collection takes
    uint8_t* io_pin = nullptr;
};

namespace {

int fill_crop_meta(int h, int w, int stride, int64_t offset, CropMeta* m) {
    if (h <= 0 || w <= 0 || stride < 3 * w) return fail(FRHOT_EINVAL, "bad crop geometry %dx%d stride %d", h, w, stride);
    const double f0 = 224.0 / (double)h, f1 = 224.0 / (double)w;
    const double f = f0 < f1 ? f0 : f1;
    const int dw = (int)((double)w * f), dh = (int)((double)h * f);
    if (dw <= 0 || dh <= 0 || dw > 224 || dh > 224)
        return fail(FRHOT_EINVAL, "crop %dx%d resizes to an empty image", h, w);
    m->offset = offset; m->h = h; m->w = w; m->stride = stride;
    m->dw = dw; m->dh = dh;
    m->pady = (224 - dh) / 2; m->padx = (224 - dw) / 2;
    m->area2x = (w == 2 * dw && h == 2 * dh) ? 1 : 0;
    m->scale_x = 1.0 / ((double)dw / (double)w);
    m->scale_y = 1.0 / ((double)dh / (double)h);
    return FRHOT_OK;
}

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Free a device buffer and forget it in one step, so that a failure further on never leaves a
// freed pointer behind.
template <class T>
hipError_t free_null(T*& p) {
    if (!p) return hipSuccess;
    T* q = p;
    p = nullptr;
    return hipFree(q);
}

int upload(void** dst, const void* src, size_t n) {
    HIP_TRY(hipMalloc(dst, n));
    HIP_TRY(hipMemcpy(*dst, src, n, hipMemcpyHostToDevice));
    return FRHOT_OK;
}

constexpr int kSplitMin = 32;   // batches from this size on run their conv stack as two halves on two streams

// What embed_enqueue asks before it enqueues the second half of a batch: a host call whose crops
// are still being packed and uploaded lets the first half start on the GPU meanwhile.
struct SecondHalfGate {
    virtual int ready(hipStream_t side_stream) = 0;     // blocks until the rest is on its way; orders side_stream behind it
    virtual ~SecondHalfGate() = default;
};

constexpr int kShareLayers = 6;     // conv layers 1..6 (conv1_2 .. conv3_3) are launched with a share of the CUs when a batch runs in two parts
constexpr int kShareMinBatch = 192, kShareMaxBatch = 320;

// Enqueue the whole embed pipeline for B crops whose pixels sit at crops_dev.
int embed_enqueue(frhot_ctx* c, const void* crops_dev, const int64_t* offsets, const int32_t* hw,
                  const int32_t* strides, int B, unsigned flags, float* out_f32, void* out_q, int Dq,
                  SecondHalfGate* gate = nullptr, int first_part = 0, bool parts_start_together = true) {
    if (!c->model) return fail(FRHOT_ESTATE, "frhot_model_load has not been called");
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    if (!crops_dev || !offsets || !hw || !strides) return fail(FRHOT_EINVAL, "null crop arguments");
    if (out_q && (Dq <= 0 || Dq > kFcN || Dq % 64)) return fail(FRHOT_EINVAL, "Dq %d must be a multiple of 64 in 64..4096", Dq);
    c->meta_host.resize(B);
    for (int i = 0; i < B; ++i) {
        int rc = fill_crop_meta(hw[2 * i], hw[2 * i + 1], strides[i], offsets[i], &c->meta_host[i]);
        if (rc) return rc;
    }
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(c->meta_dev, c->meta_host.data(), sizeof(CropMeta) * B, hipMemcpyHostToDevice, s));
    // The whole stack of a batch of kSplitMin faces or more runs as two halves on two streams (fork
    // behind the copy of the crop table, join before fc6).  Every conv layer is a persistent launch of
    // one block per CU whose last round leaves CUs idle (conv5_x: 392 tiles on 256 CUs); with a second
    // stream the other half's next launch takes those CUs at once.  Faces are independent, so the
    // outputs are the same bits (256 faces: conv stack 7.31 -> 7.06 ms; 48 faces 1.69 -> 1.50; 16 faces
    // equal; three or four parts were slower than one).  Each half resizes its own crops, so a host
    // call's first half computes while its second half is still being packed and uploaded (`gate`).
    const int nparts = B >= kSplitMin ? 2 : 1;
    if (nparts == 1 && gate) {
        const int rc = gate->ready(s);
        if (rc) return rc;
    }
    if (nparts > 1 && !c->fork) {
        HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    }
    if (nparts > 1) HIP_TRY(hipEventRecord(c->fork, s));
    const uint8_t* lut = c->lut_dev + ((flags & FRHOT_FLAG_GAMMA) ? 256 : 0);
    for (int part = 0; part < nparts; ++part) {
        // the two parts need not be equal: a host call starts with a smaller one, whose crops arrive sooner
        const int cut = first_part > 0 && first_part < B ? first_part : B / 2;
        const int b0 = part == 0 ? 0 : cut, n = nparts == 1 ? B : (part == 0 ? cut : B - cut);
        hipStream_t hs = part == 0 ? s : c->side_stream;
        // The launches of the large early layers (through conv3_3) take the part's share of the CUs as
        // persistent blocks instead of one per CU: the other part's launch of the same layer then runs
        // beside it from the start, not in the gaps its blocks leave (7.01 -> 6.95 ms per 256 faces);
        // from conv4_1 on a launch has few rounds of tiles and fills the chip better at full width.
#ifdef FR_NO_SHARE
        const int share = 0;
#else
        // Not for a host call that is alone in the library (its first part runs by itself while the second is
        // still being uploaded; a call that found another in flight has its crops on the device by the time
        // the GPU reaches it) and not for smaller batches, whose launches have too few rounds of tiles for half the CUs
        // (conv stack at 32 / 64 / 128 faces 1.05 / 1.83 / 3.47 ms without shares, 1.12 / 1.87 / 3.55 with; 192 faces 5.23 against 5.05),
        // nor for larger ones, whose launches have rounds enough to fill the chip either way (embed of 256 / 320 faces
        // 7.24 -> 7.18 / 9.02 -> 8.98 ms with shares, 384 / 512 / 1024 faces 10.88 -> 10.93 / 13.81 -> 13.94 / 26.65 -> 26.88).
        const int share = nparts > 1 && parts_start_together && B >= kShareMinBatch && B <= kShareMaxBatch
                              ? (256 * n / B + 7) / 8 * 8 : 0;
#endif
        if (part == 1) {
            if (gate) {
                const int rc = gate->ready(hs);
                if (rc) return rc;
            }
            HIP_TRY(hipStreamWaitEvent(hs, c->fork, 0));
        }
        // stage 0 times the first half's preprocessing, stage 1 everything up to the join
        if (part == 0 && c->stage_begin(0)) return fail(FRHOT_EHIP, "event record failed");
        char* const x0 = (char*)c->x0 + (size_t)b0 * 226 * 226 * 4 * 2;
        uint8_t* const u8 = c->u8 + (size_t)b0 * 224 * 224 * 3;
        HIP_TRY(launch_resize_u8((const uint8_t*)crops_dev, c->meta_dev + b0, lut, (flags & FRHOT_FLAG_NOFLIP) ? 0 : 1, u8,
                                 c->img_max + b0, c->input_scale, x0, n, hs));
        HIP_TRY(launch_normalize_pack(u8, c->img_max + b0, c->input_scale, x0, n, hs, true));
        if (part == 0 && (c->stage_end(0) || c->stage_begin(1))) return fail(FRHOT_EHIP, "event record failed");
#ifdef FR_NO_FUSE1      // A/B builds only: the first two layers as two launches (ctx_init allocates act[0] in such a build)
        HIP_TRY(launch_conv1_1(x0, c->w[0], c->bias[0], (char*)c->act[0] + act_elems(kConv[0], b0) * 2, n, hs));
        {
            ConvLayerDesc d{kConv[1].hw, kConv[1].cin, kConv[1].cout, kConv[1].pool, kConv[1].opad, 1};
            HIP_TRY(launch_conv3x3(d, (const char*)c->act[0] + act_elems(kConv[0], b0) * 2, c->w[1], c->bias[1],
                                   (char*)c->act[1] + act_elems(kConv[1], b0) * 2, c->partial, n, hs));
        }
#else
        // the first two layers are one launch: conv1_1's 6.4 MB per face live only as LDS patches of conv1_2
        HIP_TRY(launch_conv1_fused(x0, c->w[0], c->bias[0], c->w[1], c->bias[1], (char*)c->act[1] + act_elems(kConv[1], b0) * 2,
                                   c->partial, n, 1, hs, share));
#endif
        for (int l = 2; l < kNumConv; ++l) {
            // layers alternate direction so that each starts on what its predecessor wrote last
            ConvLayerDesc d{kConv[l].hw, kConv[l].cin, kConv[l].cout, kConv[l].pool, kConv[l].opad, l & 1};
            if (l <= kShareLayers) d.blocks = share;
            HIP_TRY(launch_conv3x3(d, (const char*)c->act[l - 1] + act_elems(kConv[l - 1], b0) * 2, c->w[l], c->bias[l],
                                   (char*)c->act[l] + act_elems(kConv[l], b0) * 2, c->partial, n, hs));
        }
    }
    if (nparts > 1) {
        HIP_TRY(hipEventRecord(c->join, c->side_stream));
        HIP_TRY(hipStreamWaitEvent(s, c->join, 0));
    }
    if (c->stage_end(1) || c->stage_begin(2)) return fail(FRHOT_EHIP, "event record failed");
    HIP_TRY(launch_fc_splitk(c->act[kNumConv - 1], c->w6, c->partial, B, kFcN, kFc6K, kFc6Split, s));
    HIP_TRY(launch_fc_reduce_relu(c->partial, c->b6, c->fc6_out, B, kFcN, kFc6Split, s));
    HIP_TRY(launch_fc_splitk(c->fc6_out, c->w7, c->partial, B, kFcN, kFcN, kFc7Split, s));
    HIP_TRY(launch_fc_reduce_relu_l2(c->partial, c->b7, out_f32, out_q, out_q ? Dq : kFcN, B, kFcN, kFc7Split, s));
    if (c->stage_end(2)) return fail(FRHOT_EHIP, "event record failed");
    return FRHOT_OK;
}

// Enqueue the Laplacian sums of B crops (the blur gate's measure) into out_dev ([B,2] int64).
// `meta_ready`: embed_enqueue of the same crops went before on this stream, c->meta_dev holds them.
int sharpness_enqueue(frhot_ctx* c, const void* crops_dev, const int64_t* offsets, const int32_t* hw,
                      const int32_t* strides, int B, int64_t* out_dev, bool meta_ready) {
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    if (!crops_dev || !offsets || !hw || !strides || !out_dev) return fail(FRHOT_EINVAL, "null crop arguments");
    int max_tiles = 0;
    for (int i = 0; i < B; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        if (h <= 0 || w <= 0 || h > 16384 || w > 16384 || strides[i] < 3 * w)
            return fail(FRHOT_EINVAL, "bad crop %d: %dx%d stride %d", i, h, w, strides[i]);
        const int t = ((w + 255) / 256) * ((h + 15) / 16);
        if (t > max_tiles) max_tiles = t;
    }
    if (!meta_ready) {
        c->meta_host.assign(B, CropMeta{});
        for (int i = 0; i < B; ++i) {
            CropMeta& m = c->meta_host[i];
            m.offset = offsets[i]; m.h = hw[2 * i]; m.w = hw[2 * i + 1]; m.stride = strides[i];
        }
        HIP_TRY(hipMemcpyAsync(c->meta_dev, c->meta_host.data(), sizeof(CropMeta) * B, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(launch_crop_sharpness((const uint8_t*)crops_dev, c->meta_dev, max_tiles, out_dev, B, c->stream));
    return FRHOT_OK;
}

// Packs host crops tightly into the slot's page-locked mirror and uploads them to the slot's device
// buffer on the copy stream.  The caller holds sl.m, and the slot's previous user has waited for
// its work before releasing it, so mirror and device buffer are free.  Never touches the compute
// stream.
//
// The callers' arrays are pageable, and a strided copy out of pageable memory moves row by row.
// The crops are packed by a few host threads, each a contiguous run of crops; a thread sends what
// it has packed to the device itself, in pieces of about a MiB, so the upload runs beside the
// packing instead of behind it.  With `split_at` > 0 the crops below that index go first:
// begin() returns once THEY are on their way (`sl.uploaded0` recorded behind their copies) while
// the worker threads go on with the rest, and ready() - called by embed_enqueue before it
// enqueues the batch's second half - joins them and records `sl.uploaded`.  The first half of a
// batch is then resized and convolved while the second half is still crossing PCIe.
struct CropStaging final : SecondHalfGate {
    frhot_ctx* c;
    frhot_ctx::Slot& sl;
    const uint8_t* const* crops = nullptr;
    const int32_t* hw = nullptr;
    const int32_t* strides = nullptr;
    int B = 0, split_at = 0, nthr = 1;
    size_t total = 0;
    std::vector<int64_t> offsets;
    std::vector<int32_t> packed_strides;
    std::vector<int> first0, first1;            // thread t packs crops [first0[t], first0[t+1]) and then [first1[t], first1[t+1])
    std::vector<hipError_t> errs;
    std::vector<std::thread> workers;
    std::atomic<int> first_part_left{0};
    bool finished = false;
    static constexpr size_t kPiece = 1u << 20;

    CropStaging(frhot_ctx* ctx, frhot_ctx::Slot& slot) : c(ctx), sl(slot) {}
    // Workers are joined on every way out (a thread that is still joinable when its std::thread is
    // destroyed ends the process); nothing of this call may still read the mirror afterwards.
    ~CropStaging() override {
        join();
        if (!finished) (void)hipStreamSynchronize(c->copy_stream);
    }
    void join() { for (auto& w : workers) if (w.joinable()) w.join(); }

    void pack_range(int t, int lo, int hi) {
        if (lo >= hi || errs[t] != hipSuccess) return;
        uint8_t* const pin = sl.pin;
        uint8_t* const dev = sl.dev;
        size_t sent = (size_t)offsets[lo];              // bytes of the mirror below this are on their way
        for (int i = lo; i < hi; ++i) {
            const size_t h = (size_t)hw[2 * i], row = (size_t)3 * hw[2 * i + 1];
            uint8_t* dst = pin + offsets[i];
            if ((size_t)strides[i] == row) memcpy(dst, crops[i], h * row);
            else for (size_t y = 0; y < h; ++y) memcpy(dst + y * row, crops[i] + y * (size_t)strides[i], row);
            const size_t end = i + 1 < B ? (size_t)offsets[i + 1] : total;      // (padding bytes between crops travel too)
            if (end - sent >= kPiece || i + 1 == hi) {
                const hipError_t e = hipMemcpyAsync(dev + sent, pin + sent, end - sent, hipMemcpyHostToDevice, c->copy_stream);
                if (e != hipSuccess) { errs[t] = e; return; }
                sent = end;
            }
        }
    }
    // The thread that finishes the first part last records the event behind everybody's copies.
    void first_part_done(int t) {
        if (first_part_left.fetch_sub(1) == 1 && split_at > 0) {
            const hipError_t e = hipEventRecord(sl.uploaded0, c->copy_stream);
            if (e != hipSuccess) errs[t] = e;
            first_part_left.store(-1);
        }
    }
    void run(int t, bool worker) {
        if (worker && hipSetDevice(c->device) != hipSuccess) errs[t] = hipErrorInvalidDevice;
        pack_range(t, first0[t], first0[t + 1]);
        first_part_done(t);
        pack_range(t, first1[t], first1[t + 1]);
    }
    // crops [lo, hi) cut into n runs of about equal bytes
    std::vector<int> cut(int lo, int hi, int n) const {
        std::vector<int> f(n + 1, hi);
        f[0] = lo;
        if (lo >= hi) return f;
        const size_t base = (size_t)offsets[lo], bytes = (hi < B ? (size_t)offsets[hi] : total) - base;
        for (int t = 1, i = lo; t < n; ++t) {
            while (i < hi && (size_t)offsets[i] - base < bytes / n * t) ++i;
            f[t] = i;
        }
        return f;
    }

    int begin(const uint8_t* const* crops_, const int32_t* hw_, const int32_t* strides_, int B_, int split) {
        crops = crops_; hw = hw_; strides = strides_; B = B_;
        if (!crops || !hw || !strides) return fail(FRHOT_EINVAL, "null crop arguments");
        offsets.resize(B); packed_strides.resize(B);
        total = 0;
        for (int i = 0; i < B; ++i) {
            const int h = hw[2 * i], w = hw[2 * i + 1];
            if (h <= 0 || w <= 0 || h > 16384 || w > 16384 || strides[i] < 3 * w || !crops[i])
                return fail(FRHOT_EINVAL, "bad crop %d: %dx%d stride %d", i, h, w, strides[i]);
            offsets[i] = (int64_t)total;
            packed_strides[i] = 3 * w;
            total += ((size_t)h * 3 * w + 15) & ~(size_t)15;
        }
        if (total > sl.dev_cap) {
            sl.dev_cap = 0;
            HIP_TRY(free_null(sl.dev));
            const size_t cap = total + total / 4 + 4096;
            HIP_TRY(hipMalloc((void**)&sl.dev, cap));
            sl.dev_cap = cap;
        }
        if (total > sl.pin_cap) {
            sl.pin_cap = 0;
            if (uint8_t* old = sl.pin) { sl.pin = nullptr; HIP_TRY(hipHostFree(old)); }
            const size_t cap = total + total / 4 + 4096;
            HIP_TRY(hipHostMalloc((void**)&sl.pin, cap, hipHostMallocDefault));
            sl.pin_cap = cap;
        }
        nthr = (int)(total >> 22);                      // one thread per 4 MiB
        const int hwc = (int)std::thread::hardware_concurrency();
        const int lim = hwc >= 16 ? 8 : (hwc >= 4 ? hwc / 2 : 1);
        if (nthr > lim) nthr = lim;
        if (nthr > B) nthr = B;
        if (nthr < 1) nthr = 1;
        split_at = (split > 0 && split < B) ? split : 0;
        // The calling thread packs its share of the first part only when it will go on to enqueue the
        // first half's kernels; the workers share the rest between them.
        const int lead = split_at > 0 ? split_at : B;
        first0 = cut(0, lead, nthr);
        if (split_at > 0 && nthr > 1) {
            std::vector<int> w = cut(split_at, B, nthr - 1);
            first1.assign(nthr + 1, split_at);
            for (int t = 1; t <= nthr; ++t) first1[t] = w[t - 1];
        } else {
            first1.assign(nthr + 1, B);                 // (one thread and a split: the calling thread packs the rest in finish())
        }
        errs.assign(nthr, hipSuccess);
        first_part_left.store(nthr);
        workers.resize(nthr > 1 ? nthr - 1 : 0);
        for (int t = 1; t < nthr; ++t) {
            try {
                workers[t - 1] = std::thread(&CropStaging::run, this, t, true);
            } catch (const std::system_error&) {
                run(t, false);                          // process thread limit: packed by the calling thread instead
            }
        }
        pack_range(0, first0[0], first0[1]);
        first_part_done(0);
        if (split_at == 0) return finish();
        // until the last worker has sent its share of the first part (they all began together)
        while (first_part_left.load() >= 0) std::this_thread::yield();
        for (int t = 0; t < nthr; ++t)
            if (errs[t] != hipSuccess) return finish();
        return FRHOT_OK;
    }
    // Everything is on its way: `sl.uploaded` recorded behind the last copy.
    int finish() {
        if (finished) return FRHOT_OK;
        if (nthr == 1 && split_at > 0) pack_range(0, split_at, B);
        join();
        hipError_t err = hipSuccess;
        for (int t = 0; t < nthr; ++t) if (errs[t] != hipSuccess) err = errs[t];
        if (err == hipSuccess) err = hipEventRecord(sl.uploaded, c->copy_stream);
        if (err != hipSuccess) {
            (void)hipStreamSynchronize(c->copy_stream);
            finished = true;
            HIP_TRY(err);
        }
        finished = true;
        return FRHOT_OK;
    }
    int ready(hipStream_t stream) override {
        const int rc = finish();
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(stream, sl.uploaded, 0));
        return FRHOT_OK;
    }
};

// How many crops of a host call go first: a quarter of a large batch (the GPU starts after a quarter
// of the upload, and the remaining three quarters cross PCIe while that part is convolved: 64 faces
// take 1.9 ms, 46 MB about 1.2), half of a smaller one; below kSplitMin the batch is not split.
// With another host call in flight the GPU is busy with that call's batch while this one's crops
// upload, so nothing is gained by starting early and the equal halves (the faster conv schedule) are
// kept: two callers 9.97 ms per batch with halves against 10.3 with a quarter first, one caller
// 11.8 against 11.4.
int host_first_part(int B, bool alone) {
    return B >= 4 * kSplitMin && alone ? B / 4 : (B >= kSplitMin ? B / 2 : 0);
}
struct HostCall {
    frhot_ctx* c; bool alone;
    explicit HostCall(frhot_ctx* ctx) : c(ctx), alone(ctx->host_calls.fetch_add(1) == 0) {}
    ~HostCall() { c->host_calls.fetch_sub(1); }
};

// RAII over one staging slot: the two slots are taken in turn.
struct SlotLease {
    frhot_ctx::Slot& sl;
    explicit SlotLease(frhot_ctx* c) : sl(c->slot[c->next_slot.fetch_add(1) & 1u]) { sl.m.lock(); }
    ~SlotLease() { sl.m.unlock(); }
};

// Shared tail of the host calls, entered with the crops staged: under `mu`, make the compute
// stream wait for the upload, let `enqueue` put the kernels and the copies into the slot's answer
// mirror on the stream, record `done`; then, without the lock, wait for `done`.  On failure
// both streams are drained before the slot is released.
template <class F>
int run_staged(frhot_ctx* c, frhot_ctx::Slot& sl, CropStaging& st, F&& enqueue) {
    int rc;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        // a split staging has its first part on the way (the rest follows inside `enqueue`, through the gate)
        hipError_t e = hipStreamWaitEvent(c->stream, st.split_at > 0 ? sl.uploaded0 : sl.uploaded, 0);
        rc = e == hipSuccess ? enqueue() : fail(FRHOT_EHIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
        if (rc == FRHOT_OK && !st.finished) {
            // the enqueued work did not ask for the second part (cannot happen with embed_enqueue): order it here
            rc = st.ready(c->stream);
        }
        if (rc == FRHOT_OK) {
            e = hipEventRecord(sl.done, c->stream);
            if (e != hipSuccess) rc = fail(FRHOT_EHIP, "hipEventRecord failed: %s", hipGetErrorString(e));
        }
    }
    if (rc != FRHOT_OK) {
        const std::string keep = g_err;
        st.join();
        (void)hipStreamSynchronize(c->copy_stream);
        if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamSynchronize(c->stream);
        g_err = keep;
        return rc;
    }
    HIP_TRY(hipEventSynchronize(sl.done));
    return FRHOT_OK;
}

int slot_emb_mirror(frhot_ctx* c, frhot_ctx::Slot& sl) {
    if (!sl.emb_pin) HIP_TRY(hipHostMalloc((void**)&sl.emb_pin, (size_t)c->max_batch * kFcN * 4, hipHostMallocDefault));
    return FRHOT_OK;
}

// Answer of a search over an empty gallery: -inf / -1 everywhere, by two fills on the stream.
int search_fill_empty(hipStream_t s, int Bq, int k, float* scores_dev, int64_t* rows_dev) {
    HIP_TRY(hipMemsetAsync(rows_dev, 0xFF, sizeof(int64_t) * (size_t)Bq * k, s));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)scores_dev, (int)0xFF800000u, (size_t)Bq * k, s));   // -inf
    return FRHOT_OK;
}

// Scan + final selection with the candidate lists in part_s / part_r and the score bounds in qbound.
int search_launch(frhot_gallery* g, const void* q16, const int32_t* qkeys_dev, int Bq, int k, float* scores_dev,
                  int64_t* rows_dev, int64_t row_base, float* part_s, int64_t* part_r, unsigned* qbound) {
    frhot_ctx* c = g->ctx;
    hipStream_t s = c->stream;
    const int kl = scan_list_len(k);
    int nblk = 0;
    if (c->stage_begin(3)) return fail(FRHOT_EHIP, "event record failed");
    const int use_keys = (qkeys_dev != nullptr || !g->dense) ? 1 : 0;
    HIP_TRY(launch_scan_topk(g->rows, g->keys, g->nrows, g->D, q16, qkeys_dev, use_keys, Bq, k, part_s, part_r,
                             qbound, &nblk, s));
    if (c->stage_end(3) || c->stage_begin(4)) return fail(FRHOT_EHIP, "event record failed");
    HIP_TRY(launch_topk_final(part_s, part_r, nblk, kl, Bq, k, row_base, scores_dev, rows_dev, s));
    if (c->stage_end(4)) return fail(FRHOT_EHIP, "event record failed");
    return FRHOT_OK;
}

int search_enqueue(frhot_gallery* g, const void* q16, const int32_t* qkeys_dev, int Bq, int k, float* scores_dev,
                   int64_t* rows_dev, int64_t row_base) {
    frhot_ctx* c = g->ctx;
    if (Bq <= 0 || k <= 0 || k > 16) return fail(FRHOT_EINVAL, "Bq %d / k %d out of range (k <= 16)", Bq, k);
    if (!q16 || !scores_dev || !rows_dev) return fail(FRHOT_EINVAL, "null search arguments");
    hipStream_t s = c->stream;
    if (g->nrows == 0) return search_fill_empty(s, Bq, k, scores_dev, rows_dev);
    const size_t need = scan_part_entries(Bq, k);
    if (need > g->part_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->part_cap = 0;
        HIP_TRY(free_null(g->part_s));
        HIP_TRY(free_null(g->part_r));
        HIP_TRY(hipMalloc((void**)&g->part_s, need * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&g->part_r, need * sizeof(int64_t)));
        g->part_cap = need;
    }
    if (scan_bound_bytes(Bq) > g->qbound_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->qbound_cap = 0;
        HIP_TRY(free_null(g->qbound));
        const size_t cap = scan_bound_bytes(Bq + Bq / 2);
        HIP_TRY(hipMalloc((void**)&g->qbound, cap));
        HIP_TRY(hipMemset(g->qbound, 0, cap));         // slots start out empty; no launch clears them again
        g->qbound_cap = cap;
    }
    return search_launch(g, q16, qkeys_dev, Bq, k, scores_dev, rows_dev, row_base, g->part_s, g->part_r, g->qbound);
}

// ---- by-key search ---------------------------------------------------------------------------
void key_tiles_drop(frhot_gallery* g) {
    std::unordered_map<int32_t, std::vector<int32_t>>().swap(g->key_tiles);
    g->key_tiles_entries = 0;
    g->key_tiles_off = true;
}

void key_tiles_add(frhot_gallery* g, const int64_t* rows, const int32_t* keys, int n) {
    if (g->key_tiles_off || !g->key_tiles_valid) return;      // (invalid: rebuilt from device memory when next needed)
    for (int i = 0; i < n; ++i) {
        if (keys[i] < 0) continue;
        const int32_t t = (int32_t)(rows[i] >> 8);
        std::vector<int32_t>& v = g->key_tiles[keys[i]];
        if (!v.empty() && v.back() == t) continue;
        const auto it = std::lower_bound(v.begin(), v.end(), t);
        if (it == v.end() || *it != t) {
            v.insert(it, t);
            if (++g->key_tiles_entries > g->key_tiles_limit()) { key_tiles_drop(g); return; }
        }
    }
}

// Rebuild from the keys in device memory (after a fill that generated them there).
int key_tiles_rebuild(frhot_gallery* g) {
    g->key_tiles.clear();
    g->key_tiles_entries = 0;
    if (g->nrows > 0) {
        std::vector<int32_t> keys((size_t)g->nrows);
        HIP_TRY(hipStreamSynchronize(g->ctx->stream));
        HIP_TRY(hipMemcpy(keys.data(), g->keys, keys.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < g->nrows; ++r) {
            if (keys[(size_t)r] < 0) continue;
            std::vector<int32_t>& v = g->key_tiles[keys[(size_t)r]];
            const int32_t t = (int32_t)(r >> 8);
            if (v.empty() || v.back() != t) {                    // rows ascend, so do the tiles
                v.push_back(t);
                if (++g->key_tiles_entries > g->key_tiles_limit()) { key_tiles_drop(g); break; }
            }
        }
    }
    g->key_tiles_valid = true;
    return FRHOT_OK;
}

int index_pin_grow(IndexPin& ip, size_t n) {
    if (!ip.copied) HIP_TRY(hipEventCreateWithFlags(&ip.copied, hipEventDisableTiming));
    if (ip.pending) { ip.pending = false; HIP_TRY(hipEventSynchronize(ip.copied)); }   // an enqueue-only call's upload
    if (n <= ip.cap) return FRHOT_OK;
    if (ip.p) { int32_t* old = ip.p; ip.p = nullptr; ip.cap = 0; HIP_TRY(hipHostFree(old)); }
    const size_t cap = n + n / 2 + 1024;
    HIP_TRY(hipHostMalloc((void**)&ip.p, cap * 4, hipHostMallocDefault));
    ip.cap = cap;
    return FRHOT_OK;
}

// Search with the queries' keys known on the host.  When it pays (or the gallery is told to), the
// queries are brought into key order and every key's queries scan only the tiles that may hold
// the key: one launch whose blocks each take one key's queries over a share of that key's tiles, and
// one merge that writes the answers straight to the callers' order.
// Otherwise the keys go to `qkeys_dev_buf` and one scan covers the whole gallery.  Same answers
// either way: every row's key is still tested, and the listed tiles hold every row of the key.
// `pin` must stay untouched until the stream has passed this call's copy (the caller's slot or
// lock guarantees it).  Enqueues only; entered with ctx->mu held.
int search_keyed_enqueue(frhot_gallery* g, const void* q16, const int32_t* qkeys, int32_t* qkeys_dev_buf, int Bq, int k,
                         float* scores_dev, int64_t* rows_dev, IndexPin& pin, int64_t row_base = 0) {
    frhot_ctx* c = g->ctx;
    hipStream_t s = c->stream;
    auto whole = [&]() -> int {
        if (qkeys) HIP_TRY(hipMemcpyAsync(qkeys_dev_buf, qkeys, (size_t)Bq * 4, hipMemcpyHostToDevice, s));
        return search_enqueue(g, q16, qkeys ? qkeys_dev_buf : nullptr, Bq, k, scores_dev, rows_dev, row_base);
    };
    if (!qkeys || g->search_mode == FRHOT_SEARCH_WHOLE || g->key_tiles_off || g->nrows == 0 || Bq <= 0 || k <= 0 ||
        k > 16 || !q16 || !scores_dev || !rows_dev)
        return whole();                              // (argument errors are reported by search_enqueue)
    if (!g->key_tiles_valid) {
        const int rc = key_tiles_rebuild(g);
        if (rc) return rc;
        if (g->key_tiles_off) return whole();
    }
    // queries in key order (stable: equal keys keep the callers' order)
    std::vector<int32_t> ord((size_t)Bq);
    for (int i = 0; i < Bq; ++i) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return qkeys[a] < qkeys[b]; });
    struct Group { int qbeg, qn; const std::vector<int32_t>* tiles; size_t toff; bool any; };
    std::vector<Group> groups;
    static const std::vector<int32_t> none;
    size_t ntl_total = 0;
    for (int i = 0; i < Bq;) {
        const int32_t key = qkeys[ord[(size_t)i]];
        int j = i;
        while (j < Bq && qkeys[ord[(size_t)j]] == key) ++j;
        Group gr{i, j - i, &none, ntl_total, key == FRHOT_KEY_ANY};
        if (!gr.any) {
            const auto it = g->key_tiles.find(key);
            if (it != g->key_tiles.end()) gr.tiles = &it->second;
            ntl_total += gr.tiles->size();
        }
        groups.push_back(gr);
        i = j;
    }
    // Work items: every key's queries in 64-query tiles, each tile's scan spread over nb walkers.
    const int64_t all_tiles = (g->nrows + 255) / 256;
    int qt_total = 0;
    for (const Group& gr : groups)
        if (gr.any || !gr.tiles->empty()) qt_total += (gr.qn + 63) / 64;
    const int budget = qt_total > 0 ? std::max(1, std::min(256, 512 / qt_total)) : 1;
    std::vector<ScanWork> work;
    std::vector<int32_t> nlists((size_t)Bq, 0);        // per query (in key order): lists to merge
    int max_nb = 1;
    double bytes = 0.0, longest = 0.0;                 // for the cost rule: bytes read, longest walk in tiles
    bool long_walks = false;
    for (const Group& gr : groups) {
        const int64_t ntl = gr.any ? all_tiles : (int64_t)gr.tiles->size();
        if (ntl == 0) continue;
        const int nb = (int)std::min<int64_t>(ntl, budget);
        if (nb > max_nb) max_nb = nb;
        if (ntl / nb >= 3) long_walks = true;          // (enough tiles for the bounds to come round)
        for (int i = 0; i < gr.qn; ++i) nlists[(size_t)(gr.qbeg + i)] = nb;
        for (int q0 = gr.qbeg; q0 < gr.qbeg + gr.qn; q0 += 64)
            for (int gb = 0; gb < nb; ++gb)
                work.push_back(ScanWork{q0, gr.qbeg + gr.qn, gr.any ? -1 : (int32_t)gr.toff, (int32_t)ntl, gb, nb});
        bytes += (double)((gr.qn + 63) / 64) * (double)ntl * 256.0 * g->D * 2.0;
        longest = std::max(longest, std::ceil((double)ntl / nb));
    }
    if (g->search_mode == FRHOT_SEARCH_AUTO) {
        // microseconds: a scan streams about 6.5 TB/s when it fills the chip and a lone block walks a
        // tile in D/64 steps of about 0.45 us; the whole-gallery scan is bound by bytes (or, past 256
        // queries, by the equal MFMA time of every further 256)
        const double per_us = 6.5e6;
        const double whole_us = 20.0 + (double)all_tiles * 256.0 * g->D * 2.0 * std::ceil(Bq / 256.0) / per_us;
        const double by_key_us = 30.0 + std::max(bytes / per_us, longest * (g->D / 64) * 0.45);
        if (work.size() > 65535 || by_key_us > 0.9 * whole_us) return whole();
    }
    if (work.empty()) return search_fill_empty(s, Bq, k, scores_dev, rows_dev);     // no key has a row
    // index arrays: [order | keys in that order | lists per query | tile lists | work items]
    const size_t o_keys = (size_t)Bq, o_nl = 2 * (size_t)Bq, o_tiles = 3 * (size_t)Bq, o_work = o_tiles + ntl_total;
    const size_t nidx = o_work + work.size() * (sizeof(ScanWork) / 4);
    int rc = index_pin_grow(pin, nidx);
    if (rc) return rc;
    if (nidx > g->idx_dev_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->idx_dev_cap = 0;
        HIP_TRY(free_null(g->idx_dev));
        const size_t cap = nidx + nidx / 2 + 1024;
        HIP_TRY(hipMalloc((void**)&g->idx_dev, cap * 4));
        g->idx_dev_cap = cap;
    }
    const size_t qbytes = (size_t)Bq * g->D * 2;
    if (qbytes > g->q16s_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->q16s_cap = 0;
        HIP_TRY(free_null(g->q16s));
        const size_t cap = qbytes + qbytes / 2;
        HIP_TRY(hipMalloc(&g->q16s, cap));
        g->q16s_cap = cap;
    }
    const int kl = scan_list_len(k);
    const size_t need = (size_t)max_nb * (size_t)Bq * (size_t)kl;
    if (need > g->part_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->part_cap = 0;
        HIP_TRY(free_null(g->part_s));
        HIP_TRY(free_null(g->part_r));
        HIP_TRY(hipMalloc((void**)&g->part_s, need * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&g->part_r, need * sizeof(int64_t)));
        g->part_cap = need;
    }
    // (a key's query tiles start at its first query: the last one's columns run up to 63 past Bq)
    if (long_walks && scan_bound_bytes(Bq + 64) > g->qbound_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->qbound_cap = 0;
        HIP_TRY(free_null(g->qbound));
        const size_t cap = scan_bound_bytes(Bq + Bq / 2 + 64);
        HIP_TRY(hipMalloc((void**)&g->qbound, cap));
        HIP_TRY(hipMemset(g->qbound, 0, cap));
        g->qbound_cap = cap;
    }
    int32_t* const h = pin.p;
    for (int i = 0; i < Bq; ++i) {
        h[i] = ord[(size_t)i];
        h[o_keys + i] = qkeys[ord[(size_t)i]];
        h[o_nl + i] = nlists[(size_t)i];
    }
    for (const Group& gr : groups)
        if (!gr.tiles->empty()) memcpy(h + o_tiles + gr.toff, gr.tiles->data(), gr.tiles->size() * 4);
    memcpy(h + o_work, work.data(), work.size() * sizeof(ScanWork));
    HIP_TRY(hipMemcpyAsync(g->idx_dev, h, nidx * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(pin.copied, s));
    pin.pending = true;
    const int32_t* const ord_dev = g->idx_dev;
    HIP_TRY(launch_rows_gather_f16(q16, ord_dev, Bq, g->D, g->q16s, s));
    if (c->stage_begin(3)) return fail(FRHOT_EHIP, "event record failed");
    HIP_TRY(launch_scan_topk_work(g->rows, g->keys, g->nrows, g->D, g->q16s, g->idx_dev + o_keys, Bq, k, g->part_s,
                                  g->part_r, long_walks ? g->qbound : nullptr, g->idx_dev + o_tiles,
                                  (const ScanWork*)(g->idx_dev + o_work), (int)work.size(), s));
    if (c->stage_end(3) || c->stage_begin(4)) return fail(FRHOT_EHIP, "event record failed");
    // every query merges its key's walkers' lists (none: the empty answer) and writes to its caller's row
    HIP_TRY(launch_topk_final(g->part_s, g->part_r, max_nb, kl, Bq, k, row_base, scores_dev, rows_dev, s, ord_dev,
                              g->idx_dev + o_nl));
    if (c->stage_end(4)) return fail(FRHOT_EHIP, "event record failed");
    return FRHOT_OK;
}

// Staging for n rows (and, with `with_rows`, n fp16 rows).  Buffers above kMutKeep rows are
// released again by gallery_trim_mutation_staging after the call.
constexpr size_t kMutKeep = 4096;
int gallery_grow_mutation_staging(frhot_gallery* g, size_t n, bool with_rows) {
    hipStream_t s = g->ctx->stream;
    if (n > g->mut_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->mut_cap = 0;
        HIP_TRY(free_null(g->mut_rows));
        HIP_TRY(free_null(g->mut_keys));
        const size_t cap = n < kMutKeep ? (n + n / 2 + 16 < kMutKeep ? n + n / 2 + 16 : kMutKeep) : n;
        HIP_TRY(hipMalloc((void**)&g->mut_rows, cap * 8));
        HIP_TRY(hipMalloc((void**)&g->mut_keys, cap * 4));
        g->mut_cap = cap;
    }
    if (with_rows && n > g->mut_f16_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        g->mut_f16_cap = 0;
        HIP_TRY(free_null(g->mut_f16));
        const size_t cap = n < kMutKeep ? (n + n / 2 + 16 < kMutKeep ? n + n / 2 + 16 : kMutKeep) : n;
        HIP_TRY(hipMalloc(&g->mut_f16, cap * (size_t)g->D * 2));
        g->mut_f16_cap = cap;
    }
    return FRHOT_OK;
}

void gallery_trim_mutation_staging(frhot_gallery* g) {
    if (g->mut_cap > kMutKeep) { g->mut_cap = 0; (void)free_null(g->mut_rows); (void)free_null(g->mut_keys); }
    if (g->mut_f16_cap > kMutKeep) { g->mut_f16_cap = 0; (void)free_null(g->mut_f16); }
}

int gallery_grow_query_staging(frhot_gallery* g, int Bq) {
    if (Bq <= g->q_cap) return FRHOT_OK;
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    void** ptrs[] = {(void**)&g->q16, (void**)&g->q32, (void**)&g->qk, (void**)&g->out_s, (void**)&g->out_r};
    for (void** p : ptrs) { if (*p) HIP_TRY(hipFree(*p)); *p = nullptr; }
    g->q_cap = 0;
    const int cap = Bq + Bq / 2 + 16;
    HIP_TRY(hipMalloc(&g->q16, (size_t)cap * g->D * 2));
    HIP_TRY(hipMalloc((void**)&g->q32, (size_t)cap * g->D * 4));
    HIP_TRY(hipMalloc((void**)&g->qk, (size_t)cap * 4));
    HIP_TRY(hipMalloc((void**)&g->out_s, (size_t)cap * 16 * 4));
    HIP_TRY(hipMalloc((void**)&g->out_r, (size_t)cap * 16 * 8));
    g->q_cap = cap;
    return FRHOT_OK;
}

}  // namespace

extern "C" {

const char* frhot_last_error(void) { return g_err.c_str(); }

int frhot_ctx_destroy(frhot_ctx* c);

const char* frhot_version(void) { return "frhot 0.1 gfx950 mfma-f16 igemm/scan"; }

static int ctx_init(frhot_ctx* c) {
    if (!c->stream) { HIP_TRY(hipStreamCreate(&c->stream)); c->own_stream = true; }
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    const int B = c->max_batch;
    for (auto& sl : c->slot) {
        HIP_TRY(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.uploaded0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void**)&sl.res_pin, (size_t)B * 16 * (4 + 8), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&sl.sharp_pin, (size_t)B * 16, hipHostMallocDefault));
    }
    const size_t x0_bytes = (size_t)B * 226 * 226 * 4 * 2;
    HIP_TRY(hipMalloc(&c->x0, x0_bytes));
    HIP_TRY(hipMemsetAsync(c->x0, 0, x0_bytes, c->stream));
#ifdef FR_NO_FUSE1
    for (int l = 0; l < kNumConv; ++l) {
#else
    for (int l = 1; l < kNumConv; ++l) {     // act[0] (6.4 MB per face) exists only once frhot_debug_read asks for it
#endif
        const size_t n = act_elems(kConv[l], B) * 2;
        HIP_TRY(hipMalloc(&c->act[l], n));
        HIP_TRY(hipMemsetAsync(c->act[l], 0, n, c->stream));
    }
    HIP_TRY(hipMalloc((void**)&c->u8, (size_t)B * 224 * 224 * 3));
    HIP_TRY(hipMalloc((void**)&c->img_max, sizeof(int) * B));
    HIP_TRY(hipMalloc((void**)&c->meta_dev, sizeof(CropMeta) * B));
    HIP_TRY(hipMalloc(&c->fc6_out, (size_t)B * kFcN * 2));
    // fp32 split-K slabs of fc6, then of fc7 (also the conv kernels' 1 KiB dump)
    HIP_TRY(hipMalloc((void**)&c->partial, (size_t)(kFc6Split > kFc7Split ? kFc6Split : kFc7Split) * B * kFcN * 4));
    HIP_TRY(hipMalloc((void**)&c->emb_dev, (size_t)B * kFcN * 4));
    HIP_TRY(hipMalloc(&c->q_dev, (size_t)B * kFcN * 2));
    HIP_TRY(hipMalloc((void**)&c->qkeys_dev, (size_t)B * 4));
    HIP_TRY(hipMalloc((void**)&c->res_s, (size_t)B * 16 * 4));
    HIP_TRY(hipMalloc((void**)&c->res_r, (size_t)B * 16 * 8));
    HIP_TRY(hipMalloc((void**)&c->sharp_dev, (size_t)B * 16));
    uint8_t lut[512];
    for (int i = 0; i < 256; ++i) {
        lut[i] = (uint8_t)i;
        // app/src/utils/legacy.py:108-109 with gamma 1.5: float64 pow, truncating cast
        lut[256 + i] = (uint8_t)(std::pow((double)i / 255.0, 1.0 / 1.5) * 255.0);
    }
    HIP_TRY(hipMalloc((void**)&c->lut_dev, 512));
    HIP_TRY(hipMemcpy(c->lut_dev, lut, 512, hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FRHOT_OK;
}

int frhot_ctx_create(int device_id, void* stream, int max_batch, frhot_ctx** out) {
    return guarded([&]() -> int {
    if (!out) return fail(FRHOT_EINVAL, "out is null");
    *out = nullptr;
    // 2048: the conv kernels address a layer's input with 32-bit byte offsets (conv3_2: 1.72 MB per face)
    if (max_batch <= 0 || max_batch > 2048) return fail(FRHOT_EINVAL, "max_batch %d outside 1..2048", max_batch);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(FRHOT_EINVAL, "device %d of %d", device_id, ndev);
    DeviceGuard dg(device_id);
    if (!dg.ok) return fail(FRHOT_EHIP, "cannot select device %d", device_id);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FRHOT_EHIP, "device %d is %s; this library holds gfx950 code only", device_id, prop.gcnArchName);
    HIP_TRY(kernels_init());
    frhot_ctx* c = new frhot_ctx();
    c->device = device_id;
    c->max_batch = max_batch;
    c->stream = (hipStream_t)stream;
    const int rc = ctx_init(c);
    if (rc) {
        const std::string keep = g_err;
        frhot_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return FRHOT_OK;
    });
}

int frhot_ctx_destroy(frhot_ctx* c) {
    return guarded([&]() -> int {
    if (!c) return FRHOT_OK;
    DeviceGuard dg(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& sl : c->slot) {
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.res_pin) (void)hipHostFree(sl.res_pin);
        if (sl.sharp_pin) (void)hipHostFree(sl.sharp_pin);
        if (sl.idx.p) (void)hipHostFree(sl.idx.p);
        if (sl.idx.copied) (void)hipEventDestroy(sl.idx.copied);
        if (sl.emb_pin) (void)hipHostFree(sl.emb_pin);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.uploaded0) (void)hipEventDestroy(sl.uploaded0);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->fork) (void)hipEventDestroy(c->fork);
    dist_release(c);
    void* ptrs[] = {c->x0, c->u8, c->img_max, c->meta_dev, c->lut_dev, c->fc6_out, c->partial,
                    c->emb_dev, c->q_dev, c->qkeys_dev, c->res_s, c->res_r, c->sharp_dev, c->w6, c->w7, c->b6, c->b7};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int l = 0; l < kNumConv; ++l) {
        if (c->act[l]) (void)hipFree(c->act[l]);
        if (c->w[l]) (void)hipFree(c->w[l]);
        if (c->bias[l]) (void)hipFree(c->bias[l]);
    }
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < frhot_ctx::kRing; ++j) {
            if (c->ev_b[i][j]) (void)hipEventDestroy(c->ev_b[i][j]);
            if (c->ev_e[i][j]) (void)hipEventDestroy(c->ev_e[i][j]);
        }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FRHOT_OK;
    });
}

int frhot_ctx_sync(frhot_ctx* c) {
    return guarded([&]() -> int {
    if (!c) return fail(FRHOT_EINVAL, "ctx is null");
    DeviceGuard dg(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FRHOT_OK;
    });
}

int frhot_model_load(frhot_ctx* c, const void* blob_v, size_t nbytes, float input_scale) {
    return guarded([&]() -> int {
    if (!c || !blob_v) return fail(FRHOT_EINVAL, "null argument");
    if (!(input_scale > 0.f) || !std::isfinite(input_scale)) return fail(FRHOT_EINVAL, "input_scale must be positive");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    const uint8_t* blob = (const uint8_t*)blob_v;
    if (c->model) return fail(FRHOT_ESTATE, "a model is already loaded in this context");
    auto load_bias = [&](const char* lname, int n, float** dst) -> int {
        char nm[32];
        snprintf(nm, sizeof nm, "%s.b", lname);
        const BlobEntry* e = blob_find(blob, nbytes, nm);
        if (!e || e->dtype != 1 || e->nbytes != (uint64_t)n * 4) return fail(FRHOT_EINVAL, "blob: bad or missing %s", nm);
        std::vector<float> b(n);
        memcpy(b.data(), blob + e->offset, (size_t)n * 4);
        for (float& v : b) v *= input_scale;
        return upload((void**)dst, b.data(), (size_t)n * 4);
    };
    auto find_w = [&](const char* lname, uint64_t elems) -> const BlobEntry* {
        char nm[32];
        snprintf(nm, sizeof nm, "%s.w", lname);
        const BlobEntry* e = blob_find(blob, nbytes, nm);
        if (!e || e->dtype != 0 || e->nbytes != elems * 2) { fail(FRHOT_EINVAL, "blob: bad or missing %s", nm); return nullptr; }
        return e;
    };
    for (int l = 0; l < kNumConv; ++l) {
        const LayerSpec& L = kConv[l];
        const BlobEntry* e = find_w(L.name, (uint64_t)L.cout * 9 * L.cin);
        if (!e) return FRHOT_EINVAL;
        int rc;
        if (l == 0) {
            // [64][3][3][3] -> [64][64], k = tap*4 + c, zero elsewhere
            std::vector<uint16_t> p(64 * 64, 0);
            const uint16_t* src = (const uint16_t*)(blob + e->offset);
            for (int n = 0; n < 64; ++n)
                for (int t = 0; t < 9; ++t)
                    for (int ch = 0; ch < 3; ++ch) p[n * 64 + t * 4 + ch] = src[(n * 9 + t) * 3 + ch];
            rc = upload(&c->w[0], p.data(), p.size() * 2);
        } else {
            rc = upload(&c->w[l], blob + e->offset, e->nbytes);
        }
        if (rc) return rc;
        rc = load_bias(L.name, L.cout, &c->bias[l]);
        if (rc) return rc;
    }
    const BlobEntry* e6 = find_w("fc6", (uint64_t)kFcN * kFc6K);
    if (!e6) return FRHOT_EINVAL;
    int rc = upload(&c->w6, blob + e6->offset, e6->nbytes);
    if (rc) return rc;
    const BlobEntry* e7 = find_w("fc7", (uint64_t)kFcN * kFcN);
    if (!e7) return FRHOT_EINVAL;
    rc = upload(&c->w7, blob + e7->offset, e7->nbytes);
    if (rc) return rc;
    rc = load_bias("fc6", kFcN, &c->b6);
    if (rc) return rc;
    rc = load_bias("fc7", kFcN, &c->b7);
    if (rc) return rc;
    c->input_scale = input_scale;
    c->model = true;
    return FRHOT_OK;
    });
}

int frhot_embed_dev(frhot_ctx* c, const void* crops_dev, const int64_t* offsets, const int32_t* hw,
                    const int32_t* strides, int B, unsigned flags, void* out_f32_dev, void* out_q_f16_dev, int Dq) {
    return guarded([&]() -> int {
    if (!c) return fail(FRHOT_EINVAL, "ctx is null");
    if (!out_f32_dev && !out_q_f16_dev) return fail(FRHOT_EINVAL, "no output requested");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    return embed_enqueue(c, crops_dev, offsets, hw, strides, B, flags, (float*)out_f32_dev, out_q_f16_dev, Dq);
    });
}

int frhot_embed(frhot_ctx* c, const uint8_t* const* crops, const int32_t* hw, const int32_t* strides, int B,
                unsigned flags, float* out) {
    return guarded([&]() -> int {
    if (!c || !out) return fail(FRHOT_EINVAL, "null argument");
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    DeviceGuard dg(c->device);
    HostCall call(c);
    SlotLease lease(c);
    int rc = slot_emb_mirror(c, lease.sl);
    if (rc) return rc;
    CropStaging st(c, lease.sl);
    rc = st.begin(crops, hw, strides, B, host_first_part(B, call.alone));
    if (rc) return rc;
    rc = run_staged(c, lease.sl, st, [&]() -> int {
        int r = embed_enqueue(c, lease.sl.dev, st.offsets.data(), hw, st.packed_strides.data(), B, flags, c->emb_dev,
                              nullptr, kFcN, &st, st.split_at, !call.alone);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(lease.sl.emb_pin, c->emb_dev, (size_t)B * kFcN * 4, hipMemcpyDeviceToHost, c->stream));
        return FRHOT_OK;
    });
    if (rc) return rc;
    memcpy(out, lease.sl.emb_pin, (size_t)B * kFcN * 4);
    return FRHOT_OK;
    });
}

int frhot_crop_sharpness_dev(frhot_ctx* c, const void* crops_dev, const int64_t* offsets, const int32_t* hw,
                             const int32_t* strides, int B, void* out_dev) {
    return guarded([&]() -> int {
    if (!c) return fail(FRHOT_EINVAL, "ctx is null");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    return sharpness_enqueue(c, crops_dev, offsets, hw, strides, B, (int64_t*)out_dev, false);
    });
}

int frhot_crop_sharpness(frhot_ctx* c, const uint8_t* const* crops, const int32_t* hw, const int32_t* strides, int B,
                         int64_t* out) {
    return guarded([&]() -> int {
    if (!c || !out) return fail(FRHOT_EINVAL, "null argument");
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    DeviceGuard dg(c->device);
    SlotLease lease(c);
    CropStaging st(c, lease.sl);
    int rc = st.begin(crops, hw, strides, B, 0);
    if (rc) return rc;
    rc = run_staged(c, lease.sl, st, [&]() -> int {
        int r = sharpness_enqueue(c, lease.sl.dev, st.offsets.data(), hw, st.packed_strides.data(), B, c->sharp_dev, false);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(lease.sl.sharp_pin, c->sharp_dev, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
        return FRHOT_OK;
    });
    if (rc) return rc;
    memcpy(out, lease.sl.sharp_pin, (size_t)B * 16);
    return FRHOT_OK;
    });
}

int frhot_debug_read(frhot_ctx* c, int stage, int B, void* host_out, size_t nbytes) {
    return guarded([&]() -> int {
    if (!c || !host_out) return fail(FRHOT_EINVAL, "null argument");
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    const void* src; size_t need;
    if (stage == 0) { src = c->u8; need = (size_t)B * 224 * 224 * 3; }
    else if (stage == 1) { src = c->x0; need = (size_t)B * 226 * 226 * 4 * 2; }
    else if (stage >= 10 && stage < 10 + kNumConv) { src = c->act[stage - 10]; need = act_elems(kConv[stage - 10], B) * 2; }
    else if (stage == 30) { src = c->fc6_out; need = (size_t)B * kFcN * 2; }
    else return fail(FRHOT_EINVAL, "unknown stage %d", stage);
    if (nbytes != need) return fail(FRHOT_EINVAL, "stage %d of %d images is %zu bytes, not %zu", stage, B, need, nbytes);
    if (stage == 10) {
        if (!c->model) return fail(FRHOT_ESTATE, "frhot_model_load has not been called");
        // The embed path never stores the first layer's output (it is computed into the second layer's
        // LDS patches); the stand-alone kernel makes it here from the network input of the last batch.
        if (!c->act[0]) {
            const size_t n = act_elems(kConv[0], c->max_batch) * 2;
            HIP_TRY(hipMalloc(&c->act[0], n));
            HIP_TRY(hipMemsetAsync(c->act[0], 0, n, c->stream));
        }
        HIP_TRY(launch_conv1_1(c->x0, c->w[0], c->bias[0], c->act[0], B, c->stream));
        src = c->act[0];
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host_out, src, need, hipMemcpyDeviceToHost));
    return FRHOT_OK;
    });
}

int frhot_gallery_create(frhot_ctx* c, int D, int64_t capacity, frhot_gallery** out) {
    return guarded([&]() -> int {
    if (!c || !out) return fail(FRHOT_EINVAL, "null argument");
    *out = nullptr;
    if (D < 64 || D % 64 || D > 8192) return fail(FRHOT_EINVAL, "D %d must be a multiple of 64 in 64..8192", D);
    if (capacity <= 0 || capacity > 0x7fffff00LL) return fail(FRHOT_EINVAL, "capacity out of range");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    frhot_gallery* g = new frhot_gallery();
    g->ctx = c; g->D = D; g->capacity = capacity;
    const size_t padded = ((size_t)capacity + 255) & ~(size_t)255;   // the scan reads whole 256-row tiles
    hipError_t e = hipMalloc(&g->rows, padded * D * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&g->keys, padded * 4);
    if (e == hipSuccess) e = hipMemsetAsync(g->rows, 0, padded * D * 2, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->keys, 0xFF, padded * 4, c->stream);  // all dead
    if (e != hipSuccess) {
        if (g->rows) (void)hipFree(g->rows);
        if (g->keys) (void)hipFree(g->keys);
        delete g;
        return fail(FRHOT_EHIP, "gallery allocation of %lld x %d failed: %s", (long long)capacity, D, hipGetErrorString(e));
    }
    *out = g;
    return FRHOT_OK;
    });
}

int frhot_gallery_destroy(frhot_gallery* g) {
    return guarded([&]() -> int {
    if (!g) return FRHOT_OK;
    std::lock_guard<std::mutex> lk(g->ctx->mu);
    DeviceGuard dg(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    void* ptrs[] = {g->rows, g->keys, g->part_s, g->part_r, g->qbound, g->up_vecs, g->up_keys, g->up_rows,
                    g->q16, g->q32, g->qk, g->out_s, g->out_r, g->mut_rows, g->mut_keys, g->mut_f16, g->idx_dev, g->q16s};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (g->idx.p) (void)hipHostFree(g->idx.p);
    if (g->io_pin) (void)hipHostFree(g->io_pin);
    if (g->idx.copied) (void)hipEventDestroy(g->idx.copied);
    delete g;
    return FRHOT_OK;
    });
}

int64_t frhot_gallery_rows(frhot_gallery* g) { return g ? g->nrows : 0; }

int frhot_gallery_search_mode(frhot_gallery* g, int mode) {
    return guarded([&]() -> int {
    if (!g) return fail(FRHOT_EINVAL, "gallery is null");
    if (mode != FRHOT_SEARCH_AUTO && mode != FRHOT_SEARCH_BY_KEY && mode != FRHOT_SEARCH_WHOLE)
        return fail(FRHOT_EINVAL, "search mode %d is not one of FRHOT_SEARCH_AUTO / _BY_KEY / _WHOLE", mode);
    std::lock_guard<std::mutex> lk(g->ctx->mu);
    g->search_mode = mode;
    return FRHOT_OK;
    });
}

int frhot_gallery_upsert(frhot_gallery* g, const float* vecs, const int32_t* keys, const int64_t* rows, int n) {
    return guarded([&]() -> int {
    if (!g || !vecs || !keys || !rows) return fail(FRHOT_EINVAL, "null argument");
    if (n <= 0) return fail(FRHOT_EINVAL, "n must be positive");
    int64_t hi = -1;
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= g->capacity) return fail(FRHOT_ECAPACITY, "row %lld outside capacity %lld", (long long)rows[i], (long long)g->capacity);
        if (keys[i] < 0) return fail(FRHOT_EINVAL, "key %d must be non-negative", keys[i]);
        if (rows[i] > hi) hi = rows[i];
    }
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    if (n > g->up_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        g->up_cap = 0;
        HIP_TRY(free_null(g->up_vecs));
        HIP_TRY(free_null(g->up_keys));
        HIP_TRY(free_null(g->up_rows));
        const int cap = n + n / 2 + 16;
        HIP_TRY(hipMalloc((void**)&g->up_vecs, (size_t)cap * g->D * 4));
        HIP_TRY(hipMalloc((void**)&g->up_keys, (size_t)cap * 4));
        HIP_TRY(hipMalloc((void**)&g->up_rows, (size_t)cap * 8));
        g->up_cap = cap;
    }
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(g->up_vecs, vecs, (size_t)n * g->D * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->up_keys, keys, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->up_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_gallery_upsert(g->up_vecs, g->up_keys, g->up_rows, n, g->rows, g->keys, g->D, s));
    HIP_TRY(hipStreamSynchronize(s));
    // rows stay dense when the batch only overwrites live rows or appends one contiguous run
    bool contiguous = true, inside = true;
    for (int i = 0; i < n; ++i) {
        if (rows[i] != rows[0] + i) contiguous = false;
        if (rows[i] >= g->nrows) inside = false;
    }
    if (!(inside || (contiguous && rows[0] <= g->nrows))) g->dense = false;
    if (hi + 1 > g->nrows) g->nrows = hi + 1;
    key_tiles_add(g, rows, keys, n);
    return FRHOT_OK;
    });
}

int frhot_gallery_tombstone(frhot_gallery* g, const int64_t* rows, int n) {
    return guarded([&]() -> int {
    if (!g || !rows) return fail(FRHOT_EINVAL, "null argument");
    if (n <= 0) return fail(FRHOT_EINVAL, "n must be positive");
    for (int i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= g->capacity) return fail(FRHOT_ECAPACITY, "row %lld outside capacity", (long long)rows[i]);
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    int rc = gallery_grow_mutation_staging(g, (size_t)n, false);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(g->mut_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_gallery_set_keys(g->mut_rows, n, FRHOT_KEY_DEAD, g->keys, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    gallery_trim_mutation_staging(g);
    if (e != hipSuccess) return fail(FRHOT_EHIP, "tombstone failed: %s", hipGetErrorString(e));
    g->dense = false;
    return FRHOT_OK;
    });
}

int frhot_gallery_fill_synthetic(frhot_gallery* g, uint64_t seed, int64_t row0, int64_t n, int nkeys, int dup_every) {
    return guarded([&]() -> int {
    if (!g) return fail(FRHOT_EINVAL, "gallery is null");
    if (row0 < 0 || n <= 0 || row0 + n > g->capacity) return fail(FRHOT_ECAPACITY, "rows [%lld,%lld) outside capacity %lld", (long long)row0, (long long)(row0 + n), (long long)g->capacity);
    if (dup_every < 0 || dup_every == 1) return fail(FRHOT_EINVAL, "bad dup_every");
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(launch_gallery_fill(g->rows, g->keys, g->D, seed, row0, n, nkeys, dup_every, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (row0 > g->nrows) g->dense = false;
    if (row0 + n > g->nrows) g->nrows = row0 + n;
    g->key_tiles_valid = false;         // the kernel derived the keys: a by-key search reads them back first
    return FRHOT_OK;
    });
}

int frhot_gallery_read(frhot_gallery* g, const int64_t* rows, int n, void* out_f16, int32_t* out_keys) {
    return guarded([&]() -> int {
    if (!g || !rows || !out_f16) return fail(FRHOT_EINVAL, "null argument");
    if (n <= 0) return fail(FRHOT_EINVAL, "n must be positive");
    for (int i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= g->capacity) return fail(FRHOT_ECAPACITY, "row %lld outside capacity", (long long)rows[i]);
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    int rc = gallery_grow_mutation_staging(g, (size_t)n, true);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(g->mut_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_gallery_gather(g->rows, g->keys, g->mut_rows, n, g->D, g->mut_f16, g->mut_keys, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_f16, g->mut_f16, (size_t)n * g->D * 2, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && out_keys) e = hipMemcpyAsync(out_keys, g->mut_keys, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    gallery_trim_mutation_staging(g);
    if (e != hipSuccess) return fail(FRHOT_EHIP, "gallery read failed: %s", hipGetErrorString(e));
    return FRHOT_OK;
    });
}

int frhot_gallery_write_f16(frhot_gallery* g, const void* rows_f16, const int32_t* keys, const int64_t* rows, int n) {
    return guarded([&]() -> int {
    if (!g || !rows_f16 || !keys || !rows) return fail(FRHOT_EINVAL, "null argument");
    if (n <= 0) return fail(FRHOT_EINVAL, "n must be positive");
    int64_t hi = -1;
    bool contiguous = true, inside = true;
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= g->capacity) return fail(FRHOT_ECAPACITY, "row %lld outside capacity %lld", (long long)rows[i], (long long)g->capacity);
        if (keys[i] < FRHOT_KEY_DEAD) return fail(FRHOT_EINVAL, "key %d is not a row key", keys[i]);
        if (rows[i] > hi) hi = rows[i];
        if (rows[i] != rows[0] + i) contiguous = false;
        if (rows[i] >= g->nrows) inside = false;
    }
    bool any_dead = false;
    for (int i = 0; i < n; ++i) any_dead = any_dead || keys[i] == FRHOT_KEY_DEAD;
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    int rc = gallery_grow_mutation_staging(g, (size_t)n, true);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(g->mut_f16, rows_f16, (size_t)n * g->D * 2, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->mut_keys, keys, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->mut_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_gallery_scatter(g->mut_f16, g->mut_keys, g->mut_rows, n, g->D, g->rows, g->keys, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    gallery_trim_mutation_staging(g);
    if (e != hipSuccess) return fail(FRHOT_EHIP, "gallery write failed: %s", hipGetErrorString(e));
    if (any_dead || !(inside || (contiguous && rows[0] <= g->nrows))) g->dense = false;
    if (hi + 1 > g->nrows) g->nrows = hi + 1;
    key_tiles_add(g, rows, keys, n);
    return FRHOT_OK;
    });
}

int frhot_search_dev(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_dev, int Bq, int k,
                     void* scores_dev, void* rows_dev, int64_t row_base) {
    return guarded([&]() -> int {
    if (!g) return fail(FRHOT_EINVAL, "gallery is null");
    std::lock_guard<std::mutex> lk(g->ctx->mu);
    DeviceGuard dg(g->ctx->device);
    return search_enqueue(g, q_f16_dev, qkeys_dev, Bq, k, (float*)scores_dev, (int64_t*)rows_dev, row_base);
    });
}

int frhot_search_dev_keyed(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_host, int Bq, int k,
                           void* scores_dev, void* rows_dev, int64_t row_base) {
    return guarded([&]() -> int {
    if (!g) return fail(FRHOT_EINVAL, "gallery is null");
    if (Bq <= 0 || k <= 0 || k > 16) return fail(FRHOT_EINVAL, "Bq %d / k %d out of range (k <= 16)", Bq, k);
    if (qkeys_host)
        for (int i = 0; i < Bq; ++i)
            if (qkeys_host[i] < 0 && qkeys_host[i] != FRHOT_KEY_ANY)
                return fail(FRHOT_EINVAL, "query key %d of query %d is neither a row key nor FRHOT_KEY_ANY", qkeys_host[i], i);
    std::lock_guard<std::mutex> lk(g->ctx->mu);
    DeviceGuard dg(g->ctx->device);
    if (qkeys_host) {
        const int rc = gallery_grow_query_staging(g, Bq);       // the device copy of the keys for a whole-gallery scan
        if (rc) return rc;
    }
    return search_keyed_enqueue(g, q_f16_dev, qkeys_host, g->qk, Bq, k, (float*)scores_dev, (int64_t*)rows_dev, g->idx,
                                row_base);
    });
}

int frhot_search_dev_static(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_dev, int Bq, int k,
                            void* scores_dev, void* rows_dev, int64_t row_base) {
    return guarded([&]() -> int {
    if (!g) return fail(FRHOT_EINVAL, "gallery is null");
    if (Bq <= 0 || k <= 0 || k > 16) return fail(FRHOT_EINVAL, "Bq %d / k %d out of range (k <= 16)", Bq, k);
    if (!q_f16_dev || !scores_dev || !rows_dev) return fail(FRHOT_EINVAL, "null search arguments");
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    ScanStaticScratch sc;
    HIP_TRY(scan_static_scratch(c->device, &sc));
    const size_t need = scan_part_entries(Bq, k);
    if (need > sc.entries || scan_bound_bytes(Bq) > sc.bound_bytes)
        return fail(FRHOT_ECAPACITY,
                    "%d queries of top-%d need %zu candidate entries and %zu bound bytes; the static scratch holds "
                    "%zu and %zu (8192 queries of top-5, 4096 of top-10, 2048 of top-16)",
                    Bq, k, need, scan_bound_bytes(Bq), sc.entries, sc.bound_bytes);
    if (g->nrows == 0) return search_fill_empty(c->stream, Bq, k, (float*)scores_dev, (int64_t*)rows_dev);
    // every context and stream on this GPU shares the one array: a second context gets it only once
    // the work of the one before has completed
    const hipError_t ec = scan_static_claim(c->device, c);
    if (ec == hipErrorNotReady)
        return fail(FRHOT_ESTATE, "the static scratch of GPU %d is in use by another context whose search has not "
                    "finished; wait for it (frhot_ctx_sync) or use frhot_search_dev", c->device);
    HIP_TRY(ec);
    const int rc = search_launch(g, q_f16_dev, qkeys_dev, Bq, k, (float*)scores_dev, (int64_t*)rows_dev, row_base,
                                 sc.part_s, sc.part_r, sc.qbound);
    HIP_TRY(scan_static_mark(c->device, c, c->stream));
    return rc;
    });
}

constexpr int kSmallSearch = 64;

int frhot_search(frhot_gallery* g, const float* q, const int32_t* qkeys, int Bq, int k, float* scores, int64_t* rows) {
    return guarded([&]() -> int {
    if (!g || !q || !scores || !rows) return fail(FRHOT_EINVAL, "null argument");
    if (Bq <= 0 || k <= 0 || k > 16) return fail(FRHOT_EINVAL, "Bq %d / k %d out of range (k <= 16)", Bq, k);
    frhot_ctx* c = g->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    if (qkeys)
        for (int i = 0; i < Bq; ++i)
            if (qkeys[i] < 0 && qkeys[i] != FRHOT_KEY_ANY)
                return fail(FRHOT_EINVAL, "query key %d of query %d is neither a row key nor FRHOT_KEY_ANY", qkeys[i], i);
    int rc = gallery_grow_query_staging(g, Bq);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const size_t qbytes = (size_t)Bq * g->D * 4, sbytes = (size_t)Bq * k * 4, rbytes = (size_t)Bq * k * 8;
    const bool small = Bq <= kSmallSearch;
    const size_t pin_q = (size_t)kSmallSearch * g->D * 4, pin_s = (size_t)kSmallSearch * 16 * 4;
    if (small && !g->io_pin)
        HIP_TRY(hipHostMalloc((void**)&g->io_pin, pin_q + pin_s + (size_t)kSmallSearch * 16 * 8, hipHostMallocDefault));
    if (small) memcpy(g->io_pin, q, qbytes);
    HIP_TRY(hipMemcpyAsync(g->q32, small ? (const void*)g->io_pin : (const void*)q, qbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_f32_to_unit_f16(g->q32, g->q16, Bq, g->D, s));
    rc = search_keyed_enqueue(g, g->q16, qkeys, g->qk, Bq, k, g->out_s, g->out_r, g->idx);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipMemcpyAsync(small ? (void*)(g->io_pin + pin_q) : (void*)scores, g->out_s, sbytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(small ? (void*)(g->io_pin + pin_q + pin_s) : (void*)rows, g->out_r, rbytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (small) {
        memcpy(scores, g->io_pin + pin_q, sbytes);
        memcpy(rows, g->io_pin + pin_q + pin_s, rbytes);
    }
    return FRHOT_OK;
    });
}

int frhot_merge_topk_dev(frhot_ctx* c, const void* scores_dev, const void* rows_dev, int P, int Bq, int k,
                         void* out_scores_dev, void* out_rows_dev) {
    return guarded([&]() -> int {
    if (!c || !scores_dev || !rows_dev || !out_scores_dev || !out_rows_dev) return fail(FRHOT_EINVAL, "null argument");
    if (P <= 0 || P > 256 || Bq <= 0 || k <= 0 || k > 16)
        return fail(FRHOT_EINVAL, "P %d / Bq %d / k %d out of range (P <= 256, k <= 16)", P, Bq, k);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(launch_topk_final((const float*)scores_dev, (const int64_t*)rows_dev, P, k, Bq, k, 0,
                              (float*)out_scores_dev, (int64_t*)out_rows_dev, c->stream));
    return FRHOT_OK;
    });
}

int frhot_merge_topk_packed_dev(frhot_ctx* c, const void* packed_dev, int64_t shard_stride_bytes, int P, int Bq, int k,
                                void* out_scores_dev, void* out_rows_dev) {
    return guarded([&]() -> int {
    if (!c || !packed_dev || !out_scores_dev || !out_rows_dev) return fail(FRHOT_EINVAL, "null argument");
    if (P <= 0 || P > 256 || Bq <= 0 || k <= 0 || k > 16)
        return fail(FRHOT_EINVAL, "P %d / Bq %d / k %d out of range (P <= 256, k <= 16)", P, Bq, k);
    const int64_t n = (int64_t)Bq * k;
    if (shard_stride_bytes < n * 12 || shard_stride_bytes % 8 || ((uintptr_t)packed_dev & 7u))
        return fail(FRHOT_EINVAL, "a shard's block holds %lld rows (int64) then %lld scores (float32): the stride must be "
                    "a multiple of 8 bytes and at least %lld, the buffer 8-byte aligned; got %lld",
                    (long long)n, (long long)n, (long long)(n * 12), (long long)shard_stride_bytes);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    const char* const b = (const char*)packed_dev;
    HIP_TRY(launch_topk_final((const float*)(b + n * 8), (const int64_t*)b, P, k, Bq, k, 0, (float*)out_scores_dev,
                              (int64_t*)out_rows_dev, c->stream, nullptr, nullptr, shard_stride_bytes / 4,
                              shard_stride_bytes / 8));
    return FRHOT_OK;
    });
}

// ---- the row-sharded search with the exchange inside the library -----------------------------
// RCCL is bound at run time: a process that never shards never loads it, and one whose host
// framework already holds a copy (torch ships its own) gets that copy, not a second one.
namespace {
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;
};

std::string g_rccl_why;     // why the last attempt to load it failed (guarded by the mutex in rccl())

Rccl* rccl() {
    static std::mutex m;
    static Rccl r;
    std::lock_guard<std::mutex> lk(m);
    if (r.so) return &r;
    const char* env = getenv("FRHOT_RCCL_LIB");
    const char* names[] = {env && *env ? env : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    r.why.clear();
    for (const char* n : names) {
        if (!n) continue;
        so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
        const char* e = dlerror();
        r.why += std::string(n) + ": " + (e ? e : "?") + "; ";
        if (n == env) break;                        // a library named by the caller is the only one tried
    }
    if (!so) { g_rccl_why = r.why; return nullptr; }
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(so, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(so, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(so, "ncclCommDestroy");
    r.all_gather = (decltype(r.all_gather))dlsym(so, "ncclAllGather");
    r.error_string = (decltype(r.error_string))dlsym(so, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.error_string) {
        r.why = "the library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
        g_rccl_why = r.why;
        dlclose(so);
        return nullptr;
    }
    r.so = so;
    return &r;
}

int grow(void** p, size_t* cap, size_t need_bytes) {
    if (*cap >= need_bytes) return FRHOT_OK;
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc(p, need_bytes));
    *cap = need_bytes;
    return FRHOT_OK;
}
}  // namespace

#define RCCL_TRY(r, expr)                                                                        \
    do {                                                                                         \
        const ncclResult_t e_ = (expr);                                                          \
        if (e_ != ncclSuccess) return fail(FRHOT_EHIP, "%s failed: %s", #expr, (r)->error_string(e_)); \
    } while (0)

static void dist_release(frhot_ctx* c) {
    if (c->comm) {
        if (Rccl* r = rccl()) (void)r->comm_destroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    c->dist_world = 0; c->dist_rank = 0;
    if (c->xq) (void)hipFree(c->xq);
    if (c->xk) (void)hipFree(c->xk);
    if (c->xblk) (void)hipFree(c->xblk);
    c->xq = nullptr; c->xk = nullptr; c->xblk = nullptr;
    c->xq_cap = c->xk_cap = c->xblk_cap = 0;
}

int frhot_dist_unique_id(void* id_out) {
    return guarded([&]() -> int {
    if (!id_out) return fail(FRHOT_EINVAL, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    Rccl* r = rccl();
    if (!r) return fail(FRHOT_ESTATE, "RCCL could not be loaded (%s)", g_rccl_why.c_str());
    ncclUniqueId id;
    RCCL_TRY(r, r->get_unique_id(&id));
    memcpy(id_out, &id, sizeof id);
    return FRHOT_OK;
    });
}

int frhot_dist_init(frhot_ctx* c, int rank, int world, const void* id_bytes) {
    return guarded([&]() -> int {
    if (!c || !id_bytes) return fail(FRHOT_EINVAL, "null argument");
    if (world < 1 || world > 256 || rank < 0 || rank >= world)
        return fail(FRHOT_EINVAL, "rank %d of %d out of range (1 <= world <= 256)", rank, world);
    Rccl* r = rccl();
    if (!r) return fail(FRHOT_ESTATE, "RCCL could not be loaded (%s)", g_rccl_why.c_str());
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->comm) return fail(FRHOT_ESTATE, "the context is rank %d of %d already", c->dist_rank, c->dist_world);
    DeviceGuard dg(c->device);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    RCCL_TRY(r, r->comm_init_rank(&comm, world, id, rank));
    c->comm = comm; c->dist_rank = rank; c->dist_world = world;
    return FRHOT_OK;
    });
}

int frhot_dist_shutdown(frhot_ctx* c) {
    return guarded([&]() -> int {
    if (!c) return fail(FRHOT_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    dist_release(c);
    return FRHOT_OK;
    });
}

int frhot_search_sharded_dev(frhot_gallery* g, const void* q_local, const int32_t* qkeys_local, int B, int k,
                             void* scores_dev, void* rows_dev, int64_t row_base) {
    return guarded([&]() -> int {
    if (!g || !q_local || !scores_dev || !rows_dev) return fail(FRHOT_EINVAL, "null argument");
    frhot_ctx* c = g->ctx;
    if (B <= 0 || k <= 0 || k > 16) return fail(FRHOT_EINVAL, "B %d / k %d out of range (k <= 16)", B, k);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->comm) return fail(FRHOT_ESTATE, "the context has joined no communicator (frhot_dist_init)");
    Rccl* r = rccl();
    if (!r) return fail(FRHOT_ESTATE, "RCCL could not be loaded (%s)", g_rccl_why.c_str());
    DeviceGuard dg(c->device);
    const int P = c->dist_world;
    const size_t n = (size_t)P * B, nk = n * k;
    if (n > (size_t)1 << 20) return fail(FRHOT_ECAPACITY, "%d ranks x %d queries: more than 2^20 queries in one step", P, B);
    // A shard's answer is one block of int64 words: its rows [n,k], then its scores (float32 [n,k],
    // four bytes of padding after an odd count), the layout the merge reads in place.
    const size_t words = nk + (nk + 1) / 2;
    int rc = grow(&c->xq, &c->xq_cap, n * g->D * 2);
    if (!rc && qkeys_local) rc = grow((void**)&c->xk, &c->xk_cap, n * 4);
    if (!rc) rc = grow((void**)&c->xblk, &c->xblk_cap, (size_t)(P + 1) * words * 8);
    if (rc) return rc;
    hipStream_t s = c->stream;
    ncclComm_t comm = (ncclComm_t)c->comm;
    int64_t* const mine = c->xblk;
    int64_t* const all = c->xblk + words;
    RCCL_TRY(r, r->all_gather(q_local, c->xq, (size_t)B * g->D, ncclFloat16, comm, s));
    if (qkeys_local) RCCL_TRY(r, r->all_gather(qkeys_local, c->xk, (size_t)B, ncclInt32, comm, s));
    rc = search_enqueue(g, c->xq, qkeys_local ? c->xk : nullptr, (int)n, k, (float*)(mine + nk), mine, row_base);
    if (rc) return rc;
    RCCL_TRY(r, r->all_gather(mine, all, words, ncclInt64, comm, s));
    HIP_TRY(launch_topk_final((const float*)(all + nk), all, P, k, (int)n, k, 0, (float*)scores_dev, (int64_t*)rows_dev, s,
                              nullptr, nullptr, (long)(words * 2), (long)words));
    return FRHOT_OK;
    });
}

static int recognise_impl(frhot_ctx* c, frhot_gallery* g, const uint8_t* const* crops, const int32_t* hw,
                          const int32_t* strides, const int32_t* qkeys, int B, unsigned flags, int k, float* emb_out,
                          float* scores, int64_t* rows, int64_t* sharp) {
    return guarded([&]() -> int {
    if (!c || !g || !scores || !rows) return fail(FRHOT_EINVAL, "null argument");
    if (g->ctx != c) return fail(FRHOT_EINVAL, "gallery belongs to another context");
    if (B <= 0 || B > c->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, c->max_batch);
    if (k <= 0 || k > 16) return fail(FRHOT_EINVAL, "k %d out of range (k <= 16)", k);
    if (g->D > kFcN) return fail(FRHOT_EINVAL, "gallery dimension %d exceeds the descriptor length", g->D);
    if (qkeys)
        for (int i = 0; i < B; ++i)
            if (qkeys[i] < 0 && qkeys[i] != FRHOT_KEY_ANY)
                return fail(FRHOT_EINVAL, "query key %d of crop %d is neither a row key nor FRHOT_KEY_ANY", qkeys[i], i);
    DeviceGuard dg(c->device);
    HostCall call(c);
    SlotLease lease(c);
    frhot_ctx::Slot& sl = lease.sl;
    int rc = emb_out ? slot_emb_mirror(c, sl) : FRHOT_OK;
    if (rc) return rc;
    CropStaging st(c, sl);
    rc = st.begin(crops, hw, strides, B, host_first_part(B, call.alone));
    if (rc) return rc;
    const std::vector<int64_t>& offsets = st.offsets;
    const std::vector<int32_t>& ps = st.packed_strides;
    float* const pin_s = (float*)sl.res_pin;
    int64_t* const pin_r = (int64_t*)(sl.res_pin + (size_t)c->max_batch * 16 * 4);
    rc = run_staged(c, sl, st, [&]() -> int {
        hipStream_t s = c->stream;
        int r = embed_enqueue(c, sl.dev, offsets.data(), hw, ps.data(), B, flags, c->emb_dev, c->q_dev, g->D, &st, st.split_at, !call.alone);
        if (r) return r;
        r = search_keyed_enqueue(g, c->q_dev, qkeys, c->qkeys_dev, B, k, c->res_s, c->res_r, sl.idx);
        if (r) return r;
        if (sharp) {
            r = sharpness_enqueue(c, sl.dev, offsets.data(), hw, ps.data(), B, c->sharp_dev, true);
            if (r) return r;
            HIP_TRY(hipMemcpyAsync(sl.sharp_pin, c->sharp_dev, (size_t)B * 16, hipMemcpyDeviceToHost, s));
        }
        if (emb_out) HIP_TRY(hipMemcpyAsync(sl.emb_pin, c->emb_dev, (size_t)B * kFcN * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(pin_s, c->res_s, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(pin_r, c->res_r, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
        return FRHOT_OK;
    });
    if (rc) return rc;
    if (emb_out) memcpy(emb_out, sl.emb_pin, (size_t)B * kFcN * 4);
    memcpy(scores, pin_s, (size_t)B * k * 4);
    memcpy(rows, pin_r, (size_t)B * k * 8);
    if (sharp) memcpy(sharp, sl.sharp_pin, (size_t)B * 16);
    return FRHOT_OK;
    });
}

int frhot_recognise(frhot_ctx* c, frhot_gallery* g, const uint8_t* const* crops, const int32_t* hw,
                    const int32_t* strides, const int32_t* qkeys, int B, unsigned flags, int k, float* emb_out,
                    float* scores, int64_t* rows) {
    return recognise_impl(c, g, crops, hw, strides, qkeys, B, flags, k, emb_out, scores, rows, nullptr);
}

int frhot_recognise_gated(frhot_ctx* c, frhot_gallery* g, const uint8_t* const* crops, const int32_t* hw,
                          const int32_t* strides, const int32_t* qkeys, int B, unsigned flags, int k, float* emb_out,
                          float* scores, int64_t* rows, int64_t* sharpness) {
    if (!sharpness) return guarded([&]() -> int { return fail(FRHOT_EINVAL, "sharpness is null"); });
    return recognise_impl(c, g, crops, hw, strides, qkeys, B, flags, k, emb_out, scores, rows, sharpness);
}

int frhot_ctx_timing(frhot_ctx* c, int enable) {
    return guarded([&]() -> int {
    if (!c) return fail(FRHOT_EINVAL, "ctx is null");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    if (enable && !c->ev_b[0][0]) {
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < frhot_ctx::kRing; ++j) {
                HIP_TRY(hipEventCreate(&c->ev_b[i][j]));
                HIP_TRY(hipEventCreate(&c->ev_e[i][j]));
            }
    }
    for (long& n : c->ev_n) n = 0;
    c->timing = enable != 0;
    return FRHOT_OK;
    });
}

int frhot_ctx_stage_ms(frhot_ctx* c, float* ms5) {
    return guarded([&]() -> int {
    if (!c || !ms5) return fail(FRHOT_EINVAL, "null argument");
    if (!c->timing) return fail(FRHOT_ESTATE, "timing is not enabled");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 5; ++i) {
        const long n = c->ev_n[i] < frhot_ctx::kRing ? c->ev_n[i] : frhot_ctx::kRing;
        double sum = 0.0;
        for (long j = 0; j < n; ++j) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev_b[i][j], c->ev_e[i][j]));
            sum += ms;
        }
        ms5[i] = n > 0 ? (float)(sum / (double)n) : 0.f;
    }
    return FRHOT_OK;
    });
}

}  // extern "C"

// ---- face detector -------------------------------------------------------------------------
// The network of DeepFace's "yolov8" backend (app/src/utils/legacy.py:60-65,80-86 reach it through
// DeepFace.extract_faces): YOLOv8n-pose layout with one class and five keypoints on a 640 x 640
// letterboxed frame.  The graph is fixed, so it is written out here as a list of launches over
// NHWC fp16 tensors; a concatenation is a tensor that two producers fill by channel slices.
struct DetTensor { _Float16* p = nullptr; int h = 0, w = 0, c = 0, div = 1; };   // h, w = canvas / div, set per call
struct DetSlice { const DetTensor* t; int c0, c; };

struct DetLayer { void* w = nullptr; float* b = nullptr; int k = 0, cin = 0, cout = 0; };   // cin / cout as stored (padded to 16)

struct frhot_detector {
    frhot_ctx* ctx = nullptr;
    int max_batch = 0;
    std::unordered_map<std::string, DetLayer> layers;
    std::vector<void*> owned;                       // every device allocation of the detector
    std::unordered_map<std::string, DetTensor> t;   // named tensors (also what frhot_det_debug_read serves)
    DetImageMeta* meta_dev = nullptr;
    _Float16* zeros = nullptr;                      // 4 KiB: the input of a tap outside the image
    DetCand* cand_dev = nullptr; int* count_dev = nullptr;
    DetCand* cand_pin = nullptr; int* count_pin = nullptr;
    static constexpr int kCap = 8400;               // candidates per image before suppression: every anchor of a full canvas
    std::mutex dmu;                                 // one detect call at a time (the activation tensors are the detector's)
};

namespace {

int det_alloc(frhot_detector* d, void** p, size_t n) {
    HIP_TRY(hipMalloc(p, n));
    d->owned.push_back(*p);
    return FRHOT_OK;
}

int det_tensor(frhot_detector* d, const char* name, int h, int w, int c) {
    DetTensor t; t.h = h; t.w = w; t.c = c; t.div = 640 / h;        // allocated for the full 640 x 640 canvas
    const int rc = det_alloc(d, (void**)&t.p, (size_t)d->max_batch * h * w * c * 2);
    if (rc) return rc;
    d->t[name] = t;
    return FRHOT_OK;
}

// One convolution of the blob ("<name>.w" fp16 [cout][k][k][cin], "<name>.b" fp32) with its channel
// counts padded to multiples of 16 by zeros (the first layer's 3 inputs, the heads' 1 and 15 outputs).
int det_load_layer(frhot_detector* d, const uint8_t* blob, size_t nbytes, const char* name, int k, int cin, int cout) {
    char nm[32];
    snprintf(nm, sizeof nm, "%s.w", name);
    const BlobEntry* ew = blob_find(blob, nbytes, nm);
    if (!ew || ew->dtype != 0 || ew->nbytes != (uint64_t)cout * k * k * cin * 2)
        return fail(FRHOT_EINVAL, "detector blob: bad or missing %s", nm);
    snprintf(nm, sizeof nm, "%s.b", name);
    const BlobEntry* eb = blob_find(blob, nbytes, nm);
    if (!eb || eb->dtype != 1 || eb->nbytes != (uint64_t)cout * 4) return fail(FRHOT_EINVAL, "detector blob: bad or missing %s", nm);
    const bool first = cin == 3;                    // the first layer: [16][64] with k = tap * 4 + channel (det_conv0_kernel)
    const int cip = first ? 4 : (cin + 15) / 16 * 16, cop = (cout + 15) / 16 * 16;
    std::vector<uint16_t> w(first ? (size_t)cop * 64 : (size_t)cop * k * k * cip, 0);
    const uint16_t* src = (const uint16_t*)(blob + ew->offset);
    for (int n = 0; n < cout; ++n)
        for (int t = 0; t < k * k; ++t)
            memcpy(&w[first ? (size_t)n * 64 + t * 4 : ((size_t)n * k * k + t) * cip], &src[((size_t)n * k * k + t) * cin], (size_t)cin * 2);
    std::vector<float> b(cop, 0.f);
    memcpy(b.data(), blob + eb->offset, (size_t)cout * 4);
    DetLayer L; L.k = k; L.cin = cip; L.cout = cop;
    int rc = det_alloc(d, &L.w, w.size() * 2);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(L.w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    rc = det_alloc(d, (void**)&L.b, b.size() * 4);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(L.b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    d->layers[name] = L;
    return FRHOT_OK;
}

struct DetRun {
    frhot_detector* d; int B; hipStream_t s;
    DetSlice all(const char* name) const { const DetTensor& t = d->t.at(name); return DetSlice{&t, 0, t.c}; }
    DetSlice part(const char* name, int c0, int c) const { return DetSlice{&d->t.at(name), c0, c}; }
    int conv(const char* layer, DetSlice in, DetSlice out, int stride, bool act, const DetSlice* res = nullptr) const {
        const DetLayer& L = d->layers.at(layer);
        if (L.cin != in.c || L.cout != out.c) return fail(FRHOT_ESTATE, "detector graph: %s is %d -> %d, wired %d -> %d", layer, L.cin, L.cout, in.c, out.c);
        DetConv c{};
        c.x = in.t->p; c.H = in.t->h; c.W = in.t->w; c.xc = in.t->c; c.xc0 = in.c0; c.cin = in.c;
        c.w = (const _Float16*)L.w; c.b = L.b;
        c.y = out.t->p; c.Ho = out.t->h; c.Wo = out.t->w; c.yc = out.t->c; c.yc0 = out.c0; c.cout = out.c;
        if (res) { c.r = res->t->p; c.rc = res->t->c; c.rc0 = res->c0; }
        c.zeros = d->zeros;
        c.k = L.k; c.stride = stride; c.act = act ? 1 : 0;
        if ((in.t->h + stride - 1) / stride != out.t->h || (in.t->w + stride - 1) / stride != out.t->w)
            return fail(FRHOT_ESTATE, "detector graph: %s maps %dx%d to %dx%d at stride %d", layer, in.t->h, in.t->w, out.t->h, out.t->w, stride);
        HIP_TRY(launch_det_conv(c, B, s));
        return FRHOT_OK;
    }
    // C2f: `buf` holds the (2 + n) halves side by side, `tmp` a bottleneck's inner map
    int c2f(const char* name, DetSlice in, const char* buf, const char* tmp, DetSlice out, int n, bool shortcut) const {
        const DetTensor& bt = d->t.at(buf);
        const int c = bt.c / (2 + n);
        char l[32];
        snprintf(l, sizeof l, "%s.cv1", name);
        int rc = conv(l, in, part(buf, 0, 2 * c), 1, true);
        for (int j = 0; j < n && !rc; ++j) {
            const DetSlice src = part(buf, (1 + j) * c, c);
            snprintf(l, sizeof l, "%s.m%d.cv1", name, j);
            rc = conv(l, src, all(tmp), 1, true);
            if (rc) break;
            snprintf(l, sizeof l, "%s.m%d.cv2", name, j);
            rc = conv(l, all(tmp), part(buf, (2 + j) * c, c), 1, true, shortcut ? &src : nullptr);
        }
        if (rc) return rc;
        snprintf(l, sizeof l, "%s.cv2", name);
        return conv(l, all(buf), out, 1, true);
    }
    int up2(DetSlice in, DetSlice out) const {
        HIP_TRY(launch_det_upsample2(in.t->p, in.t->c, in.c0, out.t->p, out.t->c, out.c0, in.c, in.t->h, in.t->w, B, s));
        return FRHOT_OK;
    }
    int pool5(DetSlice in, DetSlice out) const {
        HIP_TRY(launch_det_maxpool5(in.t->p, in.t->c, in.c0, out.t->p, out.t->c, out.c0, in.c, in.t->h, in.t->w, B, s));
        return FRHOT_OK;
    }
};

// The whole forward pass for B letterboxed frames of ch x cw pixels in tensor "in" (enqueue only).
int det_forward_enqueue(frhot_detector* d, int B, int ch, int cw) {
    for (auto& kv : d->t) { kv.second.h = ch / kv.second.div; kv.second.w = cw / kv.second.div; }
    DetRun r{d, B, d->ctx->stream};
#define FR_DET(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)
    {
        const DetLayer& L0 = d->layers.at("m0");
        HIP_TRY(launch_det_conv0(d->t.at("in").p, L0.w, L0.b, d->t.at("m0").p, B, ch, cw, d->ctx->stream));
    }
    FR_DET(r.conv("m1", r.all("m0"), r.all("m1"), 2, true));
    FR_DET(r.c2f("m2", r.all("m1"), "c2", "t2", r.all("m2"), 1, true));
    FR_DET(r.conv("m3", r.all("m2"), r.all("m3"), 2, true));
    FR_DET(r.c2f("m4", r.all("m3"), "c4", "t4", r.part("cat14", 128, 64), 2, true));          // P3, second part of cat(13, 4)
    FR_DET(r.conv("m5", r.part("cat14", 128, 64), r.all("m5"), 2, true));
    FR_DET(r.c2f("m6", r.all("m5"), "c6", "t6", r.part("cat11", 256, 128), 2, true));         // P4, second part of cat(10, 6)
    FR_DET(r.conv("m7", r.part("cat11", 256, 128), r.all("m7"), 2, true));
    FR_DET(r.c2f("m8", r.all("m7"), "c8", "t8", r.all("m8"), 1, true));
    FR_DET(r.conv("m9.cv1", r.all("m8"), r.part("cat9", 0, 128), 1, true));
    FR_DET(r.pool5(r.part("cat9", 0, 128), r.part("cat9", 128, 128)));
    FR_DET(r.pool5(r.part("cat9", 128, 128), r.part("cat9", 256, 128)));
    FR_DET(r.pool5(r.part("cat9", 256, 128), r.part("cat9", 384, 128)));
    FR_DET(r.conv("m9.cv2", r.all("cat9"), r.part("cat20", 128, 256), 1, true));                // P5, second part of cat(19, 9)
    FR_DET(r.up2(r.part("cat20", 128, 256), r.part("cat11", 0, 256)));
    FR_DET(r.c2f("m12", r.all("cat11"), "c12", "t12", r.part("cat17", 64, 128), 1, false));     // second part of cat(16, 12)
    FR_DET(r.up2(r.part("cat17", 64, 128), r.part("cat14", 0, 128)));
    FR_DET(r.c2f("m15", r.all("cat14"), "c15", "t15", r.all("o3"), 1, false));
    FR_DET(r.conv("m16", r.all("o3"), r.part("cat17", 0, 64), 2, true));
    FR_DET(r.c2f("m18", r.all("cat17"), "c18", "t18", r.all("o4"), 1, false));
    FR_DET(r.conv("m19", r.all("o4"), r.part("cat20", 0, 128), 2, true));
    FR_DET(r.c2f("m21", r.all("cat20"), "c21", "t21", r.all("o5"), 1, false));
    static const char* feat[3] = {"o3", "o4", "o5"};
    for (int l = 0; l < 3; ++l) {
        char a[32], b[32], o[32], n0[32], n1[32], n2[32];
        static const char* br[3] = {"box", "cls", "kpt"};
        for (int k = 0; k < 3; ++k) {
            snprintf(a, sizeof a, k == 2 ? "hk%da" : "h%da", l);
            snprintf(b, sizeof b, k == 2 ? "hk%db" : "h%db", l);
            snprintf(o, sizeof o, "%s%d", br[k], l);
            snprintf(n0, sizeof n0, "h.%s%d.0", br[k], l);
            snprintf(n1, sizeof n1, "h.%s%d.1", br[k], l);
            snprintf(n2, sizeof n2, "h.%s%d.2", br[k], l);
            FR_DET(r.conv(n0, r.all(feat[l]), r.all(a), 1, true));
            FR_DET(r.conv(n1, r.all(a), r.all(b), 1, true));
            FR_DET(r.conv(n2, r.all(b), r.all(o), 1, false));
        }
    }
#undef FR_DET
    return FRHOT_OK;
}

struct DetBox { float v[20]; int anchor; };

float det_iou(const float* a, const float* b) {
    const float ax1 = a[0] - a[2] / 2, ay1 = a[1] - a[3] / 2, ax2 = a[0] + a[2] / 2, ay2 = a[1] + a[3] / 2;
    const float bx1 = b[0] - b[2] / 2, by1 = b[1] - b[3] / 2, bx2 = b[0] + b[2] / 2, by2 = b[1] + b[3] / 2;
    const float iw = std::max(0.f, std::min(ax2, bx2) - std::max(ax1, bx1));
    const float ih = std::max(0.f, std::min(ay2, by2) - std::max(ay1, by1));
    const float inter = iw * ih;
    return inter / (a[2] * a[3] + b[2] * b[3] - inter + 1e-7f);
}

}  // namespace

extern "C" {

int frhot_det_create(frhot_ctx* c, const void* blob_v, size_t nbytes, int max_batch, frhot_detector** out) {
    return guarded([&]() -> int {
    if (!c || !blob_v || !out) return fail(FRHOT_EINVAL, "null argument");
    if (max_batch <= 0 || max_batch > 64) return fail(FRHOT_ECAPACITY, "max_batch %d outside 1..64", max_batch);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard dg(c->device);
    std::unique_ptr<frhot_detector> d(new (std::nothrow) frhot_detector());
    if (!d) return fail(FRHOT_EHIP, "out of host memory");
    d->ctx = c; d->max_batch = max_batch;
    struct Cleanup {
        std::unique_ptr<frhot_detector>& d; bool armed = true;
        ~Cleanup() { if (armed && d) { for (void* p : d->owned) (void)hipFree(p);
                                       if (d->cand_pin) (void)hipHostFree(d->cand_pin); if (d->count_pin) (void)hipHostFree(d->count_pin); } }
    } cleanup{d};
    const uint8_t* blob = (const uint8_t*)blob_v;
    int rc = FRHOT_OK;
    auto L = [&](const char* name, int k, int cin, int cout) { if (!rc) rc = det_load_layer(d.get(), blob, nbytes, name, k, cin, cout); };
    auto C2 = [&](const char* name, int c1, int c2, int n) {
        char l[32];
        const int ch = c2 / 2;
        snprintf(l, sizeof l, "%s.cv1", name); L(l, 1, c1, c2);
        for (int j = 0; j < n; ++j) {
            snprintf(l, sizeof l, "%s.m%d.cv1", name, j); L(l, 3, ch, ch);
            snprintf(l, sizeof l, "%s.m%d.cv2", name, j); L(l, 3, ch, ch);
        }
        snprintf(l, sizeof l, "%s.cv2", name); L(l, 1, (2 + n) * ch, c2);
    };
    L("m0", 3, 3, 16); L("m1", 3, 16, 32); C2("m2", 32, 32, 1); L("m3", 3, 32, 64); C2("m4", 64, 64, 2);
    L("m5", 3, 64, 128); C2("m6", 128, 128, 2); L("m7", 3, 128, 256); C2("m8", 256, 256, 1);
    L("m9.cv1", 1, 256, 128); L("m9.cv2", 1, 512, 256);
    C2("m12", 384, 128, 1); C2("m15", 192, 64, 1); L("m16", 3, 64, 64); C2("m18", 192, 128, 1);
    L("m19", 3, 128, 128); C2("m21", 384, 256, 1);
    static const int fch[3] = {64, 128, 256};
    for (int l = 0; l < 3; ++l) {
        char n[32];
        snprintf(n, sizeof n, "h.box%d.0", l); L(n, 3, fch[l], 64);
        snprintf(n, sizeof n, "h.box%d.1", l); L(n, 3, 64, 64);
        snprintf(n, sizeof n, "h.box%d.2", l); L(n, 1, 64, 64);
        snprintf(n, sizeof n, "h.cls%d.0", l); L(n, 3, fch[l], 64);
        snprintf(n, sizeof n, "h.cls%d.1", l); L(n, 3, 64, 64);
        snprintf(n, sizeof n, "h.cls%d.2", l); L(n, 1, 64, 1);
        snprintf(n, sizeof n, "h.kpt%d.0", l); L(n, 3, fch[l], 16);
        snprintf(n, sizeof n, "h.kpt%d.1", l); L(n, 3, 16, 16);
        snprintf(n, sizeof n, "h.kpt%d.2", l); L(n, 1, 16, 15);
    }
    if (rc) return rc;
    auto T = [&](const char* name, int hw, int ch) { if (!rc) rc = det_tensor(d.get(), name, hw, hw, ch); };
    T("in", 640, 4); T("m0", 320, 16); T("m1", 160, 32); T("c2", 160, 48); T("t2", 160, 16); T("m2", 160, 32);
    T("m3", 80, 64); T("c4", 80, 128); T("t4", 80, 32); T("cat14", 80, 192);
    T("m5", 40, 128); T("c6", 40, 256); T("t6", 40, 64); T("cat11", 40, 384);
    T("m7", 20, 256); T("c8", 20, 384); T("t8", 20, 128); T("m8", 20, 256); T("cat9", 20, 512); T("cat20", 20, 384);
    T("c12", 40, 192); T("t12", 40, 64); T("cat17", 40, 192);
    T("c15", 80, 96); T("t15", 80, 32); T("o3", 80, 64);
    T("c18", 40, 192); T("t18", 40, 64); T("o4", 40, 128);
    T("c21", 20, 384); T("t21", 20, 128); T("o5", 20, 256);
    static const int fhw[3] = {80, 40, 20};
    for (int l = 0; l < 3; ++l) {
        char n[32];
        snprintf(n, sizeof n, "h%da", l); T(n, fhw[l], 64);
        snprintf(n, sizeof n, "h%db", l); T(n, fhw[l], 64);
        snprintf(n, sizeof n, "hk%da", l); T(n, fhw[l], 16);
        snprintf(n, sizeof n, "hk%db", l); T(n, fhw[l], 16);
        snprintf(n, sizeof n, "box%d", l); T(n, fhw[l], 64);
        snprintf(n, sizeof n, "cls%d", l); T(n, fhw[l], 16);
        snprintf(n, sizeof n, "kpt%d", l); T(n, fhw[l], 16);
    }
    if (rc) return rc;
    rc = det_alloc(d.get(), (void**)&d->zeros, 4096);
    if (!rc) HIP_TRY(hipMemset(d->zeros, 0, 4096));
    if (!rc) rc = det_alloc(d.get(), (void**)&d->meta_dev, sizeof(DetImageMeta) * max_batch);
    if (!rc) rc = det_alloc(d.get(), (void**)&d->cand_dev, sizeof(DetCand) * (size_t)max_batch * frhot_detector::kCap);
    if (!rc) rc = det_alloc(d.get(), (void**)&d->count_dev, sizeof(int) * max_batch);
    if (rc) return rc;
    HIP_TRY(hipHostMalloc((void**)&d->cand_pin, sizeof(DetCand) * (size_t)max_batch * frhot_detector::kCap, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&d->count_pin, sizeof(int) * max_batch, hipHostMallocDefault));
    cleanup.armed = false;
    *out = d.release();
    return FRHOT_OK;
    });
}

int frhot_det_destroy(frhot_detector* d) {
    return guarded([&]() -> int {
    if (!d) return FRHOT_OK;
    {
        std::lock_guard<std::mutex> lk(d->ctx->mu);
        DeviceGuard dg(d->ctx->device);
        (void)hipStreamSynchronize(d->ctx->stream);
        for (void* p : d->owned) (void)hipFree(p);
        if (d->cand_pin) (void)hipHostFree(d->cand_pin);
        if (d->count_pin) (void)hipHostFree(d->count_pin);
    }
    delete d;
    return FRHOT_OK;
    });
}

int frhot_det_detect(frhot_detector* d, const uint8_t* const* images, const int32_t* hw, const int32_t* strides, int B,
                     float conf, float iou, int max_faces, float* out, int32_t* counts) {
    return guarded([&]() -> int {
    if (!d || !images || !hw || !strides || !out || !counts) return fail(FRHOT_EINVAL, "null argument");
    if (B <= 0 || B > d->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, d->max_batch);
    if (max_faces <= 0 || max_faces > 300) return fail(FRHOT_EINVAL, "max_faces %d outside 1..300", max_faces);
    if (!(conf > 0.f && conf < 1.f) || !(iou > 0.f && iou <= 1.f)) return fail(FRHOT_EINVAL, "conf %g / iou %g out of range", conf, iou);
    frhot_ctx* c = d->ctx;
    std::lock_guard<std::mutex> dlk(d->dmu);
    // the frames go through one of the context's staging slots: packed by a few host threads into its
    // page-locked mirror and uploaded in pieces on the copy stream, like the crops of frhot_embed
    DeviceGuard dg(c->device);
    SlotLease lease(c);
    frhot_ctx::Slot& sl = lease.sl;
    CropStaging st(c, sl);
    int rc = st.begin(images, hw, strides, B, 0);
    if (rc) return rc;
    // Letterbox geometry as ultralytics' LetterBox computes it (float64, Python round = half to even).
    // DeepFace predicts every frame on its own, which pads it only to the next multiple of 32 in each
    // direction; frames whose canvases agree run as one batch here, the others one group after the other.
    std::vector<DetImageMeta> meta(B);
    std::vector<std::pair<int, int>> canvas(B);
    for (int i = 0; i < B; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        DetImageMeta& m = meta[i];
        m.offset = st.offsets[i]; m.h = h; m.w = w; m.stride = st.packed_strides[i];
        const double r = std::min(640.0 / h, 640.0 / w);
        m.nw = (int)std::nearbyint(w * r); m.nh = (int)std::nearbyint(h * r);
        if (m.nw < 1 || m.nh < 1) return fail(FRHOT_EINVAL, "image %d: %dx%d letterboxes to an empty image", i, h, w);
        const int pw = (640 - m.nw) % 32, ph = (640 - m.nh) % 32;
        m.top = (int)std::nearbyint(ph / 2.0 - 0.1); m.left = (int)std::nearbyint(pw / 2.0 - 0.1);
        m.sy = (double)h / m.nh; m.sx = (double)w / m.nw;
        canvas[i] = {m.nh + ph, m.nw + pw};
    }
    std::vector<char> done(B, 0);
    static const int strides_l[3] = {8, 16, 32};
    for (int first = 0; first < B; ++first) {
        if (done[first]) continue;
        std::vector<int> idx;
        std::vector<DetImageMeta> gm;
        for (int i = first; i < B; ++i)
            if (!done[i] && canvas[i] == canvas[first]) { idx.push_back(i); gm.push_back(meta[i]); done[i] = 1; }
        const int n_g = (int)idx.size(), ch = canvas[first].first, cw = canvas[first].second;
        // the context's lock is held to enqueue only; the waits for the answers leave it to other callers
        std::unique_lock<std::mutex> lk(c->mu);
        hipStream_t s = c->stream;
        if (first == 0) HIP_TRY(hipStreamWaitEvent(s, sl.uploaded, 0));
        HIP_TRY(hipMemcpyAsync(d->meta_dev, gm.data(), sizeof(DetImageMeta) * n_g, hipMemcpyHostToDevice, s));
        // (Replaying the letterbox and the ~110 launches of the forward pass from a captured graph was
        // measured: 1.68 -> 1.62 ms for one frame.  The small maps' kernels are bound by their own serial
        // K loops on a handful of blocks, not by launch gaps, so the launches stay plain.)
        HIP_TRY(launch_det_letterbox(sl.dev, d->meta_dev, d->t.at("in").p, ch, cw, n_g, s));
        rc = det_forward_enqueue(d, n_g, ch, cw);
        if (rc) { (void)hipStreamSynchronize(s); return rc; }
        DetHead heads[3];
        for (int l = 0; l < 3; ++l) {
            char n[16];
            snprintf(n, sizeof n, "box%d", l); heads[l].box = d->t.at(n).p; heads[l].h = d->t.at(n).h; heads[l].w = d->t.at(n).w;
            snprintf(n, sizeof n, "cls%d", l); heads[l].cls = d->t.at(n).p;
            snprintf(n, sizeof n, "kpt%d", l); heads[l].kpt = d->t.at(n).p;
            heads[l].stride = strides_l[l];
        }
        HIP_TRY(launch_det_decode(heads, conf, d->cand_dev, d->count_dev, frhot_detector::kCap, n_g, s));
        HIP_TRY(hipMemcpyAsync(d->count_pin, d->count_dev, sizeof(int) * n_g, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(sl.done, s));
        lk.unlock();
        HIP_TRY(hipEventSynchronize(sl.done));
        // the survivors of every frame (their number is known only now), then suppression and the way back
        // to the frame on the host: a handful of boxes
        for (int j = 0; j < n_g; ++j)
            if (d->count_pin[j] > frhot_detector::kCap)
                return fail(FRHOT_ECAPACITY, "image %d: %d anchors above the confidence threshold (at most %d are kept)",
                            idx[j], d->count_pin[j], frhot_detector::kCap);
        lk.lock();
        for (int j = 0; j < n_g; ++j)
            if (d->count_pin[j] > 0)
                HIP_TRY(hipMemcpyAsync(d->cand_pin + (size_t)j * frhot_detector::kCap, d->cand_dev + (size_t)j * frhot_detector::kCap,
                                       sizeof(DetCand) * d->count_pin[j], hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(sl.done, s));
        lk.unlock();
        HIP_TRY(hipEventSynchronize(sl.done));
        for (int j = 0; j < n_g; ++j) {
            const int i = idx[j], n = d->count_pin[j];
            const DetCand* cand = d->cand_pin + (size_t)j * frhot_detector::kCap;
            std::vector<int> order(n);
            for (int q = 0; q < n; ++q) order[q] = q;
            std::sort(order.begin(), order.end(), [&](int a, int b) {
                if (cand[a].v[4] != cand[b].v[4]) return cand[a].v[4] > cand[b].v[4];
                return cand[a].anchor < cand[b].anchor;
            });
            std::vector<int> kept;
            for (int q : order) {
                bool ok = true;
                for (int p : kept) if (det_iou(cand[q].v, cand[p].v) > iou) { ok = false; break; }
                if (ok) { kept.push_back(q); if ((int)kept.size() == max_faces) break; }
            }
            const DetImageMeta& m = meta[i];
            const double r = std::min(640.0 / m.h, 640.0 / m.w);
            auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
            float* o = out + (size_t)i * max_faces * 20;
            for (size_t q = 0; q < kept.size(); ++q) {
                const float* v = cand[kept[q]].v;
                float* w = o + q * 20;
                w[0] = (float)clampd((v[0] - v[2] / 2 - m.left) / r, 0, m.w);
                w[1] = (float)clampd((v[1] - v[3] / 2 - m.top) / r, 0, m.h);
                w[2] = (float)clampd((v[0] + v[2] / 2 - m.left) / r, 0, m.w);
                w[3] = (float)clampd((v[1] + v[3] / 2 - m.top) / r, 0, m.h);
                w[4] = v[4];
                for (int k = 0; k < 5; ++k) {
                    w[5 + 3 * k] = (float)clampd((v[5 + 3 * k] - m.left) / r, 0, m.w);
                    w[6 + 3 * k] = (float)clampd((v[6 + 3 * k] - m.top) / r, 0, m.h);
                    w[7 + 3 * k] = v[7 + 3 * k];
                }
            }
            counts[i] = (int32_t)kept.size();
        }
    }
    return FRHOT_OK;
    });
}

int frhot_det_debug_read(frhot_detector* d, const char* name, int B, void* host_out, size_t nbytes) {
    return guarded([&]() -> int {
    if (!d || !name || !host_out) return fail(FRHOT_EINVAL, "null argument");
    if (B <= 0 || B > d->max_batch) return fail(FRHOT_ECAPACITY, "batch %d outside 1..%d", B, d->max_batch);
    std::lock_guard<std::mutex> dlk(d->dmu);        // (a detect call in flight changes the tensors' sizes)
    const auto it = d->t.find(name);
    if (it == d->t.end()) return fail(FRHOT_EINVAL, "the detector has no tensor '%s'", name);
    const size_t need = (size_t)B * it->second.h * it->second.w * it->second.c * 2;
    if (nbytes != need) return fail(FRHOT_EINVAL, "tensor '%s' of %d images is %zu bytes, the buffer %zu", name, B, need, nbytes);
    std::lock_guard<std::mutex> lk(d->ctx->mu);
    DeviceGuard dg(d->ctx->device);
    HIP_TRY(hipStreamSynchronize(d->ctx->stream));
    HIP_TRY(hipMemcpy(host_out, it->second.p, need, hipMemcpyDeviceToHost));
    return FRHOT_OK;
    });
}

}  // extern "C"
