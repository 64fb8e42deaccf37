// Launch wrappers of the HIP kernels behind libfrhot.so (gfx950 only).
// Every wrapper validates shapes on the host, enqueues on `stream` and returns
// a hipError_t; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace frhot {

// Geometry of one crop, computed on the host (frhot_api.cpp: fill_crop_meta).
struct CropMeta {
    int64_t offset;     // byte offset of the image inside the packed crop buffer
    int32_t h, w;       // source rows, columns
    int32_t stride;     // bytes between source rows
    int32_t dw, dh;     // resized size (aspect fit into 224x224, truncated)
    int32_t padx, pady; // left / top zero padding
    int32_t area2x;     // 1: exact 2x downscale -> 2x2 box average
    double scale_x, scale_y;  // 1/(dw/w), 1/(dh/h) in float64
};

struct ConvLayerDesc {
    int hw;       // input (= output before pooling) height and width
    int cin, cout;
    int pool;     // 1: 2x2/2 max-pool fused into the epilogue
    int opad;     // 1: output carries a zero halo of one pixel
    int reverse;  // 1: walk the pixel tiles from the last image to the first (cache reuse across layers)
    int blocks = 0;  // persistent blocks of the launch (0: one per CU); fewer leave CUs to a launch on another stream
};

// Resized bytes, per-image maximum, and the fp16 network input [B,226,226,4] of every image whose
// maximum exceeds 1; launch_normalize_pack(..., only_undivided = true) afterwards does the others.
hipError_t launch_resize_u8(const uint8_t* crops, const CropMeta* meta, const uint8_t* lut,
                            int flip, uint8_t* out_u8, int* out_max, float scale, void* x0_f16, int B,
                            hipStream_t s);
// out[b] = {sum, sum of squares} of the 3x3 Laplacian of crop b's grey image (int64 pairs, device);
// max_tiles: the largest ceil(w/256) * ceil(h/16) of the batch.
hipError_t launch_crop_sharpness(const uint8_t* crops, const CropMeta* meta, int max_tiles, int64_t* out, int B,
                                 hipStream_t s);
hipError_t launch_normalize_pack(const uint8_t* u8, const int* img_max, float scale,
                                 void* x0_f16, int B, hipStream_t s, bool only_undivided = false);
hipError_t launch_conv1_1(const void* x0, const void* w64, const float* bias, void* y,
                          int B, hipStream_t s);
// The first two layers as one launch: conv1_1 computed into the LDS patches of the 224 x 224 x 64 layer
// (pooled output y [B,114,114,64]); bit-identical to launch_conv1_1 followed by launch_conv3x3.
hipError_t launch_conv1_fused(const void* x0, const void* w64, const float* bias1, const void* w, const float* bias,
                              void* y, void* dump, int B, int reverse, hipStream_t s, int blocks = 0);   // blocks: 0 = one per CU
// `dump`: >= 1 KiB of device scratch (dead lanes of a partial tile store there).
hipError_t launch_conv3x3(const ConvLayerDesc& d, const void* x, const void* w, const float* bias,
                          void* y, void* dump, int B, hipStream_t s);
// Prototype (libraries built with -DFR_WINOGRAD=1 only; the micro-benchmark): the same convolution
// by Winograd F(2x2,3x3) for the layers that have such a form (others: hipErrorInvalidValue).  U = launch_wino_weights(w): 16 * cout * cin fp16, made once per layer.
hipError_t launch_wino_weights(const void* w, void* U, int cin, int cout, hipStream_t s);
hipError_t launch_conv3x3_wino(const ConvLayerDesc& d, const void* x, const void* U, const float* bias, void* y,
                               int B, hipStream_t s);
// ---- face detector (YOLOv8n-pose layout): NHWC fp16 tensors addressed as channel slices ----
struct DetConv {
    const _Float16* x; int H, W, xc, xc0, cin;       // input [B,H,W,xc], channels xc0 .. xc0+cin (cin % 16 == 0)
    const _Float16* w; const float* b;               // [cout][k][k][cin] fp16, [cout] fp32 (cout % 16 == 0)
    _Float16* y; int Ho, Wo, yc, yc0, cout;          // output [B,Ho,Wo,yc], channels yc0 .. yc0+cout
    const _Float16* r; int rc, rc0;                  // residual slice (output geometry) added after the activation, or null
    const _Float16* zeros;                           // 4 KiB of zeros: what a tap outside the image reads
    int k, stride, act;                              // k in {1,3}, stride in {1,2} (k=1: stride 1), act: SiLU
};
struct DetImageMeta {
    int64_t offset; int h, w, stride;                // the frame inside the staged bytes
    int nh, nw, top, left;                           // resized size and its place on the canvas
    double sy, sx;                                   // source pixels per resized pixel
};
struct DetHead { const _Float16 *box, *cls, *kpt; int h, w, stride; };   // [B,h,w,64], [B,h,w,16] (channel 0), [B,h,w,16] (15)
struct DetCand { float v[20]; int anchor; };         // cx, cy, w, h (canvas), score, 5 x (x, y, visibility)
hipError_t launch_det_conv(const DetConv& d, int B, hipStream_t s);
hipError_t launch_det_conv0(const void* x, const void* w64, const float* bias, void* y, int B, int H, int W, hipStream_t s);
hipError_t launch_det_maxpool5(const void* x, int xc, int xc0, void* y, int yc, int yc0, int C, int H, int W, int B,
                               hipStream_t s);
hipError_t launch_det_upsample2(const void* x, int xc, int xc0, void* y, int yc, int yc0, int C, int H, int W, int B,
                                hipStream_t s);
hipError_t launch_det_letterbox(const uint8_t* imgs, const DetImageMeta* meta, void* out4, int ch, int cw, int B, hipStream_t s);   // [B,ch,cw,4]
hipError_t launch_det_decode(const DetHead* heads3, float conf, DetCand* cand, int* counts, int cap, int B, hipStream_t s);
// Split-K GEMM partial[z][m][n] = sum_{k in slice z} x[m][k] * w[n][k]; M rows of x,
// N rows of w, K multiple of 64*splitk, N multiple of 128.
hipError_t launch_fc_splitk(const void* x, const void* w, float* partial, int M, int N, int K,
                            int splitk, hipStream_t s);
hipError_t launch_fc_reduce_relu(const float* partial, const float* bias, void* y_f16,
                                 int M, int N, int splitk, hipStream_t s);
// fc7 epilogue: bias + ReLU + L2 norm; writes unit f32 [M,N] and/or fp16 [M,Dq]
// (leading Dq components re-normalised).
hipError_t launch_fc_reduce_relu_l2(const float* partial, const float* bias, float* out_f32,
                                    void* out_q_f16, int Dq, int M, int N, int splitk, hipStream_t s);

hipError_t launch_gallery_upsert(const float* vecs, const int32_t* keys, const int64_t* rows, int n,
                                 void* g_f16, int32_t* g_keys, int D, hipStream_t s);
hipError_t launch_gallery_set_keys(const int64_t* rows, int n, int32_t key, int32_t* g_keys, hipStream_t s);
hipError_t launch_gallery_fill(void* g_f16, int32_t* g_keys, int D, uint64_t seed, int64_t row0,
                               int64_t n, int nkeys, int dup_every, hipStream_t s);
hipError_t launch_gallery_gather(const void* g_f16, const int32_t* g_keys, const int64_t* rows, int n,
                                 int D, void* out_f16, int32_t* out_keys, hipStream_t s);
hipError_t launch_gallery_scatter(const void* in_f16, const int32_t* in_keys, const int64_t* rows, int n,
                                  int D, void* g_f16, int32_t* g_keys, hipStream_t s);
hipError_t launch_f32_to_unit_f16(const float* q, void* q_f16, int Bq, int D, hipStream_t s);

// Scan: per-block partial top-k lists into part_s/part_r ([nblk][Bq][k]); returns nblk via *nblk_out.
int scan_max_blocks();
// g_f16 / g_keys must be allocated in whole tiles of 256 rows (reads run to the tile end).
// qbound: scan_bound_bytes(Bq) bytes of device scratch (cross-block score bounds), or null.  The
// scratch must be ZEROED once when it is allocated: the 64-query kernel never clears it (its slots
// carry the number of the launch that wrote them), the wider kernels clear it on the stream themselves.
size_t scan_bound_bytes(int Bq);
void scan_set_sel_variant(int v);          // 64-query tile: 0 = the kernel without the selector wave (micro-benchmark A/B), 1 = with
void scan_set_bound_min_tiles(int tiles);   // blocks that walk fewer tiles scan without the bounds (< 0: built-in rule)
hipError_t launch_scan_topk(const void* g_f16, const int32_t* g_keys, int64_t nrows, int D,
                            const void* q_f16, const int32_t* qkeys, int use_keys, int Bq, int k,
                            float* part_s, int64_t* part_r, unsigned* qbound, int* nblk_out, hipStream_t s);
// One block's share of a search that looks up every key's queries in that key's tiles only.
struct ScanWork {
    int32_t q0, qend;       // first query of the block's 64-query tile; end of its key's queries
    int32_t tile_off, ntl;  // the key's tile list: offset into tile_lists (< 0: every tile of the gallery) and length
    int32_t gb, nb;         // this block walks tiles gb, gb + nb, ... of the list
};
// One launch over `nwork` such items; the queries (device, [Bq, D] fp16) are in key order and
// qkeys holds their keys in the same order.  Candidate lists land at
// part[(gb * Bq + query) * list length]: part_s / part_r hold (largest nb) * Bq lists.
// qbound: null, or scan_bound_bytes(Bq + 64) bytes (query tiles start at any query here).
hipError_t launch_scan_topk_work(const void* g_f16, const int32_t* g_keys, int64_t nrows, int D, const void* q_f16,
                                 const int32_t* qkeys, int Bq, int k, float* part_s, int64_t* part_r,
                                 unsigned* qbound, const int32_t* tile_lists, const ScanWork* work, int nwork,
                                 hipStream_t s);
// out[i] = in[index[i]]: n rows of D fp16 (D a multiple of 8), all device memory.
hipError_t launch_rows_gather_f16(const void* in, const int32_t* index, int n, int D, void* out, hipStream_t s);
// Upper bound on the list entries (one score and one row each) launch_scan_topk writes to
// part_s / part_r for Bq queries of top-k, whatever the gallery size.
size_t scan_part_entries(int Bq, int k);
// The static device scratch of the allocation-free search (a file-scope __device__ array of the
// code object; every GPU holds its own copy).  Valid after kernels_init() ran with `device` current.
struct ScanStaticScratch {
    float* part_s; int64_t* part_r; unsigned* qbound;
    size_t entries;       // list entries part_s / part_r hold
    size_t bound_bytes;   // bytes behind qbound
};
hipError_t scan_static_scratch(int device, ScanStaticScratch* out);
// One user of a GPU's static scratch at a time.  claim: `owner` (any stable address, e.g. the
// calling context) takes the array over; hipErrorNotReady while work another owner marked is still
// running on that GPU.  mark: record, on the stream the owner just enqueued on, the point after
// which the array is free again.  Neither allocates.
hipError_t scan_static_claim(int device, const void* owner);
hipError_t scan_static_mark(int device, const void* owner, hipStream_t s);
// Length of the per-block candidate lists the scan writes for a request of k (5, 10 or 16).
int scan_list_len(int k);
// Final selection over nlists lists of kin (layout [nlists][Bq][kin]) -> best kout per query.
hipError_t launch_topk_final(const float* cand_s, const int64_t* cand_r, int nlists, int kin, int Bq, int kout,
                             int64_t row_base, float* out_s, int64_t* out_r, hipStream_t s,
                             const int32_t* out_index = nullptr,      // query i's answer goes to row out_index[i]
                             const int32_t* nlists_q = nullptr,       // query i merges its first nlists_q[i] lists
                             int64_t stride_s = 0, int64_t stride_r = 0);   // elements between consecutive lists' blocks
                                                                            // in cand_s / cand_r (0: Bq * kin, back to back)

// Sets the dynamic-LDS attributes and resolves the static scratch's address on the CURRENT device;
// call once per device (repeat calls are cheap).
hipError_t kernels_init();

}  // namespace frhot
