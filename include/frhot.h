/*
 * frhot.h — C ABI of libfrhot.so, the MI355X (gfx950) recognition hot path.
 *
 * The library replaces three things the reference reaches through Python
 * packages and HTTP hops: the DeepFace "VGG-Face" descriptor, the Qdrant exact
 * cosine search, and the glue between them.  Every entry point below names the
 * reference interface it stands in for (paths relative to the reference repo).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns
 *     FRHOT_OK (0) or a negative FRHOT_E* code, and frhot_last_error() gives the
 *     calling thread's message for the last failure;
 *   - "host" pointers are ordinary process memory owned by the caller; "dev"
 *     pointers are device (HBM) addresses on the context's GPU, owned by the
 *     caller (for example torch tensors); the library owns what it allocates
 *     behind the opaque handles;
 *   - a context is bound to one GPU and one HIP stream; calls on one context
 *     are serialised by an internal mutex, so callers may share it between
 *     threads; *_dev calls only enqueue work on the stream and do not wait.
 */
#ifndef FRHOT_H
#define FRHOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRHOT_OK            0
#define FRHOT_EINVAL       (-1)   /* bad argument / shape */
#define FRHOT_EHIP         (-2)   /* a HIP runtime call failed */
#define FRHOT_ESTATE       (-3)   /* call order (model not loaded, ...) */
#define FRHOT_ECAPACITY    (-4)   /* batch / gallery capacity exceeded */

#define FRHOT_EMBED_DIM     4096
#define FRHOT_INPUT_SIZE    224

/* preprocessing flags of frhot_embed* */
#define FRHOT_FLAG_GAMMA    1u    /* apply the gamma-1.5 LUT first (app/src/utils/legacy.py:106-110) */
#define FRHOT_FLAG_NOFLIP   2u    /* skip the channel reverse DeepFace.represent performs */

/* query keys */
#define FRHOT_KEY_ANY       (-2)  /* query matches every live row */
#define FRHOT_KEY_DEAD      (-1)  /* row key of a tombstoned row; as a query key it matches nothing */

typedef struct frhot_ctx frhot_ctx;
typedef struct frhot_gallery frhot_gallery;

/* Message of the last failing call on this thread ("" if none). */
const char* frhot_last_error(void);

/* Library build information (arch, kernel set); static string. */
const char* frhot_version(void);

/*
 * Context = one GPU + one stream + the model + activation workspace.
 * `stream` is a hipStream_t to enqueue on (NULL: the library creates its own).
 * `max_batch` (1..2048) bounds the crops one frhot_embed* call may carry; larger
 * batches go out as several calls.
 * Replaces: the per-process DeepFace model singleton warmed up in
 * app/src/api/app.py:112-118.
 */
int frhot_ctx_create(int device_id, void* stream, int max_batch, frhot_ctx** out);
int frhot_ctx_destroy(frhot_ctx* ctx);
int frhot_ctx_sync(frhot_ctx* ctx);

/*
 * Load VGG-Face weights from the library's flat blob format (see
 * frhot/weights.py): 13 conv tensors [Cout,3,3,Cin], fc6 [4096,7,7,512],
 * fc7 [4096,4096], all fp16, and fp32 biases.  `input_scale` (a power of two,
 * e.g. 1/16) multiplies the network input and every bias so that activations
 * stay inside fp16 range; it cancels in the final L2 normalisation.
 * Replaces: the weight download + Keras model build behind
 * DeepFace.represent(model_name="VGG-Face") (app/src/utils/legacy.py:52-54,
 * app/Dockerfile:79-81).
 */
int frhot_model_load(frhot_ctx* ctx, const void* blob, size_t nbytes, float input_scale);

/*
 * Embed B crops given as host buffers: crops[i] points at an uint8 BGR HWC
 * image of hw[2i] rows x hw[2i+1] columns whose rows are strides[i] bytes apart.
 * Writes B unit-norm float32 descriptors of 4096 components to `out` (host).
 * Replaces: get_embedding(imgf, imgf_real)[0] (app/src/utils/legacy.py:48-59,74),
 * i.e. DeepFace.represent(detector_backend="skip", normalization="VGGFace2")
 * including its resize / pad / mean subtraction and the final L2 norm.
 */
int frhot_embed(frhot_ctx* ctx, const uint8_t* const* crops, const int32_t* hw,
                const int32_t* strides, int B, unsigned flags, float* out);

/*
 * Same computation with the pixels already in HBM: `crops_dev` is one device
 * buffer, image i starts at byte offsets[i] (host array) with geometry hw /
 * strides (host arrays).  Outputs are device pointers (either may be NULL):
 * `out_f32_dev` [B,4096] float32 unit-norm; `out_q_f16_dev` [B,Dq] fp16 holding
 * the leading Dq components re-normalised to unit length (Dq == 4096 is the
 * descriptor itself) — the layout frhot_search_dev takes as queries.
 * Enqueues on the context stream and returns without waiting.
 */
int frhot_embed_dev(frhot_ctx* ctx, const void* crops_dev, const int64_t* offsets,
                    const int32_t* hw, const int32_t* strides, int B, unsigned flags,
                    void* out_f32_dev, void* out_q_f16_dev, int Dq);

/*
 * Stage outputs for tests and profiling (device -> host copies, waits):
 * stage 0: uint8 [B,224,224,3] resized+padded image (before float conversion);
 * stage 1: fp16 [B,226,226,4] network input (halo and 4th channel zero);
 * stage 10+l (l = 0..12): fp16 output of conv layer l (conv1_1 .. conv5_3, after its
 *   pooling where it has one), [B,P,P,C] with a one-pixel zero halo except l = 12; the embed path
 *   computes conv1_1 inside conv1_2's kernel and never stores it, so stage 10 is made on request from
 *   stage 1 by the stand-alone first-layer kernel (same arithmetic, same bits), into a buffer of
 *   6.4 MB per face of max_batch that is allocated at the first such request;
 * stage 30: fp16 [B,4096] fc6 output (input-scaled like every activation).
 */
int frhot_debug_read(frhot_ctx* ctx, int stage, int B, void* host_out, size_t nbytes);

/*
 * Gallery = device-resident fp16 unit rows ([capacity, D]; kept in HBM in a tiled order private to
 * the library, rows cross this interface row-major) + int32 key per row.
 * A key stands for the reference's collection name f"{store_id}_Employees" /
 * f"{store_id}_Customers" (app/src/services/face_service.py:355-360), which
 * also subsumes the store_id payload filter of qdrant_database_FE/app.py:320-327.
 * Replaces: create_collection (qdrant_database_FE/app.py:108-128).
 */
int frhot_gallery_create(frhot_ctx* ctx, int D, int64_t capacity, frhot_gallery** out);
int frhot_gallery_destroy(frhot_gallery* g);

/*
 * Write n rows: vecs [n,D] float32 (host; normalised to unit length inside, a
 * zero row is stored as zeros), keys[n], destination row index rows[n].
 * Replaces: client.upsert of qdrant_database_FE/app.py:289-293 (Qdrant
 * normalises COSINE vectors at upsert).
 */
int frhot_gallery_upsert(frhot_gallery* g, const float* vecs, const int32_t* keys,
                         const int64_t* rows, int n);

/*
 * How frhot_search / frhot_recognise (the calls that know their queries' keys on the host) look
 * for rows.  The reference keeps every collection in storage of its own and a search reads that
 * collection alone (qdrant_database_FE/app.py:108-128,316-337).  The gallery is one table, so the
 * library keeps, per key, the list of 256-row tiles that may hold rows of the key, and a by-key
 * search scans only those (one launch: every block takes one key's queries and a share of that
 * key's tiles).  The answers are the same either way (every row's key is still tested).
 *   FRHOT_SEARCH_AUTO    by key when a cost estimate says it is the faster way (default)
 *   FRHOT_SEARCH_BY_KEY  by key whenever the queries carry keys
 *   FRHOT_SEARCH_WHOLE   always one scan of the whole gallery
 * Rows of a key that sit together (the host side allocates a key's rows from tiles of its own
 * while it can) make the lists short; a gallery whose keys are strewn over every tile (more than
 * 8 list entries per tile on average) gives the lists up and scans whole, whatever the mode.  The *_dev searches, whose keys live on the device, always
 * scan the whole gallery.
 */
#define FRHOT_SEARCH_AUTO    0
#define FRHOT_SEARCH_BY_KEY  1
#define FRHOT_SEARCH_WHOLE   2
int frhot_gallery_search_mode(frhot_gallery* g, int mode);

/* Mark rows deleted (key := FRHOT_KEY_DEAD).  Replaces: client.delete of
 * qdrant_database_FE/app.py:237-240,423-435. */
int frhot_gallery_tombstone(frhot_gallery* g, const int64_t* rows, int n);

/*
 * Fill rows [row0, row0+n) with the library's deterministic synthetic gallery
 * (integer hash -> unit fp16 rows; `nkeys` > 0 assigns key = hash % nkeys, `nkeys`
 * < 0 key = row / -nkeys, i.e. runs of -nkeys rows per key as collections stored
 * one after the other, 0 key 0) and plant every `dup_every`-th row as an exact duplicate of its
 * predecessor (0: none).  Benchmark / test data only; no reference counterpart.
 */
int frhot_gallery_fill_synthetic(frhot_gallery* g, uint64_t seed, int64_t row0, int64_t n,
                                 int nkeys, int dup_every);

/* Copy n stored rows back as fp16 [n,D] plus their keys (host buffers). */
int frhot_gallery_read(frhot_gallery* g, const int64_t* rows, int n, void* out_f16, int32_t* out_keys);

/*
 * Write n rows exactly as given: rows_f16 [n,D] fp16 (host; must already be unit
 * length, nothing is normalised), keys[n] (FRHOT_KEY_DEAD allowed), destination
 * rows[n].  The inverse of frhot_gallery_read: what one returned, this restores bit
 * for bit.  Replaces: client.recover_snapshot of qdrant_database_FE/app.py:171-196
 * (the point data of a collection snapshot going back into the store).
 */
int frhot_gallery_write_f16(frhot_gallery* g, const void* rows_f16, const int32_t* keys,
                            const int64_t* rows, int n);

/* Number of rows the scan covers: 1 + the highest row ever written. */
int64_t frhot_gallery_rows(frhot_gallery* g);

/*
 * Exact cosine top-k of Bq queries against every live row whose key matches
 * qkeys[i] (FRHOT_KEY_ANY: all live rows; qkeys NULL: ANY for all; any other negative
 * key: FRHOT_EINVAL here and in frhot_recognise, an empty answer in the *_dev forms).  `q` is
 * [Bq,D] float32 on the host (normalised inside).  scores [Bq,k] float32
 * descending, rows [Bq,k] int64 (-1 and score -inf where fewer than k rows
 * match).  Ties in score are ordered by ascending row.
 * Replaces: client.search(exact=True, limit=5, store_id filter) of
 * qdrant_database_FE/app.py:316-337.
 */
int frhot_search(frhot_gallery* g, const float* q, const int32_t* qkeys, int Bq, int k,
                 float* scores, int64_t* rows);

/*
 * Device-pointer form: q_f16_dev [Bq,D] fp16 unit rows, qkeys_dev [Bq] int32 or
 * NULL, outputs scores_dev [Bq,k] float32 and rows_dev [Bq,k] int64 with
 * `row_base` added to every valid row (the shard's first global row).
 * Enqueues on the context stream and returns without waiting.
 */
int frhot_search_dev(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_dev,
                     int Bq, int k, void* scores_dev, void* rows_dev, int64_t row_base);

/*
 * frhot_search_dev with the queries' keys in HOST memory (the queries and the answers stay on the
 * device): the form a row-sharded search uses when its queries carry keys, because with the keys
 * on the host the shard is searched by key (see frhot_gallery_search_mode).  Enqueues only.
 * Keys as for frhot_search (other negative values than FRHOT_KEY_ANY: FRHOT_EINVAL); null = none.
 * Replaces: client.search on one collection (qdrant_database_FE/app.py:316-337) for one shard.
 */
int frhot_search_dev_keyed(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_host, int Bq, int k,
                           void* scores_dev, void* rows_dev, int64_t row_base);

/*
 * frhot_search_dev for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arguments, same answer bit for bit; the scan's per-block candidate lists and score
 * bounds live in a static device array of the library (about 5 MB on every GPU the library
 * runs on, held until the process ends) instead of buffers the gallery grows on demand, and
 * the call makes no hipMalloc / hipFree / allocating library call.  Limits: walkers x queries x
 * list length <= 327,680 entries and at most 8192 queries, i.e. up to 8192 queries of top-5,
 * 4096 of top-10, 2048 of top-16; beyond that FRHOT_ECAPACITY.  All calls on one GPU share
 * that GPU's array.  Calls on one context are serialised by its stream as usual; a call from
 * ANOTHER context of the process on the same GPU while the previous context's search is still
 * running is refused with FRHOT_ESTATE (its kernels would overlap on the array).  Not detected:
 * two processes never share the array (each loads its own copy), and a caller that runs one
 * context's work on several streams of its own must order them itself.
 */
int frhot_search_dev_static(frhot_gallery* g, const void* q_f16_dev, const int32_t* qkeys_dev,
                            int Bq, int k, void* scores_dev, void* rows_dev, int64_t row_base);

/*
 * Merge P per-shard answers (the all-gathered outputs of frhot_search_dev laid
 * out [P,Bq,k]) into the global top-k [Bq,k].  All pointers are device memory.
 * P <= 256; every per-shard list must be in the order frhot_search_dev writes
 * (score descending, row ascending inside equal scores, unused tail rows -1).
 * Replaces: nothing in the reference (one Qdrant node); it is the reduce step
 * of the row-sharded scan.
 */
int frhot_merge_topk_dev(frhot_ctx* ctx, const void* scores_dev, const void* rows_dev,
                         int P, int Bq, int k, void* out_scores_dev, void* out_rows_dev);

/*
 * The same merge reading the shards' answers where one all-gather leaves them: shard p's block
 * starts at packed_dev + p * shard_stride_bytes and holds its rows (int64 [Bq,k]) followed by its
 * scores (float32 [Bq,k]); the stride is a multiple of 8 bytes and at least 12 * Bq * k (blocks may
 * be padded).  A sharded search exchanges one such block per rank, so the merge needs no unpacking
 * copies.  Replaces: nothing in the reference (see frhot_merge_topk_dev).
 */
int frhot_merge_topk_packed_dev(frhot_ctx* ctx, const void* packed_dev, int64_t shard_stride_bytes,
                                int P, int Bq, int k, void* out_scores_dev, void* out_rows_dev);

/*
 * The row-sharded search as ONE call per rank, with the exchange inside the library (RCCL
 * over xGMI): for hosts that are not Python and have no torch.distributed.  One process
 * per GPU, one context per process; rank r's gallery holds the global rows
 * [row_base, row_base + its rows).  Replaces: nothing in the reference (one Qdrant node,
 * qdrant_database_FE/app.py:316-337 searches it whole); these are the gather / scan /
 * gather / merge steps of the search over a gallery too large or too hot for one GPU.
 *
 * frhot_dist_unique_id: 128 bytes that rank 0 makes and hands to every rank by any means
 *   of the host's own (a file, a socket, its job launcher).
 * frhot_dist_init: joins the context to a communicator of `world` ranks (collective: every
 *   rank calls it with the same id).  RCCL is loaded at this call (dlopen of the name in
 *   FRHOT_RCCL_LIB, else librccl.so.1 as the process already holds it or as the loader finds
 *   it); FRHOT_ESTATE if it cannot be loaded or the context has joined already.
 * frhot_dist_shutdown: leaves the communicator (collective); frhot_ctx_destroy does the
 *   same for a context that still holds one.
 * frhot_search_sharded_dev: every rank passes ITS B queries (fp16 unit rows on the device,
 *   and keys or NULL: all ranks pass keys or none does) and receives the global top-k of all
 *   world * B queries, rank-major: scores_dev [world*B, k] float32, rows_dev [world*B, k]
 *   int64 global rows.  Collective: same B and k on every rank.  Enqueues on the context
 *   stream (two all-gathers of a few hundred KB, the shard's scan between them, the merge)
 *   and returns without waiting.  The answer is the one frhot_search_dev gives on the whole
 *   gallery in one piece, bit for bit, ties included.  Its exchange buffers grow on demand
 *   at the call (hipMalloc), like a gallery's.  world = 1 runs the same steps.
 */
int frhot_dist_unique_id(void* id_out_128_bytes);
int frhot_dist_init(frhot_ctx* ctx, int rank, int world, const void* id_128_bytes);
int frhot_dist_shutdown(frhot_ctx* ctx);
int frhot_search_sharded_dev(frhot_gallery* g, const void* q_local_f16_dev, const int32_t* qkeys_local_dev,
                             int B, int k, void* scores_dev, void* rows_dev, int64_t row_base);

/*
 * The face detector of the reference's DeepFace.extract_faces(detector_backend="yolov8") calls
 * (app/src/utils/legacy.py:60-65 on the whole frame, :80-86 in detect_face): a YOLOv8n-pose network
 * with one class and five keypoints (the yolov8n-face.pt file of app/Dockerfile:82-84) on the
 * letterboxed frame (scaled to fit 640 x 640, padded to the next multiple of 32 in each direction,
 * as ultralytics does for a frame predicted on its own), then confidence selection and greedy IoU
 * suppression.  Frames of one call whose canvases agree run as one batch.
 * frhot_det_create: blob = the library's weight container ("FRHOTW01") holding, for every
 *   convolution of the network, "<name>.w" fp16 [cout][k][k][cin] with the batch norm folded in and
 *   "<name>.b" fp32 (names: frhot/detector.py, CONVS).  max_batch <= 64 frames per call; the
 *   activations take about 33 MB of device memory per frame.
 * frhot_det_detect: B BGR uint8 frames given like frhot_embed's crops.  For frame i, counts[i]
 *   faces (at most max_faces <= 300, descending confidence) are written to out[i][j][0..19] =
 *   x1, y1, x2, y2 in frame pixels (clipped to the frame), confidence, then five keypoints as
 *   (x, y, visibility): right eye, left eye, nose, right and left mouth corner in image order.
 *   conf: class score a face must exceed (the reference's backend uses 0.25); iou: boxes that
 *   overlap a better one by more than this are dropped (0.7).  Waits for the result.
 * frhot_det_debug_read: one named activation tensor of the last call's last group of frames
 *   ([B,h,w,c] fp16, h and w following that group's canvas), for tests.
 * Anti-spoofing (the `is_real` field DeepFace adds with a second network) is not part of this.
 */
typedef struct frhot_detector frhot_detector;
int frhot_det_create(frhot_ctx* ctx, const void* blob, size_t nbytes, int max_batch, frhot_detector** out);
int frhot_det_destroy(frhot_detector* det);
int frhot_det_detect(frhot_detector* det, const uint8_t* const* images, const int32_t* hw, const int32_t* strides,
                     int B, float conf, float iou, int max_faces, float* out, int32_t* counts);
int frhot_det_debug_read(frhot_detector* det, const char* name, int B, void* host_out, size_t nbytes);

/*
 * Fused embed -> search on one GPU: B host crops, qkeys[B] (or NULL), top-k
 * against gallery g of dimension D (4096, or < 4096 to match on the leading D
 * components re-normalised).  emb_out may be NULL.
 * Replaces: get_embedding + DatabaseClient.search_point round trip
 * (app/src/services/face_service.py:231-233,275-291).
 */
int frhot_recognise(frhot_ctx* ctx, frhot_gallery* g, const uint8_t* const* crops,
                    const int32_t* hw, const int32_t* strides, const int32_t* qkeys, int B,
                    unsigned flags, int k, float* emb_out, float* scores, int64_t* rows);

/*
 * The measure behind the check-in blur gate, for B crops given like frhot_embed's:
 * out[2i] = sum and out[2i+1] = sum of squares of the 3x3 Laplacian (up + down + left +
 * right - 4 centre, borders reflected without repeating the edge pixel) of crop i's grey
 * image (uint8, (3735 B + 19235 G + 9798 R + 16384) >> 15), both exact integers.  With
 * n = h * w the variance the gate compares is out[2i+1] / n - (out[2i] / n)^2; a crop
 * passes when that is >= the threshold (100 in the reference's settings).  The gamma LUT
 * is NOT applied: the reference measures the crop before adjust_gamma.
 * Replaces: check_detect_blur(img) (app/src/utils/legacy.py:125-129), i.e.
 * cv2.Laplacian(cv2.cvtColor(img, COLOR_BGR2GRAY), CV_64F).var(), called on the crop at
 * app/src/services/face_service.py:193.
 */
int frhot_crop_sharpness(frhot_ctx* ctx, const uint8_t* const* crops, const int32_t* hw,
                         const int32_t* strides, int B, int64_t* out);

/*
 * Same with the pixels already in HBM (arguments as frhot_embed_dev); out_dev is
 * [B,2] int64 in device memory.  Stream-ordered, does not synchronise.
 */
int frhot_crop_sharpness_dev(frhot_ctx* ctx, const void* crops_dev, const int64_t* offsets,
                             const int32_t* hw, const int32_t* strides, int B, void* out_dev);

/*
 * frhot_recognise that also returns frhot_crop_sharpness's pairs of the same crops
 * (sharpness [B,2], host) from the one upload: the caller drops the answers of the crops
 * that fail the gate.
 * Replaces: check_detect_blur + get_embedding + search_point of one check-in request
 * (app/src/services/face_service.py:193,212-219,231-233,275-291).
 */
int frhot_recognise_gated(frhot_ctx* ctx, frhot_gallery* g, const uint8_t* const* crops,
                          const int32_t* hw, const int32_t* strides, const int32_t* qkeys, int B,
                          unsigned flags, int k, float* emb_out, float* scores, int64_t* rows,
                          int64_t* sharpness);

/*
 * Stage timing with HIP events on the context stream.  frhot_ctx_timing(ctx, 1)
 * starts (and resets) recording: every later frhot_embed* / frhot_search* call
 * brackets its stages with event pairs (a ring of the last 128 calls) without
 * waiting.  frhot_ctx_stage_ms waits for the stream and returns the MEAN
 * milliseconds per call of each stage since recording started: 0 preprocess,
 * 1 conv stack, 2 fc6+fc7+norm, 3 gallery scan kernel, 4 top-k final selection.
 * No reference counterpart beyond the [TIMING] log lines of
 * app/src/services/face_service.py:229-236,286-296.
 */
int frhot_ctx_timing(frhot_ctx* ctx, int enable);
int frhot_ctx_stage_ms(frhot_ctx* ctx, float* ms5);

#ifdef __cplusplus
}
#endif
#endif /* FRHOT_H */
