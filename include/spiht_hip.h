/*
 * spiht_hip.h -- C ABI of libspiht_hip.so, the MI355X (gfx950) SPIHT hot path.
 *
 * This is the drop-in boundary: the entry points are what the reference's
 * extension module `spiht.spiht` (PyO3, /root/reference/src/lib.rs) exposes,
 * re-expressed with plain pointers and sizes, plus batched / fused forms so the
 * DWT output never has to leave HBM.  No torch types, no C++ types.
 *
 * Every function returns an int status (SPIHT_OK = 0) and never aborts.
 * Host pointers unless a parameter says "device".
 */
#ifndef SPIHT_HIP_H
#define SPIHT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SPIHT_OK = 0,
    SPIHT_ERR_LL = 1,        /* ll_h <= 1 || ll_w <= 1: the reference asserts (encoder_decoder.rs:160-161, 310-311) */
    SPIHT_ERR_EMPTY = 2,     /* empty array: the reference unwrap()s None (encoder_decoder.rs:165) */
    SPIHT_ERR_SHAPE = 3,     /* LL offspring fall outside [h,w]: the reference panics on an out-of-bounds index */
    SPIHT_ERR_CAPACITY = 4,  /* caller's output buffer too small; *out_nbits holds the size needed */
    SPIHT_ERR_HIP = 5,       /* a HIP runtime call failed; see spiht_last_hip_error() */
    SPIHT_ERR_ARG = 6,       /* null pointer / negative size / unknown wavelet or mode */
    SPIHT_ERR_MAGNITUDE = 7, /* max|x| >= 2^30: outside the range the reference handles */
    SPIHT_ERR_INTERNAL = 8,  /* device-side list overflow guard tripped (a bug, never expected) */
    SPIHT_ERR_TOO_LARGE = 9, /* c*h*w >= 2^30 or stream >= 2^32 bits */
    SPIHT_ERR_NOMEM = 10
};

/* signal extension modes of the DWT (pywt names) */
enum { SPIHT_MODE_REFLECT = 0, SPIHT_MODE_SYMMETRIC = 1, SPIHT_MODE_PERIODIC = 2, SPIHT_MODE_ZERO = 3,
       SPIHT_MODE_CONSTANT = 4,
       /* the modes whose extended samples are computed, not picked: a slower two-pass forward transform */
       SPIHT_MODE_SMOOTH = 5, SPIHT_MODE_ANTISYMMETRIC = 6, SPIHT_MODE_ANTIREFLECT = 7,
       /* another length rule: ceil(n / 2) coefficients per level, 2 n samples back (spiht_geometry_mode); two-pass in both
        * directions */
       SPIHT_MODE_PERIODIZATION = 8 };

#define SPIHT_MAX_BITS_UNLIMITED 0xFFFFFFFFFFFFFFFFull

typedef struct spiht_ctx spiht_ctx;

const char *spiht_strerror(int status);
const char *spiht_last_hip_error(void);
/* ABI version of this header; bumped on any signature change and on every round that adds entry points (now 2). */
int spiht_abi_version(void);

/* One context per (process, GPU): device id, streams, scratch.  Thread-safe per context
 * (calls on one context are serialised by an internal mutex); no globals survive a call. */
int spiht_ctx_create(int device, spiht_ctx **out);
/* priority > 0: the context's stream gets the device's highest stream priority (its workgroups are placed first when
 * kernels of several contexts wait for room on the CUs); 0: as spiht_ctx_create. */
int spiht_ctx_create_priority(int device, int priority, spiht_ctx **out);
void spiht_ctx_destroy(spiht_ctx *ctx);
/* Block until everything queued on the context's stream has finished. */
int spiht_ctx_synchronize(spiht_ctx *ctx);
/* Order across contexts without blocking the host: everything queued on `ctx` after this call waits (on the
 * device) for everything queued on `other` before it.  Lets one context's stream encode step i+1 while another
 * decodes step i. */
int spiht_ctx_wait_on(spiht_ctx *ctx, spiht_ctx *other);
/* Finer-grained ordering: an event marks the point a context's queue has reached (spiht_event_record); work queued on
 * another context after spiht_ctx_wait_event starts only once that point has been passed.  Neither call blocks the host. */
typedef struct spiht_event spiht_event;
int spiht_event_create(spiht_ctx *ctx, spiht_event **out);
void spiht_event_destroy(spiht_event *ev);
int spiht_event_record(spiht_event *ev, spiht_ctx *ctx);
int spiht_ctx_wait_event(spiht_ctx *ctx, spiht_event *ev);
/* Placement between two contexts' kernels without a timer (csrc/pipeline.cpp: the list decoder of a batch is launched
 * behind the persistent workgroups of the previous batch's inverse level 1).  The large levels of the inverse transform run
 * as persistent workgroups -- a fixed number per CU, each too large for one more than that number to fit -- that count
 * themselves in as they start.  _resident_ticket: where that count lives and what it will read once every such launch
 * queued on ctx SO FAR has all its workgroups on the CUs (*d_counter NULL: the context has no such counter).
 * _wait_resident: queues on ctx's stream a one-wavefront kernel that waits for the count to reach target, at most
 * timeout_us microseconds (<= 10 000; it never holds its stream for ever): what is queued behind it starts when those
 * workgroups are resident. */
int spiht_ctx_resident_ticket(spiht_ctx *ctx, const void **d_counter, uint32_t *target);
int spiht_ctx_wait_resident(spiht_ctx *ctx, const void *d_counter, uint32_t target, uint32_t timeout_us);
/* The context's HIP stream (*stream is a hipStream_t) so a caller can queue its own work -- e.g. the RCCL gather of
 * the streams between encode and decode -- in order with the library's. */
int spiht_ctx_stream(spiht_ctx *ctx, void **stream);
/* Stage timing: when enabled, HIP events bracket each kernel group of the next calls
 * (on the context's own stream); spiht_ctx_get_timing returns accumulated ms and launches. */
int spiht_ctx_set_timing(spiht_ctx *ctx, int enabled);
int spiht_ctx_reset_timing(spiht_ctx *ctx);
int spiht_ctx_num_stages(void);
const char *spiht_ctx_stage_name(int stage);
int spiht_ctx_get_timing(spiht_ctx *ctx, int stage, double *ms, uint64_t *launches);

/* ---------------------------------------------------------------------------------------
 * L2 boundary, single image, host buffers.
 * ------------------------------------------------------------------------------------- */

/* Replaces `encode(x, ll_h, ll_w, max_bits) -> (bytes, max_n)`  (src/lib.rs:24-32, which calls
 * encoder_decoder.rs:155-303 and packs bits LSB-first, lib.rs:29).
 * x: int32 [c,h,w] with element strides (any strides, as PyReadonlyArray3 allows).
 * max_bits: 0 behaves as "never reached" exactly like the reference (encoder_decoder.rs:196).
 * out/out_cap: caller-owned byte buffer; on SPIHT_ERR_CAPACITY *out_nbits is the bit count needed. */
int spiht_encode_i32(spiht_ctx *ctx, const int32_t *x, int64_t c, int64_t h, int64_t w, int64_t stride_c,
                     int64_t stride_h, int64_t stride_w, int64_t ll_h, int64_t ll_w, uint64_t max_bits,
                     uint8_t *out, uint64_t out_cap, uint64_t *out_nbits, uint8_t *max_n);

/* Upper bound, in bytes, of the stream spiht_encode_i32 can produce for this geometry when the largest
 * magnitude is max_abs (pass 0x3FFFFFFF if unknown).  Used by bindings to size `out`. */
int spiht_encode_bound(int64_t c, int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, uint32_t max_abs,
                       uint64_t max_bits, uint64_t *bound_bytes);

/* Replaces `decode(data_u8, n, c, h, w, ll_h, ll_w) -> ndarray[int32,(c,h,w)]`  (src/lib.rs:35-42 ->
 * encoder_decoder.rs:307-454).  All 8*nbytes bits are data (lib.rs:38).  out: c*h*w int32, C-contiguous,
 * fully written (zeros where nothing was decoded). */
int spiht_decode_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h,
                     int64_t w, int64_t ll_h, int64_t ll_w, int32_t *out);

/* Replaces `decode_with_metadata(data_u8, n, c, h, w, ll_h, ll_w, top_slice, other_slices)
 *           -> (ndarray[int32,(c,h,w)], ndarray[int32,(8*nbytes+1, 8)])`
 * (src/lib.rs:47-56 -> encoder_decoder.rs:631-841, Slices::from_vec :482-527).
 * top_slice: {start_i, end_i, start_j, end_j} of the LL block; other_slices: [level][3][4], coarsest level
 * first, per level the three filters in the caller's order (the reference wrapper passes da, ad, dd,
 * spiht_wrapper.py:240), each {start_i, end_i, start_j, end_j}.  level = other_slices.len().
 * meta: (8*nbytes + 1) rows of 8 int32 {action 0..6, local_h, local_w, channel, filter 0..3, depth, n, value of
 * the coefficient before the bit is read} (doc comment :616-630); row t describes the operation that reads
 * stream bit t, row 8*nbytes the operation left waiting when the stream ended, rows of bits that were never
 * reached are zero.
 * Errors: SPIHT_ERR_SHAPE when the tree is deeper than `level` (the reference indexes other_slices out of
 * bounds -> panic, :603), when a slice is empty or reversed (end <= start, usize underflow) or c*h*w >= 2^28. */
int spiht_decode_with_metadata_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c,
                                   int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, const int64_t *top_slice,
                                   const int64_t *other_slices, int64_t level, int32_t *out, int32_t *meta);

/* ---------------------------------------------------------------------------------------
 * Batched, device-resident forms (new; the reference codes one image per call).
 * All pointers below are DEVICE pointers (hipMalloc'd by anyone, e.g. a torch tensor's data_ptr()).
 * Work is queued on the context's stream; call spiht_ctx_synchronize() before reading results.
 * ------------------------------------------------------------------------------------- */

/* B images of identical geometry. d_x: int32 [B,c,h,w] contiguous.
 * d_out: B slots of slot_stride bytes (multiple of 4, >= ceil(min(max_bits,bound)/8) rounded up to 4);
 * bytes past ceil(nbits/8) in a slot are zeroed.  d_nbits: uint64 [B].  d_max_n: uint8 [B]. */
int spiht_encode_batch_i32(spiht_ctx *ctx, const int32_t *d_x, int64_t B, int64_t c, int64_t h, int64_t w,
                           int64_t ll_h, int64_t ll_w, uint64_t max_bits, uint8_t *d_out, uint64_t slot_stride,
                           uint64_t *d_nbits, uint8_t *d_max_n);

/* d_data: B slots of slot_stride bytes (multiple of 4); d_nbytes: uint64 [B] valid bytes per slot
 * (all 8*nbytes bits are data); d_max_n: uint8 [B]; d_out: int32 [B,c,h,w], fully written. */
int spiht_decode_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride, const uint64_t *d_nbytes,
                           const uint8_t *d_max_n, int64_t B, int64_t c, int64_t h, int64_t w, int64_t ll_h,
                           int64_t ll_w, int32_t *d_out);

/* ---------------------------------------------------------------------------------------
 * Fused image path: pixels -> DWT -> quantise -> SPIHT and back, everything in HBM.
 * Mirrors spiht_wrapper.encode_image / decode_image (wrapper:142-216) without colour conversion.
 * ------------------------------------------------------------------------------------- */

/* Geometry of the packed coefficient array for an H x W image (wrapper:92-139, pywt.wavedecn_shapes).
 * level < 0 means "None" (pywt's maximum useful level).  Any out pointer may be NULL. */
/* Every discrete wavelet of PyWavelets, 106 names (the reference hands SpihtSettings.wavelet to pywt as it is,
 * spiht_wrapper.py:163, :276): haar, db1-38, sym2-20, coif1-17, bior / rbio 1.1 ... 6.8, dmey (csrc/wavelets.h, generated
 * from PyWavelets by tools/gen_wavelets.py).  Filters of up to 20 taps run in the tiled level kernels; longer ones (up to
 * 102 taps) in the plain two-pass levels, their filters read from device memory.  Ids are positions in that table:
 * bior2.2 = 0, bior4.4 = 1, bior6.8 = 2, haar = 3. */
int spiht_wavelet_id(const char *name);           /* < 0 if unknown */
int spiht_wavelet_taps(int wavelet);              /* filter length (pywt dec_len); < 0: no such id */
int spiht_mode_id(const char *name);              /* the pywt names of the nine modes above; <0 if unknown */
int spiht_geometry(int64_t H, int64_t W, int wavelet, int level, int *level_used, int64_t *ll_h, int64_t *ll_w,
                   int64_t *enc_h, int64_t *enc_w, int64_t *rec_H, int64_t *rec_W);
/* ... for an extension mode: periodization has its own sizes (pywt.dwt_coeff_len), every other mode those above */
int spiht_geometry_mode(int64_t H, int64_t W, int wavelet, int mode, int level, int *level_used, int64_t *ll_h, int64_t *ll_w,
                        int64_t *enc_h, int64_t *enc_w, int64_t *rec_H, int64_t *rec_W);

/* d_img: float64 [B,c,H,W] (device).  channel_mults: HOST array of c doubles or NULL
 * (SpihtSettings.per_channel_quant_scales); q_scale: SpihtSettings.quantization_scale.
 * Outputs as spiht_encode_batch_i32.  d_coeffs: optional device int32 [B,c,enc_h,enc_w] that receives the
 * quantised coefficient array handed to the coder (NULL to keep it in scratch). */
int spiht_encode_image_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                 int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                 uint64_t max_bits, uint8_t *d_out, uint64_t slot_stride, uint64_t *d_nbits,
                                 uint8_t *d_max_n, int32_t *d_coeffs);

/* Inverse: streams -> float64 [B,c,rec_H,rec_W] (rec_H/rec_W from spiht_geometry; may exceed H/W by one on
 * odd axes exactly as pywt.waverec2 does).  d_rec: optional device int32 [B,c,enc_h,enc_w] receiving the
 * decoded coefficient array. */
int spiht_decode_image_batch_f64(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride,
                                 const uint64_t *d_nbytes, const uint8_t *d_max_n, int64_t B, int64_t c, int64_t H,
                                 int64_t W, int wavelet, int mode, int level, double q_scale,
                                 const double *channel_mults, double *d_img_out, int32_t *d_rec);

/* DWT halves on their own (device pointers), for parity tests and profiling:
 * forward: float64 [B,c,H,W] -> int32 [B,c,enc_h,enc_w] (quantised, zero padded)  (wrapper:163-172)
 * inverse: int32 [B,c,enc_h,enc_w] -> float64 [B,c,rec_H,rec_W]                   (wrapper:259-276) */
int spiht_dwt_quant_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                              int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                              int32_t *d_coeffs);
int spiht_dequant_idwt_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, int64_t B, int64_t c, int64_t H, int64_t W,
                                 int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                 double *d_img_out);

/* Single-precision forms of the two encode-side entry points: d_img float32 [B,c,H,W].  PyWavelets transforms
 * float32 (and float16) pixels in float32 and the reference wrapper quantises the float32 array in float32
 * (`pywt.wavedec2` dtype rule, spiht_wrapper.py:163-172) -- so a float32 image does not give the stream its float64
 * copy gives; these reproduce the float32 result, including pywt's order of additions at the right / bottom edge.
 * (Levels above pywt's dwt_max_level -- inputs shorter than the filter -- are coded as PyWavelets codes them, in either
 * precision: it only warns, spiht_wrapper.py:163.)  SPIHT_ERR_ARG for level == 0.
 * The decode side is float64 in the reference whatever the pixels were. */
int spiht_dwt_quant_batch_f32(spiht_ctx *ctx, const float *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                              int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                              int32_t *d_coeffs);
int spiht_encode_image_batch_f32(spiht_ctx *ctx, const float *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                 int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                 uint64_t max_bits, uint8_t *d_out, uint64_t slot_stride, uint64_t *d_nbits,
                                 uint8_t *d_max_n, int32_t *d_coeffs);

/* Significance pyramid on its own (device pointers), for parity tests and profiling:
 * d_x int32 [B,c,h,w] -> d_dmsb, d_lmsb uint8 [B,c,h,w] (1 + msb of the D / L set maxima of the node with
 * that index, 0 = empty/zero; only nodes with offspring are written) and d_maxabs uint32 [B] (NULL: not computed). */
int spiht_pyramid_batch_i32(spiht_ctx *ctx, const int32_t *d_x, int64_t B, int64_t c, int64_t h, int64_t w,
                            int64_t ll_h, int64_t ll_w, uint8_t *d_dmsb, uint8_t *d_lmsb, uint32_t *d_maxabs);

/* The two halves of each direction on their own (device pointers, asynchronous).  The transform + pyramid half is
 * HBM-bound, the list-coding half is latency-bound and leaves the HBM idle, so a caller that keeps several batches in
 * flight runs them on two contexts ordered with events (bench.py):
 *   spiht_dwt_pyramid_batch_f64   pixels -> coefficients + D/L pyramid + max|coef|   (front half of encode_image);
 *                                 with d_dmsb = d_lmsb = NULL the pyramid is left out and can be queued on another
 *                                 context with spiht_pyramid_batch_i32(..., d_maxabs = NULL) (NULL: max|coef| is
 *                                 already known, no pass over the coefficients for it)
 *   spiht_encode_lists_batch_i32  coefficients + pyramid -> streams                  (back half; = k_encode)
 *   spiht_decode_lists_batch_i32  streams -> coefficients; d_out must be ZERO-FILLED by the caller (spiht_dev_memset)
 *   spiht_dequant_idwt_batch_f64  coefficients -> pixels                             (above)
 *   spiht_unscatter_lists_batch_i32  after the inverse transform has read d_out: puts back the zeros in exactly the
 *                                 cells the context's last spiht_decode_lists_batch_i32 wrote (through the decoder's
 *                                 lists), so the array is zero again for the next decode without a full zero-fill;
 *                                 falls back to the zero-fill when another list-coding call on the context came in
 *                                 between.  (The reference allocates a fresh zero array per call,
 *                                 encoder_decoder.rs:308; this keeps one array zero instead.)
 * Results are identical to the fused entry points. */
int spiht_dwt_pyramid_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                int32_t *d_coeffs, uint8_t *d_dmsb, uint8_t *d_lmsb, uint32_t *d_maxabs);
int spiht_encode_lists_batch_i32(spiht_ctx *ctx, const int32_t *d_x, const uint8_t *d_dmsb, const uint8_t *d_lmsb,
                                 const uint32_t *d_maxabs, int64_t B, int64_t c, int64_t h, int64_t w, int64_t ll_h,
                                 int64_t ll_w, uint64_t max_bits, uint8_t *d_out, uint64_t slot_stride,
                                 uint64_t *d_nbits, uint8_t *d_max_n);
int spiht_decode_lists_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride, const uint64_t *d_nbytes,
                                 const uint8_t *d_max_n, int64_t B, int64_t c, int64_t h, int64_t w, int64_t ll_h,
                                 int64_t ll_w, int32_t *d_out_zeroed);
int spiht_unscatter_lists_batch_i32(spiht_ctx *ctx, int32_t *d_out, int64_t B, int64_t c, int64_t h, int64_t w);
/* Occupancy of the inverse transform's level-1 tiles, handed from the list decoder to the inverse transform.  The
 * reference's decoder fills a dense array and waverec2 reads all of it (spiht_wrapper.py:248-276); at the metric's bit
 * rates nearly every cell of the level-1 detail bands -- three quarters of the array -- is zero, and the decoder knows
 * every cell it writes: it sets one 32-bit word per (plane, inverse-transform tile) whose staged band region holds a
 * decoded cell, and level 1 of the inverse transform does not read the detail bands of a tile whose word is zero (zeros go
 * through the same arithmetic: the same bits).  The image-level decode calls do this internally (option "l1_flags");
 * the split calls take the words explicitly:
 *   spiht_l1_flags_words                 words per image for this geometry (0: not applicable -- fewer than two levels,
 *                                        periodization)
 *   spiht_decode_lists_flags_batch_i32   spiht_decode_lists_batch_i32 for the arrays of H x W images + the words
 *                                        (zero-filled by the call; d_flags NULL: no flags)
 *   spiht_dequant_idwt_flags_batch_f64   spiht_dequant_idwt_batch_f64 reading them (d_flags NULL: reads everything) */
int spiht_l1_flags_words(int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level, uint64_t *words_per_image);
int spiht_decode_lists_flags_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride, const uint64_t *d_nbytes,
                                       const uint8_t *d_max_n, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet,
                                       int mode, int level, int32_t *d_out_zeroed, uint32_t *d_flags);
int spiht_dequant_idwt_flags_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const uint32_t *d_flags, int64_t B, int64_t c,
                                       int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                       const double *channel_mults, double *d_img_out);
/* Switches of this library's own making; the results are the same bits whatever they are set to.  "l1_flags" (default 1):
 * the occupancy words above inside the image-level decode calls; "pads_persist" (default 0): the caller promises that a
 * coefficient array this context's forward transform has filled is not written by anyone else before the same context
 * fills it again with the same geometry -- the zero padding of coeffs_to_array is then written once per array instead
 * of once per call (a caller that recycles its arrays, e.g. the pipelined schedule).  Those two: value 0 / 1.
 * "wide_encode" (default 1): an encode call of few images (at most half as many as the device has CUs, each of 2^18
 * coefficients or more) codes each image on a group of workgroups, one per CU, instead of one workgroup -- the latency of a
 * single call (csrc/encode_wide.hip); 0: always one workgroup per image; 2: groups whatever the size of the image
 * (tests); 3: as 2, and every group finds itself "given up" (tests of the fallback below).  "wide_groups" (default 0 = by
 * image size, 2 ... 64): workgroups per image of that path, 0 ... 256; "wide_solo" (default 24576): list entries up to
 * which a bit plane is still coded by the group's first workgroup alone.  The workgroups of a group wait for one another
 * and should all be resident: the library keeps a launch within what the device holds of that kernel and queues such
 * launches of one process one at a time per device; when other work holds the CUs all the same (another process, a long
 * kernel of the caller's), a group gives up after a bounded wait (tens of milliseconds) and the image is coded by one
 * workgroup instead, queued behind on the same stream -- same bits, never an error (spiht_ctx_wide_stats tells).
 * "idwt_groups" (default 0 = 4): persistent workgroups per CU of the large inverse-transform levels; 3 leaves a list
 * decoder's workgroup room beside them (the pipelined schedule sets it around its own calls). */
int spiht_ctx_set_option(spiht_ctx *ctx, const char *name, int64_t value);
int spiht_ctx_get_option(spiht_ctx *ctx, const char *name, int64_t *value);
/* The last encode call of this context that took the several-CUs-per-image path: its images (groups) and how many of them
 * gave up for lack of residency and were coded by the single-workgroup kernel instead.  Waits for the context's stream. */
int spiht_ctx_wide_stats(spiht_ctx *ctx, uint32_t *groups, uint32_t *gave_up);

/* The inverse transform (spiht_dequant_idwt_batch_f64) in two parts, for the same kind of schedule: the coarse levels
 * (level .. 2: a quarter of the bytes) into d_approx [B*c, 2*hs[2]-F+2, 2*ws[2]-F+2] float64 -- the approximation level 1
 * starts from; spiht_idwt_approx_shape gives its size -- and level 1 from d_rec + d_approx to the pixels.  With fewer than
 * two levels the coarse call does nothing and level 1 does everything.  Same bits as the one-call form. */
int spiht_idwt_coarse_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, int64_t B, int64_t c, int64_t H, int64_t W,
                                int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                double *d_approx);
int spiht_idwt_level1_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const double *d_approx, int64_t B, int64_t c,
                                int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                const double *channel_mults, double *d_img_out);
int spiht_idwt_approx_shape(int64_t H, int64_t W, int wavelet, int level, int64_t *a_h, int64_t *a_w);
/* ... level 1 reading the decoder's occupancy words (spiht_decode_lists_flags_batch_i32; d_flags NULL: reads everything) */
int spiht_idwt_level1_flags_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const double *d_approx, const uint32_t *d_flags,
                                      int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                      const double *channel_mults, double *d_img_out);

/* Colour model change on the device (the reference converts on the host through colour-science, color_models.py:6-13,
 * called from spiht_wrapper.py:158-160 and :278-279): B three-channel float64 images [B,3,npix]; per pixel
 * w = M * spow(A * u, p) with spow(x, p) = sign(x)|x|^p -- RGB -> IPT is (XYZ->LMS * RGB->XYZ, 0.43, LMS->IPT), the way
 * back the inverses with 1/0.43.  A, M: row-major 3x3 host arrays.  d_out may equal d_in.  Asynchronous. */
int spiht_color3_batch_f64(spiht_ctx *ctx, const double *d_in, double *d_out, int64_t B, int64_t npix, const double *A,
                           const double *M, double p);
/* The colour model of the coded picture as a property of the context (spiht_wrapper.py:158-160, :278-279: RGB -> model
 * before the transform, model -> RGB after the inverse transform), fused into the transform: while set, every image-level
 * entry point of the context takes and returns RGB pixels of 3-channel float64 images and codes them in the other model --
 * the forward level-1 kernel converts on its loads (w = M_f * spow(A_f * u, p_f)), the inverse level-1 kernel on its
 * stores (A_i, M_i, p_i); the converted picture never exists in memory.  Same bits as spiht_color3_batch_f64 followed by
 * the plain transform.  A_f == NULL clears the setting.  Images with other channel counts are coded as they are. */
int spiht_ctx_set_color3(spiht_ctx *ctx, const double *A_f, const double *M_f, double p_f, const double *A_i,
                         const double *M_i, double p_i);
/* ... and what is set at the moment (*on == 0: nothing, the arrays are left alone; output pointers other than `on` may be
 * NULL) -- for code that borrows a context and puts the caller's setting back (csrc/pipeline.cpp). */
int spiht_ctx_get_color3(spiht_ctx *ctx, int *on, double *A_f, double *M_f, double *p_f, double *A_i, double *M_i, double *p_i);

/* Progressive decoding of one stream to K bit budgets from ONE walk (the reference decodes a prefix per frame,
 * make_gif.py:46-61: `decode(original_bytes[:byte_len], ...)`, i.e. K walks).  budgets_bits: host array,
 * ascending (else SPIHT_ERR_ARG); out[k] = what spiht_decode_i32 returns for the first budgets_bits[k] bits of the
 * stream (8 x a byte length for a byte prefix; a budget past the end means the whole stream) -- bit-exact, duplicated
 * tree nodes included.  The walk records which list entry every stream position belongs to (as decode_with_metadata,
 * src/encoder_decoder.rs:631-841), the records are sorted by node, and one thread per node replays the node's operations
 * once, leaving its value in every budget it passes.
 *   spiht_decode_budgets_i32      out: host int32 [K, c, h, w]; synchronous
 *   spiht_decode_budgets_dev_i32  d_out: device int32 [K, c, h, w] (zero-filled by the call); asynchronous apart from
 *                                 the upload of the stream (feeds spiht_dequant_idwt_batch_f64 with B = K) */
int spiht_decode_budgets_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h, int64_t w,
                             int64_t ll_h, int64_t ll_w, const uint64_t *budgets_bits, int64_t K, int32_t *out);
int spiht_decode_budgets_dev_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h,
                                 int64_t w, int64_t ll_h, int64_t ll_w, const uint64_t *budgets_bits, int64_t K, int32_t *d_out);

/* The context's (recursive) mutex held across a SEQUENCE of calls by one thread: spiht_ctx_set_color3 is state of the
 * context, so set / image calls / clear must not interleave with another thread's calls on the same context (those
 * would be coded in the wrong colour model, silently).  spiht_amd.color_models.fused brackets its block with these.
 * Unlock from the thread that locked.  (The reference's colour step is a pure function of the pixel array,
 * spiht_wrapper.py:158-160 -- there is no shared state to protect there.) */
int spiht_ctx_lock(spiht_ctx *ctx);
int spiht_ctx_unlock(spiht_ctx *ctx);

/* Wavefronts per workgroup of the list decoder on this context: 12 (default; the shortest walk of one stream) or 8
 * (4 % slower alone, a lighter neighbour for HBM-bound kernels running beside it on other contexts -- what the
 * pipelined schedule uses for its list-coding contexts).  Output identical.  Other values: SPIHT_ERR_ARG.
 * Replaces nothing in the reference (src/encoder_decoder.rs:307-454 is one thread); a scheduling knob of this library. */
int spiht_ctx_set_decoder_waves(spiht_ctx *ctx, int waves);

/* d_nbytes[b] = ceil(d_nbits[b] / 8) for b < B (device arrays): turns the encoder's bit counts into the byte
 * counts the decoder takes, without a host round trip. */
int spiht_nbits_to_nbytes(spiht_ctx *ctx, const uint64_t *d_nbits, int64_t B, uint64_t *d_nbytes);

/* The drop-in calls on HOST arrays, one C call each (what spiht_amd.encode_image / decode_image / decode_from_rec_arr
 * bind): replace encode_image (spiht_wrapper.py:142-189: wavedec2 -> coeffs_to_array -> channel scales -> quantize ->
 * spiht.encode), decode_image (:192-216, :259-276: spiht.decode -> dequantize -> waverec2) and decode_from_rec_arr
 * (:259-276) without the colour step.  img: [c,H,W] C-contiguous; out / out_cap as in spiht_encode_i32; img_out:
 * float64 [c, rec_H, rec_W] (spiht_geometry).  Pixels, coefficients and stream stay in the context's grow-only device
 * buffers: no device allocation per call once a size has been seen.  Synchronous. */
int spiht_encode_image_host_f64(spiht_ctx *ctx, const double *img, int64_t c, int64_t H, int64_t W, int wavelet,
                                int mode, int level, double q_scale, const double *channel_mults, uint64_t max_bits,
                                uint8_t *out, uint64_t out_cap, uint64_t *out_nbits, uint8_t *max_n);
int spiht_encode_image_host_f32(spiht_ctx *ctx, const float *img, int64_t c, int64_t H, int64_t W, int wavelet,
                                int mode, int level, double q_scale, const double *channel_mults, uint64_t max_bits,
                                uint8_t *out, uint64_t out_cap, uint64_t *out_nbits, uint8_t *max_n);
int spiht_decode_image_host_f64(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t H,
                                int64_t W, int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                double *img_out);
int spiht_dequant_idwt_host_f64(spiht_ctx *ctx, const int32_t *rec, int64_t c, int64_t H, int64_t W, int wavelet, int mode,
                                int level, double q_scale, const double *channel_mults, double *img_out);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, every rank codes its own images (the reference's encode / decode
 * are pure functions of one image, src/lib.rs:24-42 -- nothing is exchanged while coding); the ONE exchange of the
 * path is the gather of the finished streams.  It runs on RCCL (ncclAllGather over xGMI), inside this library, on
 * the context's stream; librccl is loaded on first use.
 * ------------------------------------------------------------------------------------- */
typedef struct spiht_comm spiht_comm;
#define SPIHT_COMM_ID_BYTES 128
/* Rank 0 makes the job's id (ncclGetUniqueId) and hands its 128 bytes to every rank by any host channel
 * (spiht_amd/dist.py: a TCP exchange on MASTER_ADDR). */
int spiht_comm_unique_id(uint8_t *id128);
/* Collective over all ranks: joins the communicator of `world` ranks as `rank`, on the context's GPU. */
int spiht_comm_create(spiht_ctx *ctx, const uint8_t *id128, int world, int rank, spiht_comm **out);
void spiht_comm_destroy(spiht_comm *comm);
int spiht_comm_info(spiht_comm *comm, int *world, int *rank, int *rccl_version);
/* All-gather of B stream slots (slot_stride bytes each), bit counts and start planes per rank: rank r's rows land in
 * rows [r*B, (r+1)*B) of d_all_* (sizes world*B) on every rank.  Device pointers; queued on the context's stream
 * after the encoder kernels queued before it -- the host does not block (spiht_ctx_synchronize to wait). */
int spiht_gather_streams(spiht_ctx *ctx, spiht_comm *comm, const uint8_t *d_slots, const uint64_t *d_nbits,
                         const uint8_t *d_max_n, int64_t B, uint64_t slot_stride, uint8_t *d_all_slots,
                         uint64_t *d_all_nbits, uint8_t *d_all_max_n);
/* The RCCL shared library this process uses: the soname that was loaded, or every candidate tried with the loader's
 * reason when none could be ("" before the first spiht_comm_* call).  For the job's log line. */
const char *spiht_rccl_library(void);
/* spiht_pipeline_submit with the gather of a multi-GPU job in it (see the pipeline section below) */
typedef struct spiht_pipeline spiht_pipeline;
int spiht_pipeline_submit_gather(spiht_pipeline *p, const double *d_img, uint8_t *d_out, uint64_t *d_nbits, uint8_t *d_max_n,
                                 double *d_img_out, spiht_comm *comm, uint8_t *d_all_slots, uint64_t *d_all_nbits,
                                 uint8_t *d_all_max_n, int rank);
/* Where rank r's rows start in the gathered arrays, in bytes from each array's start (rank-major: rows [r*B, (r+1)*B)).
 * Pure arithmetic, no device: callable anywhere. */
int spiht_gather_row_offsets(int rank, int world, int64_t B, uint64_t slot_stride, uint64_t *off_slots, uint64_t *off_nbits,
                             uint64_t *off_max_n);
/* Host-side job control over the same communicator (both block): every rank has arrived and its context's queue is
 * empty; *value becomes the maximum over ranks. */
int spiht_comm_barrier(spiht_ctx *ctx, spiht_comm *comm);
int spiht_comm_allreduce_max_f64(spiht_ctx *ctx, spiht_comm *comm, double *value);

/* ---------------------------------------------------------------------------------------
 * The pipelined round trip (csrc/pipeline.cpp): B images per step, pixels -> streams -> pixels, consecutive steps
 * software-pipelined over three contexts the pipeline owns -- the HBM-bound passes of steps i+1 / i-1 on one, the list
 * coder of step i on another -- ordered with events; no call blocks the host except _synchronize.  This is the schedule
 * the throughput metric is measured on (bench.py); it replaces a caller's loop over the reference's one-image calls
 * (spiht_wrapper.py:142-216).  All image / stream pointers are DEVICE pointers as in the batched calls above and must stay
 * valid until the step that uses them has completed (d_out / d_nbits / d_max_n may be the same for every step: a step's
 * decoder has read them before the next step's encoder writes them).
 *   create        geometry and settings as spiht_encode_image_batch_f64; max_bits 0 = unlimited
 *   info          stream slot size in bytes (spiht_encode_bound) and the decoded picture size
 *   set_color3    colour model of the coded picture (as spiht_ctx_set_color3) for every step; kept by the pipeline
 *   submit        queue one step; the decoded pictures of step i are complete after submit of step i+1 has been followed
 *                 by _synchronize -- or after _flush + waiting
 *   submit_gather the same in a multi-GPU job: the streams are all-gathered (spiht_gather_streams) between encoder and
 *                 decoder on the list-coding stream, and the decoder reads rows [rank*B, (rank+1)*B) of the gathered buffers
 *   flush         queue the inverse transform of the last submitted step
 *   synchronize   flush, then wait for everything; returns the first latched device error
 *   contexts      the three contexts (stage timing: spiht_ctx_set_timing / _get_timing) */
typedef struct spiht_pipeline spiht_pipeline;
int spiht_pipeline_create(int device, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level,
                          double q_scale, const double *channel_mults, uint64_t max_bits, spiht_pipeline **out);
/* ... the HBM-bound passes on the caller's context h_ctx (on `device`; not destroyed with the pipeline): a process has few
 * hardware queues for its HIP streams, a caller that holds a context already should not add a fourth stream.  Nothing of
 * the pipeline's stays on h_ctx between calls: while submit / flush queue work they hold its mutex and have the pipeline's
 * colour model and options on it, and put back what the caller had set.  A call that fails half-way leaves the pipeline
 * unusable: every later call returns that first error (destroy it). */
int spiht_pipeline_create_on(spiht_ctx *h_ctx, int device, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode,
                             int level, double q_scale, const double *channel_mults, uint64_t max_bits, spiht_pipeline **out);
void spiht_pipeline_destroy(spiht_pipeline *p);
int spiht_pipeline_info(spiht_pipeline *p, uint64_t *slot_stride, int64_t *rec_H, int64_t *rec_W);
int spiht_pipeline_set_color3(spiht_pipeline *p, const double *A_f, const double *M_f, double p_f, const double *A_i,
                              const double *M_i, double p_i);
int spiht_pipeline_submit(spiht_pipeline *p, const double *d_img, uint8_t *d_out, uint64_t *d_nbits, uint8_t *d_max_n,
                          double *d_img_out);
int spiht_pipeline_flush(spiht_pipeline *p);
int spiht_pipeline_synchronize(spiht_pipeline *p);
int spiht_pipeline_contexts(spiht_pipeline *p, spiht_ctx **h, spiht_ctx **l0, spiht_ctx **l1);

/* Thin device-memory helpers so a host language without a HIP binding can drive the batched API. */
/* Page-locked host memory for the arrays the host-array calls RETURN (decode_image gives back a new array,
 * spiht_wrapper.py:192-216): a device -> host copy into it is one DMA at the link's speed, into fresh pageable memory
 * several times slower (page faults + staging).  Pooled inside the library (pinning is what costs); at most 4 GiB are
 * handed out -- beyond that, or when the system refuses, SPIHT_ERR_NOMEM and the caller takes ordinary memory: every
 * host-array call accepts any host pointer.  Not tied to a context or device. */
int spiht_host_alloc(uint64_t bytes, void **h_ptr);
int spiht_host_free(void *h_ptr);
int spiht_dev_alloc(spiht_ctx *ctx, uint64_t bytes, void **d_ptr);
int spiht_dev_free(spiht_ctx *ctx, void *d_ptr);
int spiht_dev_upload(spiht_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int spiht_dev_download(spiht_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int spiht_dev_memset(spiht_ctx *ctx, void *d_dst, int value, uint64_t bytes);
int spiht_dev_copy(spiht_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes); /* device to device, asynchronous */

#ifdef __cplusplus
}
#endif
#endif /* SPIHT_HIP_H */
