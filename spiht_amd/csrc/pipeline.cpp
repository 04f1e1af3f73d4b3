// The pipelined round-trip schedule behind the C ABI (include/spiht_hip.h: spiht_pipeline_*): the round trip of a sequence
// of batches, for a caller in any language (Python: spiht_amd.batch.Pipeline).  Written against the public entry points only.
//
// The transform / pyramid / inverse-transform passes are HBM-bound, the list coder is latency-bound and leaves the HBM
// idle, so consecutive batches are software-pipelined over three contexts ordered with events (the host never blocks):
//
//     H :  A(i)                [X(i-1) done] I(i-1)  U(i-1)*   A(i+1)               [X(i) done] I(i) ...
//     Ls:  [A(i), X(i-1) done] E(i) (gather) X(i)              [A(i+1), X(i) done] E(i+1) X(i+1) ...
//
// A = DWT + quantise + pyramid, E / X = encoder / decoder list kernels, I = dequantise + inverse DWT, U = the zeros put
// back into the coefficient array X scattered into (on that batch's list-coding context, right behind I).  Everything
// the schedule needs between the stages -- coefficient arrays, pyramid, decoder output, occupancy words -- lives in two
// buffer sets owned by the pipeline; the two list-coding contexts (s = i & 1) each keep the decoder lists of their set.
// The reference codes one image per call on the CPU (spiht_wrapper.py:142-216); this replaces a loop over such calls.
#include "../../include/spiht_hip.h"

#ifndef PIPE_DEC_WAVES
#define PIPE_DEC_WAVES 8  // the lighter neighbour for the HBM-bound kernels (decode.hip); -DPIPE_DEC_WAVES=12 for A/B runs
#endif

#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#define CHK(expr)                      \
    do {                               \
        int _s = (expr);               \
        if (_s != SPIHT_OK) return _s; \
    } while (0)

struct spiht_pipeline {
    int device = 0;
    int64_t B = 0, c = 0, H = 0, W = 0;
    int wavelet = 0, mode = 0, level = -1;
    double q = 1.0;
    std::vector<double> mults;
    uint64_t max_bits = 0, slot_stride = 0, flag_words = 0;
    int64_t ll_h = 0, ll_w = 0, enc_h = 0, enc_w = 0;
    spiht_ctx *Hc = nullptr, *Lc[2] = {nullptr, nullptr};
    bool owns_h = true;   // false: the caller's context (spiht_pipeline_create_on)
    spiht_event *ev_a[2] = {nullptr, nullptr}, *ev_d[2] = {nullptr, nullptr}, *ev_i[2] = {nullptr, nullptr};
    spiht_event *ev_c = nullptr;  // the coarse levels of the pending inverse transform are through
    double *approx = nullptr;     // what they leave for its level 1 (spiht_idwt_approx_shape)
    // two buffer sets
    int32_t *coeffs[2] = {nullptr, nullptr}, *rec[2] = {nullptr, nullptr};
    uint8_t *dmsb[2] = {nullptr, nullptr}, *lmsb[2] = {nullptr, nullptr};
    uint32_t *maxabs[2] = {nullptr, nullptr}, *flags[2] = {nullptr, nullptr};
    uint64_t *nbytes[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    uint64_t step = 0;
    bool pending = false;      // a batch whose inverse transform has not been queued yet
    int pending_slot = 0;
    double *pending_out = nullptr;
    // colour model of the coded pictures (spiht_pipeline_set_color3): the pipeline's own, put on the H context around its calls
    bool color_on = false;
    double cAf[9], cMf[9], cAi[9], cMi[9], cpf = 1.0, cpi = 1.0;
    int poisoned = SPIHT_OK;   // a call failed half-way: the schedule's state is void, every later call returns this
    const double *mp() const { return mults.empty() ? nullptr : mults.data(); }
};

// The H context may be the caller's (spiht_pipeline_create_on): nothing of the pipeline's stays on it between calls.  While a
// call of the pipeline queues work it holds the context's mutex and has its own settings on it -- colour model, zero padding
// written once per array (the coefficient arrays are the pipeline's own and only its forward transform writes them), three
// persistent inverse-transform workgroups per CU (a decoder workgroup that arrives behind them still fits: registers) -- and
// what the caller had there before comes back when the call returns.
struct HScope {
    spiht_pipeline *p;
    int64_t old_pads = 0, old_groups = 0;
    int old_on = 0;
    double oAf[9], oMf[9], oAi[9], oMi[9], opf = 1.0, opi = 1.0;
    int st = SPIHT_OK;
    explicit HScope(spiht_pipeline *pp) : p(pp) {
        st = spiht_ctx_lock(p->Hc);
        if (st != SPIHT_OK) { p = nullptr; return; }
        (void)spiht_ctx_get_option(p->Hc, "pads_persist", &old_pads);
        (void)spiht_ctx_get_option(p->Hc, "idwt_groups", &old_groups);
        (void)spiht_ctx_get_color3(p->Hc, &old_on, oAf, oMf, &opf, oAi, oMi, &opi);
        (void)spiht_ctx_set_option(p->Hc, "pads_persist", 1);
        (void)spiht_ctx_set_option(p->Hc, "idwt_groups", 3);
        if (p->color_on) st = spiht_ctx_set_color3(p->Hc, p->cAf, p->cMf, p->cpf, p->cAi, p->cMi, p->cpi);
        else st = spiht_ctx_set_color3(p->Hc, nullptr, nullptr, 1.0, nullptr, nullptr, 1.0);
    }
    ~HScope() {
        if (!p) return;
        (void)spiht_ctx_set_option(p->Hc, "pads_persist", old_pads);
        (void)spiht_ctx_set_option(p->Hc, "idwt_groups", old_groups);
        if (old_on) (void)spiht_ctx_set_color3(p->Hc, oAf, oMf, opf, oAi, oMi, opi);
        else (void)spiht_ctx_set_color3(p->Hc, nullptr, nullptr, 1.0, nullptr, nullptr, 1.0);
        (void)spiht_ctx_unlock(p->Hc);
    }
};

static void pipeline_free(spiht_pipeline *p) {
    if (!p) return;
    if (p->Hc) {
        (void)spiht_ctx_synchronize(p->Hc);
        for (int s = 0; s < 2; s++) {
            if (p->Lc[s]) (void)spiht_ctx_synchronize(p->Lc[s]);
            void *bufs[] = {p->coeffs[s], p->rec[s], p->dmsb[s], p->lmsb[s], p->maxabs[s], p->flags[s], p->nbytes[s]};
            for (void *b : bufs)
                if (b) (void)spiht_dev_free(p->Hc, b);
            spiht_event_destroy(p->ev_a[s]);
            spiht_event_destroy(p->ev_d[s]);
            spiht_event_destroy(p->ev_i[s]);
        }
    }
    if (p->Hc && p->approx) (void)spiht_dev_free(p->Hc, p->approx);
    spiht_event_destroy(p->ev_c);
    for (int s = 0; s < 2; s++)
        if (p->Lc[s]) spiht_ctx_destroy(p->Lc[s]);
    if (p->Hc && p->owns_h) spiht_ctx_destroy(p->Hc);
    delete p;
}

static int pipeline_create(spiht_ctx *h_ctx, int device, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level,
                           double q_scale, const double *channel_mults, uint64_t max_bits, spiht_pipeline **out) {
    if (!out || B < 1 || c < 1 || H < 1 || W < 1) return SPIHT_ERR_ARG;
    *out = nullptr;
    spiht_pipeline *p = new (std::nothrow) spiht_pipeline();
    if (!p) return SPIHT_ERR_NOMEM;
    p->Hc = h_ctx;
    p->owns_h = h_ctx == nullptr;
    p->device = device; p->B = B; p->c = c; p->H = H; p->W = W;
    p->wavelet = wavelet; p->mode = mode; p->level = level; p->q = q_scale;
    if (channel_mults) p->mults.assign(channel_mults, channel_mults + c);
    p->max_bits = max_bits == 0 ? SPIHT_MAX_BITS_UNLIMITED : max_bits;
    int st = spiht_geometry_mode(H, W, wavelet, mode, level, nullptr, &p->ll_h, &p->ll_w, &p->enc_h, &p->enc_w, nullptr, nullptr);
    if (st == SPIHT_OK) st = spiht_encode_bound(c, p->enc_h, p->enc_w, p->ll_h, p->ll_w, 0x3FFFFFFFu, p->max_bits, &p->slot_stride);
    if (st == SPIHT_OK) st = spiht_l1_flags_words(c, H, W, wavelet, mode, level, &p->flag_words);
    if (st != SPIHT_OK) { pipeline_free(p); return st; }
    if (p->slot_stride < 4) p->slot_stride = 4;
    if (!p->Hc && (st = spiht_ctx_create(device, &p->Hc)) != SPIHT_OK) { pipeline_free(p); return st; }
    const uint64_t n = (uint64_t)c * p->enc_h * p->enc_w;
    for (int s = 0; s < 2 && st == SPIHT_OK; s++) {
        if ((st = spiht_ctx_create(device, &p->Lc[s])) != SPIHT_OK) break;
        // decoder workgroups of 8 wavefronts: a longer walk, a lighter neighbour for the transforms beside it
        if ((st = spiht_ctx_set_decoder_waves(p->Lc[s], PIPE_DEC_WAVES)) != SPIHT_OK) break;
        // one workgroup per image whatever the batch size: the several-CUs-per-image encoder is for single calls -- its
        // workgroups take a whole CU each and would wait for the transforms beside them to leave one
        if ((st = spiht_ctx_set_option(p->Lc[s], "wide_encode", 0)) != SPIHT_OK) break;
        if ((st = spiht_event_create(p->Hc, &p->ev_a[s])) != SPIHT_OK) break;
        if ((st = spiht_event_create(p->Lc[s], &p->ev_d[s])) != SPIHT_OK) break;
        if ((st = spiht_event_create(p->Hc, &p->ev_i[s])) != SPIHT_OK) break;
        struct { void **ptr; uint64_t bytes; } al[] = {
            {(void **)&p->coeffs[s], (uint64_t)B * n * 4}, {(void **)&p->rec[s], (uint64_t)B * n * 4},
            {(void **)&p->dmsb[s], (uint64_t)B * n},       {(void **)&p->lmsb[s], (uint64_t)B * n},
            {(void **)&p->maxabs[s], (uint64_t)B * 4},     {(void **)&p->nbytes[s], (uint64_t)B * 8},
            {(void **)&p->flags[s], (uint64_t)B * p->flag_words * 4}};
        for (auto &a : al) {
            if (a.bytes == 0) continue;
            if ((st = spiht_dev_alloc(p->Hc, a.bytes, a.ptr)) != SPIHT_OK) break;
        }
        if (st == SPIHT_OK) st = spiht_dev_memset(p->Hc, p->rec[s], 0, (uint64_t)B * n * 4);  // zero once: U keeps it zero
    }
    if (st == SPIHT_OK) st = spiht_event_create(p->Hc, &p->ev_c);
    if (st == SPIHT_OK) {
        int64_t a_h = 0, a_w = 0;
        st = spiht_idwt_approx_shape(H, W, wavelet, level, &a_h, &a_w);
        if (st == SPIHT_OK && a_h > 0 && a_w > 0 && mode != SPIHT_MODE_PERIODIZATION)
            st = spiht_dev_alloc(p->Hc, (uint64_t)B * c * a_h * a_w * 8, (void **)&p->approx);
    }
    if (st == SPIHT_OK) st = spiht_ctx_synchronize(p->Hc);
    if (st != SPIHT_OK) { pipeline_free(p); return st; }
    *out = p;
    return SPIHT_OK;
}

extern "C" int spiht_pipeline_create(int device, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level,
                                     double q_scale, const double *channel_mults, uint64_t max_bits, spiht_pipeline **out) {
    return pipeline_create(nullptr, device, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, out);
}
// ... with the caller's context for the HBM-bound passes (the two list-coding contexts are still the pipeline's own).  A
// process has few hardware queues for its HIP streams (four by default on this runtime): a caller that already holds a
// context for its uploads should hand it over instead of having a fifth stream share a queue with one of the pipeline's --
// measured: dwt_rest 2.3 instead of 1.7 ms per step with the extra stream.
extern "C" int spiht_pipeline_create_on(spiht_ctx *h_ctx, int device, int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode,
                                        int level, double q_scale, const double *channel_mults, uint64_t max_bits, spiht_pipeline **out) {
    if (!h_ctx) return SPIHT_ERR_ARG;
    return pipeline_create(h_ctx, device, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, out);
}

extern "C" void spiht_pipeline_destroy(spiht_pipeline *p) { pipeline_free(p); }

extern "C" int spiht_pipeline_info(spiht_pipeline *p, uint64_t *slot_stride, int64_t *rec_H, int64_t *rec_W) {
    if (!p) return SPIHT_ERR_ARG;
    if (slot_stride) *slot_stride = p->slot_stride;
    return spiht_geometry_mode(p->H, p->W, p->wavelet, p->mode, p->level, nullptr, nullptr, nullptr, nullptr, nullptr, rec_H, rec_W);
}

extern "C" int spiht_pipeline_contexts(spiht_pipeline *p, spiht_ctx **h, spiht_ctx **l0, spiht_ctx **l1) {
    if (!p) return SPIHT_ERR_ARG;
    if (h) *h = p->Hc;
    if (l0) *l0 = p->Lc[0];
    if (l1) *l1 = p->Lc[1];
    return SPIHT_OK;
}

extern "C" int spiht_pipeline_set_color3(spiht_pipeline *p, const double *A_f, const double *M_f, double p_f, const double *A_i,
                                         const double *M_i, double p_i) {
    if (!p) return SPIHT_ERR_ARG;
    if (!A_f) { p->color_on = false; return SPIHT_OK; }
    if (!M_f || !A_i || !M_i) return SPIHT_ERR_ARG;
    memcpy(p->cAf, A_f, sizeof(p->cAf)); memcpy(p->cMf, M_f, sizeof(p->cMf));
    memcpy(p->cAi, A_i, sizeof(p->cAi)); memcpy(p->cMi, M_i, sizeof(p->cMi));
    p->cpf = p_f; p->cpi = p_i;
    p->color_on = true;  // (kept by the pipeline, put on the H context around each of its calls: HScope)
    return SPIHT_OK;
}

// The inverse transform of batch s in two parts.  Part 1: the coarse levels, queued beside the encoder of the batch after it;
// ev_c marks their end.  Part 2: level 1.  The decoder of the batch after waits for ev_c, so that it starts together with
// level 1 -- whose launch stands right behind ev_c in its own queue and so reaches the CUs first.  That order matters: a CU
// serves the older wavefronts first, and the inverse level 1 beside a decoder that got there first takes
// 4.4 ms, with the decoder behind it 3.75.  Without an approximation buffer (periodization, fewer than two levels) part 1
// is empty and part 2 the whole transform.
static int queue_inverse_coarse(spiht_pipeline *p, int s) {
    CHK(spiht_ctx_wait_event(p->Hc, p->ev_d[s]));
    if (p->approx)
        CHK(spiht_idwt_coarse_batch_f64(p->Hc, p->rec[s], p->B, p->c, p->H, p->W, p->wavelet, p->mode, p->level, p->q, p->mp(), p->approx));
    CHK(spiht_event_record(p->ev_c, p->Hc));
    return SPIHT_OK;
}
static int queue_inverse_level1(spiht_pipeline *p, int s, double *d_img_out) {
    if (p->approx)
        CHK(spiht_idwt_level1_flags_batch_f64(p->Hc, p->rec[s], p->approx, p->flags[s], p->B, p->c, p->H, p->W, p->wavelet, p->mode,
                                              p->level, p->q, p->mp(), d_img_out));
    else
        CHK(spiht_dequant_idwt_flags_batch_f64(p->Hc, p->rec[s], p->flags[s], p->B, p->c, p->H, p->W, p->wavelet, p->mode, p->level,
                                               p->q, p->mp(), d_img_out));
    CHK(spiht_event_record(p->ev_i[s], p->Hc));
    // ... and the zeros back into the array as soon as that has read it, on the batch's own list-coding context
    CHK(spiht_ctx_wait_event(p->Lc[s], p->ev_i[s]));
    CHK(spiht_unscatter_lists_batch_i32(p->Lc[s], p->rec[s], p->B, p->c, p->enc_h, p->enc_w));
    return SPIHT_OK;
}
static int queue_inverse(spiht_pipeline *p, int s, double *d_img_out) {
    CHK(queue_inverse_coarse(p, s));
    return queue_inverse_level1(p, s, d_img_out);
}

static int submit_impl(spiht_pipeline *p, const double *d_img, uint8_t *d_out, uint64_t *d_nbits, uint8_t *d_max_n, double *d_img_out,
                       spiht_comm *comm, uint8_t *d_all_slots, uint64_t *d_all_nbits, uint8_t *d_all_max_n, int rank) {
    const int s = (int)(p->step & 1), o = s ^ 1;
    spiht_ctx *L = p->Lc[s];
    // H: front half of the encoder
    CHK(spiht_dwt_pyramid_batch_f64(p->Hc, d_img, p->B, p->c, p->H, p->W, p->wavelet, p->mode, p->level, p->q, p->mp(),
                                    p->coeffs[s], p->dmsb[s], p->lmsb[s], p->maxabs[s]));
    CHK(spiht_event_record(p->ev_a[s], p->Hc));
    // L: list coding, after the previous batch's decoder on the other context (list kernels never run beside one another)
    if (p->used[o]) CHK(spiht_ctx_wait_event(L, p->ev_d[o]));
    CHK(spiht_ctx_wait_event(L, p->ev_a[s]));
    CHK(spiht_encode_lists_batch_i32(L, p->coeffs[s], p->dmsb[s], p->lmsb[s], p->maxabs[s], p->B, p->c, p->enc_h, p->enc_w, p->ll_h,
                                     p->ll_w, p->max_bits == SPIHT_MAX_BITS_UNLIMITED ? 0 : p->max_bits, d_out, p->slot_stride,
                                     d_nbits, d_max_n));
    const uint8_t *x_out = d_out;
    const uint64_t *x_nbits = d_nbits;
    const uint8_t *x_maxn = d_max_n;
    if (comm) {  // the one exchange of a multi-GPU job, on the list-coding stream; the decoder reads this rank's gathered rows
        CHK(spiht_gather_streams(L, comm, d_out, d_nbits, d_max_n, p->B, p->slot_stride, d_all_slots, d_all_nbits, d_all_max_n));
        int world = 0;
        CHK(spiht_comm_info(comm, &world, nullptr, nullptr));
        uint64_t o_s = 0, o_n = 0, o_m = 0;
        CHK(spiht_gather_row_offsets(rank, world, p->B, p->slot_stride, &o_s, &o_n, &o_m));
        x_out = d_all_slots + o_s;
        x_nbits = (const uint64_t *)((const uint8_t *)d_all_nbits + o_n);
        x_maxn = d_all_max_n + o_m;
    }
    CHK(spiht_nbits_to_nbytes(L, x_nbits, p->B, p->nbytes[s]));
    // H: the previous batch's inverse transform, in two parts: the coarse levels (beside this batch's encoder) and level 1 ...
    if (p->pending) {
        CHK(queue_inverse_coarse(p, p->pending_slot));
        CHK(queue_inverse_level1(p, p->pending_slot, p->pending_out));
        p->pending = false;
        // ... L: this batch's decoder not before the coarse levels are through, and not before level 1's persistent workgroups
        // (three per CU, too large for a fourth: dwt.hip) are on the CUs: a CU serves its older wavefronts first, the inverse
        // level 1 beside a decoder that got there first takes 4.4 instead of 3.7 ms, and decoder workgroups that arrive
        // TOGETHER with it land unevenly (12.5 instead of 8.8 ms for the decoder, round 4).  The wait is for that fact, not for
        // a time: a one-wavefront kernel on L that watches the count of started workgroups (spiht_ctx_wait_resident).
        const void *ctr = nullptr;
        uint32_t target = 0;
        CHK(spiht_ctx_resident_ticket(p->Hc, &ctr, &target));
        CHK(spiht_ctx_wait_event(L, p->ev_c));
        CHK(spiht_ctx_wait_resident(L, ctr, target, 1000));
    }
    CHK(spiht_decode_lists_flags_batch_i32(L, x_out, p->slot_stride, p->nbytes[s], x_maxn, p->B, p->c, p->H, p->W, p->wavelet,
                                           p->mode, p->level, p->rec[s], p->flags[s]));
    CHK(spiht_event_record(p->ev_d[s], L));
    p->used[s] = true;
    p->pending = true;
    p->pending_slot = s;
    p->pending_out = d_img_out;
    p->step++;
    return SPIHT_OK;
}

extern "C" int spiht_pipeline_submit_gather(spiht_pipeline *p, const double *d_img, uint8_t *d_out, uint64_t *d_nbits,
                                            uint8_t *d_max_n, double *d_img_out, spiht_comm *comm, uint8_t *d_all_slots,
                                            uint64_t *d_all_nbits, uint8_t *d_all_max_n, int rank) {
    if (!p || !d_img || !d_out || !d_nbits || !d_max_n || !d_img_out) return SPIHT_ERR_ARG;
    if (comm && (!d_all_slots || !d_all_nbits || !d_all_max_n || rank < 0)) return SPIHT_ERR_ARG;
    if (p->poisoned != SPIHT_OK) return p->poisoned;
    HScope sc(p);
    int st = sc.st;
    if (st == SPIHT_OK) st = submit_impl(p, d_img, d_out, d_nbits, d_max_n, d_img_out, comm, d_all_slots, d_all_nbits, d_all_max_n, rank);
    // a failure between the first and the last queued call leaves events, buffer sets and the pending inverse transform half
    // advanced: nothing later can be trusted, and every later call says so with the first error
    if (st != SPIHT_OK) p->poisoned = st;
    return st;
}

extern "C" int spiht_pipeline_submit(spiht_pipeline *p, const double *d_img, uint8_t *d_out, uint64_t *d_nbits, uint8_t *d_max_n,
                                     double *d_img_out) {
    return spiht_pipeline_submit_gather(p, d_img, d_out, d_nbits, d_max_n, d_img_out, nullptr, nullptr, nullptr, nullptr, 0);
}

extern "C" int spiht_pipeline_flush(spiht_pipeline *p) {
    if (!p) return SPIHT_ERR_ARG;
    if (p->poisoned != SPIHT_OK) return p->poisoned;
    if (p->pending) {
        HScope sc(p);
        int st = sc.st;
        if (st == SPIHT_OK) st = queue_inverse(p, p->pending_slot, p->pending_out);
        if (st != SPIHT_OK) { p->poisoned = st; return st; }
        p->pending = false;
    }
    return SPIHT_OK;
}

extern "C" int spiht_pipeline_synchronize(spiht_pipeline *p) {
    if (!p) return SPIHT_ERR_ARG;
    int st = spiht_pipeline_flush(p);  // (a poisoned pipeline: the contexts are still waited for, the error reported)
    spiht_ctx *cs[3] = {p->Lc[0], p->Lc[1], p->Hc};
    for (spiht_ctx *cx : cs) {  // (every context is waited for, the first error is reported)
        const int s1 = spiht_ctx_synchronize(cx);
        if (st == SPIHT_OK) st = s1;
    }
    return st;
}
