// C ABI of libspiht_hip.so (include/spiht_hip.h): context, scratch, geometry and the host side of the
// encode/decode entry points.  Mirrors what /root/reference/src/lib.rs does around encode()/decode():
// take a strided int32 view, run the coder, pack/unpack bytes -- here by queueing HIP kernels.
#include "../../include/spiht_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "wavelets.h"

extern "C" {
int spiht_launch_absmax(const int32_t *d_x, int B, uint32_t n, uint32_t *d_maxabs, hipStream_t st);
int spiht_launch_pyramid(const Geom *g, int B, const int32_t *d_x, uint8_t *d_dmsb, uint8_t *d_lmsb, hipStream_t st);
int spiht_launch_encode(const EncArgs *a, hipStream_t st);
int spiht_launch_encode_wide(const EncArgs *a, const WideArgs *w, int groups, hipStream_t st);
int spiht_wide_groups_per_cu(void);
int spiht_launch_decode(const DecArgs *a, hipStream_t st);
int spiht_launch_decode_w8(const DecArgs *a, hipStream_t st);  // the 8-wavefront build of decode.hip
int spiht_launch_unscatter(const DecArgs *a, hipStream_t st);
int spiht_meta_sort_temp_bytes(uint64_t rows, size_t *bytes);
int spiht_launch_metadata(const MetaArgs *a, uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                          void *temp, size_t temp_bytes, hipStream_t st);
int spiht_launch_budget_fold(const MetaArgs *a, uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                             void *temp, size_t temp_bytes, const uint64_t *d_budgets, int K, int32_t *d_out, hipStream_t st);
int spiht_launch_nbits_to_nbytes(const uint64_t *d_nbits, int B, uint64_t *d_nbytes, hipStream_t st);
int spiht_launch_color3(const double *d_in, double *d_out, int B, size_t npix, const double *A, const double *M, double p,
                        hipStream_t st);
int spiht_launch_dwt_level(const DwtKArgs *a, int planes, hipStream_t st);
int spiht_launch_dwt_level_ext(const DwtKArgs *a, int planes, void *t_lo, void *t_hi, void *b_aa, void *b_ad, void *b_da,
                               void *b_dd, const double *d_filt, hipStream_t st);
int spiht_launch_idwt_level_per(const IdwtKArgs *a, int planes, double *t_lo, double *t_hi, const double *d_filt, int per,
                                hipStream_t st);
int spiht_launch_idwt_level(const IdwtKArgs *a, int planes, hipStream_t st, TileCtr *tc);
int spiht_launch_quant_plain(const double *in, int32_t *out, size_t n_per_plane, int planes, int c, const double *mults,
                             double q, uint32_t *maxabs, hipStream_t st);
int spiht_launch_zero_pads(int L, const int64_t *hs, const int64_t *ws, const int64_t *offh, const int64_t *offw, int enc_h,
                           int enc_w, int32_t *coeffs, int planes, hipStream_t st);
int spiht_launch_dequant_plain(const int32_t *in, double *out, size_t n_per_plane, int planes, int c,
                               const double *mults, double q, hipStream_t st);
}

static thread_local std::string g_hip_err;

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return SPIHT_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)
#define LAUNCHCHK(expr)                                                                    \
    do {                                                                                   \
        int _e = (expr);                                                                   \
        if (_e != 0) {                                                                     \
            g_hip_err = std::string(#expr) + ": launch error " + std::to_string(_e);       \
            return SPIHT_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)
#define CHK(expr)                      \
    do {                               \
        int _s = (expr);               \
        if (_s != SPIHT_OK) return _s; \
    } while (0)

enum Stage {
    ST_H2D = 0, ST_D2H, ST_ABSMAX, ST_PYRAMID, ST_ENC_LISTS, ST_DEC_LISTS, ST_DWT_L1, ST_DWT_REST, ST_IDWT_REST,
    ST_IDWT_L1, ST_MEMSET, ST_GATHER, ST_COUNT
};
static const char *STAGE_NAMES[ST_COUNT] = {"h2d", "d2h", "absmax", "pyramid", "encode_lists", "decode_lists",
                                            "dwt_level1", "dwt_rest", "idwt_rest", "idwt_level1", "memset", "gather"};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct spiht_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;  // recursive: the host-array entry points call the batched ones
    int num_cu = 256;
    float log2_thresh[32];
    // grow-only scratch
    DevBuf x, dmsb, lmsb, maxabs, out, nbits, maxn, err, lists, coeffs, a0, a1, data, nbytes, rec, mults, img;
    DevBuf trace, meta;  // decode_with_metadata
    DevBuf tilebuf;      // tile counters of the persistent inverse-transform kernel
    TileCtr tilectr = {nullptr, {0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0};
    DevBuf himg, hrec;   // host-array image entry points: pixels in / out, coefficient array in
    std::vector<double> mults_host;  // what ctx->mults holds (uploaded again only when the scales change)
    // colour model of the coded picture (spiht_ctx_set_color3): applied inside level 1 of the transforms of 3-channel images
    bool color_on = false;
    Color3 col_fwd, col_inv;
    int dec_waves = 12;  // wavefronts per decoder workgroup (spiht_ctx_set_decoder_waves)
    // spiht_ctx_set_option
    bool opt_l1_flags = true;   // the decoder flags the occupied level-1 tiles for the inverse transform
    bool opt_pads_persist = false;  // coefficient arrays this context has filled keep their zero padding (see dwt_forward)
    struct PadKey { const void *p; int planes; int64_t H, W; int F, L, mode; };
    std::vector<PadKey> pads_zeroed;  // arrays whose padding strips this context has zeroed (opt_pads_persist)
    DevBuf l1flags;             // L1Flags words of the fused decode path
    DevBuf exttmp;              // intermediates of the two-pass forward level (extension modes that compute their samples)
    DevBuf widebuf;             // control blocks and scan descriptors of the several-CUs-per-image encoder (encode_wide.hip)
    int opt_wide_solo = 24576;  // list entries up to which a plane stays with workgroup 0 (4 k ... 64 k measured the same)
    int opt_wide_g = 0;         // workgroups per image of that encoder (0: by the size of the image)
    int opt_wide_encode = 1;    // few images per call: one image on several CUs (2: whatever the image's size -- tests)
    int wide_per_cu = -1;       // workgroups of k_encode_wide a CU holds (occupancy query, once; 0: unknown -> one)
    std::vector<WideCtl> wide_forced;  // opt_wide_encode == 3 (tests): control blocks that say "gave up" before the launch
    int wide_last_groups = 0;   // groups of the last several-CUs-per-image launch (spiht_ctx_wide_stats)
    DevBuf filt;                // the filters of wavelet `filt_wavelet` on the device, for the two-pass levels (any length)
    int filt_wavelet = -1;
    // decoder output of the fused image path: kept all-zero between calls (k_unscatter), so no per-call zero-fill
    DevBuf recz, lspcnt;
    bool recz_clean = false;
    // spiht_decode_lists_batch_i32 -> spiht_unscatter_lists_batch_i32: the launch whose scatter can still be undone
    DecArgs last_dec;
    bool last_dec_valid = false;
    // timing
    bool timing = false;
    struct Rec { int stage; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double ms[ST_COUNT];
    uint64_t launches[ST_COUNT];
};

// memory at p goes back to the allocator: what is remembered about arrays there is void (option "pads_persist")
static void forget_pads(spiht_ctx *ctx, const void *p) {
    auto &v = ctx->pads_zeroed;
    v.erase(std::remove_if(v.begin(), v.end(), [&](const spiht_ctx::PadKey &k) { return k.p == p; }), v.end());
}

static int ensure(spiht_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return SPIHT_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        forget_pads(ctx, b.p);
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
            b.p = nullptr;
            return SPIHT_ERR_NOMEM;
        }
    }
    b.cap = want;
    return SPIHT_OK;
}

struct StageTimer {
    spiht_ctx *ctx;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(spiht_ctx *c, int s) : ctx(c), stage(s) {
        if (!ctx->timing) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->pool.empty()) { e = ctx->pool.back(); ctx->pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = get();
        b = get();
        if (a) (void)hipEventRecord(a, ctx->stream);
    }
    ~StageTimer() {
        if (!ctx->timing || !a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({stage, a, b});
    }
};

static void drain_timing(spiht_ctx *ctx) {
    if (ctx->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &r : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ctx->ms[r.stage] += ms;
            ctx->launches[r.stage] += 1;
        }
        ctx->pool.push_back(r.a);
        ctx->pool.push_back(r.b);
    }
    ctx->pending.clear();
}

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------

static int dwt_max_level(int64_t len, int F) {  // pywt common.c dwt_max_level
    if (F <= 1 || len < F - 1) return 0;
    int64_t q = len / (F - 1);
    int l = 0;
    while (q > 1) { q >>= 1; l++; }
    return l;
}

struct ImgGeom {
    int L;
    int64_t hs[SPIHT_MAX_LEVELS + 1], ws[SPIHT_MAX_LEVELS + 1];      // band sizes, [0] = image
    int64_t offh[SPIHT_MAX_LEVELS + 1], offw[SPIHT_MAX_LEVELS + 1];  // detail block offsets per level
    int64_t ll_h, ll_w, enc_h, enc_w, rec_H, rec_W;
    bool per;  // periodization: ceil(n / 2) coefficients per level and 2 n samples back (the rule below with a two-tap filter)
};

// mode: only periodization changes the geometry (pywt.dwt_coeff_len); the level count follows the real filter length whatever
// the mode (pywt.dwt_max_level knows none)
static int img_geometry(int64_t H, int64_t W, int F_real, int level, ImgGeom *g, int mode = SPIHT_MODE_REFLECT) {
    if (H < 1 || W < 1) return SPIHT_ERR_ARG;
    int L = level;
    if (L < 0) L = std::min(dwt_max_level(H, F_real), dwt_max_level(W, F_real));
    if (L > SPIHT_MAX_LEVELS) return SPIHT_ERR_ARG;
    g->per = mode == SPIHT_MODE_PERIODIZATION;
    const int F = g->per ? 2 : F_real;
    g->L = L;
    g->hs[0] = H;
    g->ws[0] = W;
    for (int l = 1; l <= L; l++) {
        g->hs[l] = (g->hs[l - 1] + F - 1) / 2;
        g->ws[l] = (g->ws[l - 1] + F - 1) / 2;
    }
    g->ll_h = g->hs[L];
    g->ll_w = g->ws[L];
    int64_t ah = g->ll_h, aw = g->ll_w;
    for (int l = L; l >= 1; l--) {
        g->offh[l] = ah;
        g->offw[l] = aw;
        ah += g->hs[l];
        aw += g->ws[l];
    }
    g->enc_h = ah;
    g->enc_w = aw;
    int64_t rh = g->ll_h, rw = g->ll_w;
    for (int l = L; l >= 1; l--) {
        rh = 2 * g->hs[l] - F + 2;
        rw = 2 * g->ws[l] - F + 2;
    }
    g->rec_H = rh;
    g->rec_W = rw;
    return SPIHT_OK;
}

static int make_geom(int64_t c, int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, Geom *g) {
    if (!(ll_h > 1) || !(ll_w > 1)) return SPIHT_ERR_LL;
    if (c <= 0 || h <= 0 || w <= 0) return SPIHT_ERR_EMPTY;
    // the reference indexes arr[(k, l, m)] for the offspring of every root node: out of bounds -> panic
    int64_t need_h = (ll_h % 2 == 0) ? 2 * ll_h : 2 * ll_h - 1;
    int64_t need_w = (ll_w % 2 == 0) ? 2 * ll_w : 2 * ll_w - 1;
    if (h < need_h || w < need_w) return SPIHT_ERR_SHAPE;
    if ((double)c * (double)h * (double)w >= 1073741824.0) return SPIHT_ERR_TOO_LARGE;
    g->c = (int32_t)c; g->h = (int32_t)h; g->w = (int32_t)w;
    g->ll_h = (int32_t)ll_h; g->ll_w = (int32_t)ll_w;
    g->hw = (uint32_t)(h * w);
    g->n = (uint32_t)(c * h * w);
    g->pad = 0;
    g->div_w = fastdiv_make((uint32_t)w);
    g->div_hw = fastdiv_make(g->hw);
    return SPIHT_OK;
}

// number of tree-node instances (duplicates counted) and of instances with offspring
static void instance_counts(const Geom &g, uint64_t *nodes, uint64_t *parents) {
    const uint64_t h = (uint64_t)g.h, w = (uint64_t)g.w;
    const uint64_t he = h & ~1ull, we = w & ~1ull;  // (i|1) < h  <=>  i < he
    const uint64_t hp = h / 2, wp = w / 2;          // has offspring <=> i < hp && j < wp
    uint64_t nn = (uint64_t)g.ll_h * g.ll_w, pp = 0;
    for (int64_t i = 0; i < g.ll_h; i++)
        for (int64_t j = 0; j < g.ll_w; j++) {
            if (i % 2 == 0 && j % 2 == 0) continue;
            pp += 1;
            uint64_t ri = (uint64_t)((i & 1) * g.ll_h + (i & ~1ll)), rj = (uint64_t)((j & 1) * g.ll_w + (j & ~1ll));
            for (int q = 0; q < 4; q++) {
                uint64_t ci = ri + (q >> 1), cj = rj + (q & 1);
                // sub-tree of (ci,cj): depth t block rows [ci<<t, (ci+1)<<t)
                for (int t = 0; t < 32; t++) {
                    uint64_t r0 = ci << t, c0 = cj << t, sz = 1ull << t;
                    uint64_t lim_h = t == 0 ? h : he, lim_w = t == 0 ? w : we;
                    uint64_t rows = r0 >= lim_h ? 0 : std::min(sz, lim_h - r0);
                    uint64_t cols = c0 >= lim_w ? 0 : std::min(sz, lim_w - c0);
                    if (rows == 0 || cols == 0) break;
                    nn += rows * cols;
                    uint64_t prow = r0 >= hp ? 0 : std::min(sz, hp - r0);
                    uint64_t pcol = c0 >= wp ? 0 : std::min(sz, wp - c0);
                    pp += prow * pcol;
                }
            }
        }
    *nodes = nn * (uint64_t)g.c;
    *parents = pp * (uint64_t)g.c;
}

static uint64_t bound_bits(const Geom &g, uint32_t max_abs) {
    uint64_t nodes, parents;
    instance_counts(g, &nodes, &parents);
    int planes = 1;
    while (planes < 32 && (1ull << planes) <= (uint64_t)max_abs) planes++;
    planes += 1;  // Q1: the start plane can be one above the true msb
    // per node: <= planes LIP zeros + sig + sign + <= planes refinement bits; per parent: <= planes A bits + planes B bits
    return nodes * (2ull * planes + 2) + parents * (2ull * planes);
}

static void list_caps(const Geom &g, uint64_t max_bits, ListCaps *caps, uint64_t *nodes_out) {
    uint64_t nodes, parents;
    instance_counts(g, &nodes, &parents);
    uint64_t roots = (uint64_t)g.c * g.ll_h * g.ll_w;
    uint64_t mb = max_bits;
    uint64_t lip = nodes, lsp = nodes, lis = parents + 4 * roots;  // + leaf A entries under root B entries (Q5)
    if (mb < (1ull << 40)) {
        // what a stream of mb bits can put on the lists ... plus what ONE chunk of the encoder appends: its scans
        // give every entry of the chunk its list slots before the bit budget cuts the chunk short (encode.hip: up to 2048
        // LIP entries or 1024 LIS entries per chunk; encode_wide.hip: 8192 or 2048; at most four appends each; found by
        // tests/test_gpu_spiht.py::test_random_geometries_and_budgets)
        const uint64_t chunk_slack = 16384;
        lip = std::min(lip, roots + mb + chunk_slack);
        lsp = std::min(lsp, mb / 2 + 1 + chunk_slack);
        lis = std::min(lis, roots + 4 * mb + chunk_slack);
    }
    // multiples of 64 entries: every slot's lists start 256-byte aligned (the encoder reads entry pairs as 8-byte loads)
    caps->lip = (uint32_t)std::min<uint64_t>((lip + 127) & ~63ull, 0xFFFFFFC0ull);
    caps->lsp = (uint32_t)std::min<uint64_t>((lsp + 127) & ~63ull, 0xFFFFFFC0ull);
    caps->lis = (uint32_t)std::min<uint64_t>((lis + 127) & ~63ull, 0xFFFFFFC0ull);
    caps->pad = 0;
    if (nodes_out) *nodes_out = nodes;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------

extern "C" const char *spiht_strerror(int s) {
    switch (s) {
    case SPIHT_OK: return "ok";
    case SPIHT_ERR_LL: return "assertion failed: ll_h > 1 && ll_w > 1";
    case SPIHT_ERR_EMPTY: return "empty coefficient array";
    case SPIHT_ERR_SHAPE: return "offspring of the ll_h x ll_w root block fall outside the array (index out of bounds)";
    case SPIHT_ERR_CAPACITY: return "output buffer too small";
    case SPIHT_ERR_HIP: return "HIP runtime error";
    case SPIHT_ERR_ARG: return "invalid argument";
    case SPIHT_ERR_MAGNITUDE: return "coefficient magnitude >= 2^30 is outside the supported range";
    case SPIHT_ERR_INTERNAL: return "internal list capacity guard tripped";
    case SPIHT_ERR_TOO_LARGE: return "array or stream too large (c*h*w must be < 2^30, stream < 2^32 bits)";
    case SPIHT_ERR_NOMEM: return "out of device memory";
    default: return "unknown status";
    }
}
extern "C" const char *spiht_last_hip_error(void) { return g_hip_err.c_str(); }
extern "C" int spiht_abi_version(void) { return 2; }  // 2: round 3 (pipeline, occupancy words, options, geometry_mode, every wavelet / mode)

extern "C" int spiht_ctx_create(int device, spiht_ctx **out) { return spiht_ctx_create_priority(device, 0, out); }

// priority > 0: the context's stream is created with the device's highest stream priority -- its workgroups are placed
// before those of normal streams when both wait for room on the CUs (the list-coding contexts of a pipelined schedule)
extern "C" int spiht_ctx_create_priority(int device, int priority, spiht_ctx **out) {
    if (!out) return SPIHT_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        g_hip_err = "no such HIP device";
        return SPIHT_ERR_HIP;
    }
    HIPCHK(hipSetDevice(device));
    spiht_ctx *ctx = new spiht_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount;
        ctx->tilectr.lds_per_cu = (int32_t)std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor, (size_t)1 << 30);
    }
    ctx->tilectr.num_cu = ctx->num_cu;
    hipError_t e;
    if (priority > 0) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest);
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return SPIHT_ERR_HIP;
    }
    // `(max as f32).log2() as u8` with THIS host's libm (what Rust's f32::log2 lowers to on linux-gnu):
    // thresh[k] = smallest float below 2^k whose truncated log2f already reads k
    for (int k = 0; k < 32; k++) {
        float t = ldexpf(1.0f, k);
        if (k >= 1) {
            float m = nextafterf(t, 0.0f);
            int guard = 0;
            while (guard++ < 64 && m >= 1.0f && (int)log2f(m) == k) { t = m; m = nextafterf(m, 0.0f); }
        }
        ctx->log2_thresh[k] = t;
    }
    for (int s = 0; s < ST_COUNT; s++) { ctx->ms[s] = 0; ctx->launches[s] = 0; }
    int rc = ensure(ctx, ctx->err, 8192);  // (one error word; the rest is for diagnostic builds)
    if (rc != SPIHT_OK) { spiht_ctx_destroy(ctx); return rc; }
    (void)hipMemset(ctx->err.p, 0, 8192);
    rc = ensure(ctx, ctx->tilebuf, TILECTR_WORDS * sizeof(uint32_t));
    if (rc != SPIHT_OK) { spiht_ctx_destroy(ctx); return rc; }
    (void)hipMemset(ctx->tilebuf.p, 0, TILECTR_WORDS * sizeof(uint32_t));
    ctx->tilectr.dev = (uint32_t *)ctx->tilebuf.p;
    *out = ctx;
    return SPIHT_OK;
}

extern "C" void spiht_ctx_destroy(spiht_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->widebuf, &ctx->filt, &ctx->exttmp, &ctx->l1flags, &ctx->x, &ctx->dmsb, &ctx->lmsb, &ctx->maxabs, &ctx->out, &ctx->nbits, &ctx->maxn, &ctx->err,
                      &ctx->lists, &ctx->coeffs, &ctx->a0, &ctx->a1, &ctx->data, &ctx->nbytes, &ctx->rec, &ctx->mults,
                      &ctx->img, &ctx->trace, &ctx->meta, &ctx->recz, &ctx->lspcnt, &ctx->himg, &ctx->hrec, &ctx->tilebuf};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto &r : ctx->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int read_err(spiht_ctx *ctx);
static int clear_err(spiht_ctx *ctx);
// Waits for the context's stream and reports (then clears) what the device-side guards of the batched calls
// queued since the last synchronize recorded.
extern "C" int spiht_ctx_synchronize(spiht_ctx *ctx) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    int st = read_err(ctx);  // includes the stream synchronize
    if (st != SPIHT_OK) {
        ctx->recz_clean = false;  // a guard tripped: the decoder's lists may not describe what it wrote
        ctx->last_dec_valid = false;
        (void)clear_err(ctx);
        (void)hipStreamSynchronize(ctx->stream);
    }
    return st;
}
extern "C" int spiht_ctx_wait_on(spiht_ctx *ctx, spiht_ctx *other) {
    if (!ctx || !other || ctx->device != other->device) return SPIHT_ERR_ARG;
    if (ctx == other) return SPIHT_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, other->stream);
    hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ctx->stream, ev, 0) : e1;
    (void)hipEventDestroy(ev);  // released once the wait has been satisfied
    HIPCHK(e2);
    return SPIHT_OK;
}
extern "C" int spiht_ctx_set_timing(spiht_ctx *ctx, int enabled) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    drain_timing(ctx);
    ctx->timing = enabled != 0;
    return SPIHT_OK;
}
extern "C" int spiht_ctx_reset_timing(spiht_ctx *ctx) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    drain_timing(ctx);
    for (int s = 0; s < ST_COUNT; s++) { ctx->ms[s] = 0; ctx->launches[s] = 0; }
    return SPIHT_OK;
}
extern "C" int spiht_ctx_num_stages(void) { return ST_COUNT; }
extern "C" const char *spiht_ctx_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? STAGE_NAMES[s] : ""; }
extern "C" int spiht_ctx_get_timing(spiht_ctx *ctx, int stage, double *ms, uint64_t *launches) {
    if (!ctx || stage < 0 || stage >= ST_COUNT) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    drain_timing(ctx);
    if (ms) *ms = ctx->ms[stage];
    if (launches) *launches = ctx->launches[stage];
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// device-side error word
// ------------------------------------------------------------------------------------------------
static int clear_err(spiht_ctx *ctx) {
    HIPCHK(hipMemsetAsync(ctx->err.p, 0, 4, ctx->stream));
    return SPIHT_OK;
}
static int read_err(spiht_ctx *ctx) {
    uint32_t e = 0;
    HIPCHK(hipMemcpyAsync(&e, ctx->err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (e & 2u) return SPIHT_ERR_MAGNITUDE;
    if (e & 4u) return SPIHT_ERR_CAPACITY;
    if (e & 1u) {
        g_hip_err = "device error word 0x" + [](uint32_t v) { char b[16]; snprintf(b, sizeof b, "%x", v); return std::string(b); }(e);
        return SPIHT_ERR_INTERNAL;
    }
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// list coder plumbing
// ------------------------------------------------------------------------------------------------

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// carve per-slot list scratch out of ctx->lists; returns pointers
struct ListPtrs {
    uint32_t *lip0, *lip1, *lsp, *lis0, *lis1, *lis2;
    int32_t *lsp_val;
};

static int alloc_lists(spiht_ctx *ctx, const ListCaps &caps, int want_slots, bool decoder, int *nslots, ListPtrs *p) {
    size_t per_slot = ((size_t)caps.lip * 2 + (size_t)caps.lsp * (decoder ? 2 : 1) + (size_t)caps.lis * 3) * 4;
    // keep the scratch under ~24 GiB
    const size_t budget = (size_t)24 << 30;
    int slots = want_slots;
    if ((size_t)slots * per_slot > budget) slots = (int)std::max<size_t>(1, budget / per_slot);
    size_t s_lip = align256((size_t)caps.lip * 4 * slots), s_lsp = align256((size_t)caps.lsp * 4 * slots),
           s_lis = align256((size_t)caps.lis * 4 * slots);
    size_t total = 2 * s_lip + (decoder ? 2 : 1) * s_lsp + 3 * s_lis;
    ctx->last_dec_valid = false;  // the lists get a new user
    CHK(ensure(ctx, ctx->lists, total));
    char *base = (char *)ctx->lists.p;
    p->lip0 = (uint32_t *)base; base += s_lip;
    p->lip1 = (uint32_t *)base; base += s_lip;
    p->lsp = (uint32_t *)base; base += s_lsp;
    p->lsp_val = nullptr;
    if (decoder) { p->lsp_val = (int32_t *)base; base += s_lsp; }
    p->lis0 = (uint32_t *)base; base += s_lis;
    p->lis1 = (uint32_t *)base; base += s_lis;
    p->lis2 = (uint32_t *)base; base += s_lis;
    *nslots = slots;
    return SPIHT_OK;
}

static int encode_lists_device(spiht_ctx *ctx, const Geom &g, const int32_t *d_x, const uint8_t *d_dmsb,
                               const uint8_t *d_lmsb, const uint32_t *d_maxabs, int B, uint64_t max_bits,
                               const ListCaps &caps, int nslots, const ListPtrs &lp, uint8_t *d_out, uint64_t slot_stride,
                               uint64_t *d_nbits, uint8_t *d_maxn);

// Encode B device-resident coefficient arrays.  max_bits already validated; queues work on ctx->stream.
static int encode_device(spiht_ctx *ctx, const Geom &g, const int32_t *d_x, int B, uint64_t max_bits_in, uint8_t *d_out,
                         uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_maxn, bool have_maxabs = false) {
    if (slot_stride % 4 != 0) return SPIHT_ERR_ARG;
    if ((uint64_t)B * (uint64_t)g.c > 65535ull) return SPIHT_ERR_ARG;
    const uint64_t max_bits = max_bits_in == 0 ? SPIHT_MAX_BITS_UNLIMITED : max_bits_in;  // encoder_decoder.rs:196
    CHK(ensure(ctx, ctx->dmsb, (size_t)B * g.n));
    CHK(ensure(ctx, ctx->lmsb, (size_t)B * g.n));
    CHK(ensure(ctx, ctx->maxabs, (size_t)B * 4));
    ListCaps caps;
    list_caps(g, std::min<uint64_t>(max_bits, slot_stride * 8), &caps, nullptr);
    int nslots = 0;
    ListPtrs lp;
    CHK(alloc_lists(ctx, caps, std::min(B, ctx->num_cu), false, &nslots, &lp));
    {
        StageTimer t(ctx, ST_MEMSET);
        HIPCHK(hipMemsetAsync(d_out, 0, (size_t)B * slot_stride, ctx->stream));
    }
    if (!have_maxabs) {
        StageTimer t(ctx, ST_ABSMAX);
        LAUNCHCHK(spiht_launch_absmax(d_x, B, g.n, (uint32_t *)ctx->maxabs.p, ctx->stream));
    }
    {
        StageTimer t(ctx, ST_PYRAMID);
        LAUNCHCHK(spiht_launch_pyramid(&g, B, d_x, (uint8_t *)ctx->dmsb.p, (uint8_t *)ctx->lmsb.p, ctx->stream));
    }
    return encode_lists_device(ctx, g, d_x, (const uint8_t *)ctx->dmsb.p, (const uint8_t *)ctx->lmsb.p,
                               (const uint32_t *)ctx->maxabs.p, B, max_bits, caps, nslots, lp, d_out, slot_stride, d_nbits,
                               d_maxn);
}

// the list-coding half of the encoder: coefficient arrays + their significance pyramid -> streams
static int encode_lists_device(spiht_ctx *ctx, const Geom &g, const int32_t *d_x, const uint8_t *d_dmsb,
                               const uint8_t *d_lmsb, const uint32_t *d_maxabs, int B, uint64_t max_bits,
                               const ListCaps &caps, int nslots, const ListPtrs &lp, uint8_t *d_out, uint64_t slot_stride,
                               uint64_t *d_nbits, uint8_t *d_maxn) {
    EncArgs a;
    memset(&a, 0, sizeof(a));
    a.g = g;
    a.caps = caps;
    a.B = B;
    a.nslots = nslots;
    a.x = d_x;
    a.dmsb = d_dmsb;
    a.lmsb = d_lmsb;
    a.maxabs = d_maxabs;
    a.max_bits = max_bits;
    a.out = d_out;
    a.slot_stride = slot_stride;
    a.out_nbits = d_nbits;
    a.out_maxn = d_maxn;
    a.lip0 = lp.lip0; a.lip1 = lp.lip1; a.lsp = lp.lsp; a.lis0 = lp.lis0; a.lis1 = lp.lis1; a.lis2 = lp.lis2;
    a.err = (uint32_t *)ctx->err.p;
    memcpy(a.log2_thresh, ctx->log2_thresh, sizeof(a.log2_thresh));
    // Few images per call: each on a group of G workgroups (encode_wide.hip) instead of one -- a single image's list coding
    // is bound by the one CU it runs on.  Every workgroup of a group must be resident at once: B * G within what the device
    // holds of that kernel (asked of the runtime once per context).
    int G = (int)std::min<uint64_t>(64, std::max<uint64_t>(2, g.n >> 18));
    if (ctx->opt_wide_g > 0) G = ctx->opt_wide_g;
    if (ctx->wide_per_cu < 0) ctx->wide_per_cu = std::max(0, spiht_wide_groups_per_cu());
    G = std::min(G, std::max(1, ctx->wide_per_cu) * ctx->num_cu / std::max(B, 1));
    if (ctx->opt_wide_encode && G >= 2 && (g.n >= (1u << 18) || ctx->opt_wide_encode >= 2) && nslots >= B) {
        WideArgs w;
        const uint64_t cap_max = std::max<uint64_t>(caps.lip, std::max<uint64_t>(caps.lsp, caps.lis));
        w.maxchunks = (uint32_t)(cap_max / 2048 + 2);  // (the smaller of the two chunk sizes: WIDE_U * 1024 entries)
        w.G = (uint32_t)G;
        w.solo = (uint32_t)ctx->opt_wide_solo;
        w.pad = 0;
        const size_t ctl_bytes = align256((size_t)B * sizeof(WideCtl)), desc_bytes = (size_t)B * 2 * w.maxchunks * 4 * 8;
        CHK(ensure(ctx, ctx->widebuf, ctl_bytes + desc_bytes));
        w.ctl = (WideCtl *)ctx->widebuf.p;
        w.desc = (uint64_t *)((char *)ctx->widebuf.p + ctl_bytes);
        StageTimer t(ctx, ST_ENC_LISTS);
        HIPCHK(hipMemsetAsync(ctx->widebuf.p, 0, ctl_bytes + desc_bytes, ctx->stream));
        ctx->wide_last_groups = B;
        if (ctx->opt_wide_encode == 3) {  // tests: every group finds itself given up -> k_encode<redo> codes every image
            HIPCHK(hipStreamSynchronize(ctx->stream));  // (an earlier launch may still copy from the vector)
            WideCtl gave_up;
            memset(&gave_up, 0, sizeof(gave_up));
            gave_up.bad = 2u;
            ctx->wide_forced.assign((size_t)B, gave_up);
            HIPCHK(hipMemcpyAsync(ctx->widebuf.p, ctx->wide_forced.data(), (size_t)B * sizeof(WideCtl), hipMemcpyHostToDevice, ctx->stream));
        }
        // The workgroups of a group wait for one another, so a grid should become resident as a whole: two such grids of
        // different contexts, each half resident on a full GPU, would wait for each other.  One at a time per device, by an
        // event chain between the contexts' streams (the host does not block).  What that chain cannot see -- another
        // process, another kernel holding the CUs -- ends in the group giving up after a bounded wait (encode_wide.hip), and
        // the launch of k_encode<redo> right behind codes exactly those images with one workgroup each: same bits, later.
        static std::mutex wide_mu;
        static hipEvent_t wide_last[64] = {};
        {
            std::lock_guard<std::mutex> wl(wide_mu);
            hipEvent_t &ev = wide_last[ctx->device & 63];
            if (ev) HIPCHK(hipStreamWaitEvent(ctx->stream, ev, 0));
            else HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            LAUNCHCHK(spiht_launch_encode_wide(&a, &w, B, ctx->stream));
            HIPCHK(hipEventRecord(ev, ctx->stream));
        }
        a.redo = w.ctl;
        a.nslots = B;
        LAUNCHCHK(spiht_launch_encode(&a, ctx->stream));
        return SPIHT_OK;
    }
    {
        StageTimer t(ctx, ST_ENC_LISTS);
        LAUNCHCHK(spiht_launch_encode(&a, ctx->stream));
    }
    return SPIHT_OK;
}

static int decode_device(spiht_ctx *ctx, const Geom &g, const uint8_t *d_data, uint64_t slot_stride,
                         const uint64_t *d_nbytes, const uint8_t *d_maxn, int B, int32_t *d_out,
                         uint32_t *d_tr_ent = nullptr, uint8_t *d_tr_act = nullptr, uint64_t tr_stride = 0,
                         bool zero_out = true, DecArgs *args_out = nullptr, const L1Flags *fl = nullptr) {
    if (slot_stride % 4 != 0) return SPIHT_ERR_ARG;
    if (slot_stride * 8 >= 0xFFFFFF00ull) return SPIHT_ERR_TOO_LARGE;
    ListCaps caps;
    list_caps(g, slot_stride * 8, &caps, nullptr);
    int nslots = 0;
    ListPtrs lp;
    CHK(alloc_lists(ctx, caps, std::min(B, ctx->num_cu * 8), true, &nslots, &lp));
    if (zero_out) {
        StageTimer t(ctx, ST_MEMSET);
        HIPCHK(hipMemsetAsync(d_out, 0, (size_t)B * g.n * 4, ctx->stream));
    }
    DecArgs a;
    memset(&a, 0, sizeof(a));
    a.g = g;
    a.caps = caps;
    a.B = B;
    a.nslots = nslots;
    a.data = d_data;
    a.slot_stride = slot_stride;
    a.nbytes = d_nbytes;
    a.max_n = d_maxn;
    a.out = d_out;
    a.lip0 = lp.lip0; a.lip1 = lp.lip1; a.lsp_idx = lp.lsp; a.lsp_val = lp.lsp_val;
    a.lis0 = lp.lis0; a.lis1 = lp.lis1; a.lis2 = lp.lis2;
    a.err = (uint32_t *)ctx->err.p;
    a.tr_ent = d_tr_ent; a.tr_act = d_tr_act; a.tr_stride = tr_stride;
    if (fl && fl->p) {  // (zero-filled here: the decoder only ever sets words)
        a.fl = *fl;
        HIPCHK(hipMemsetAsync(fl->p, 0, (size_t)B * g.c * fl->gy * fl->gx * 4, ctx->stream));
    }
    if (args_out) {  // the caller wants to undo the scatter later: one LSP length per slot
        CHK(ensure(ctx, ctx->lspcnt, (size_t)nslots * 4));
        a.lsp_count = (uint32_t *)ctx->lspcnt.p;
    }
    {
        StageTimer t(ctx, ST_DEC_LISTS);
        LAUNCHCHK(ctx->dec_waves == 8 ? spiht_launch_decode_w8(&a, ctx->stream) : spiht_launch_decode(&a, ctx->stream));
    }
    if (args_out) *args_out = a;
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// L2 boundary, single image, host buffers
// ------------------------------------------------------------------------------------------------

extern "C" int spiht_encode_bound(int64_t c, int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, uint32_t max_abs,
                                  uint64_t max_bits, uint64_t *bound_bytes) {
    if (!bound_bytes) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    uint64_t bits = bound_bits(g, max_abs);
    if (max_bits != 0) bits = std::min(bits, max_bits);
    *bound_bytes = ((bits + 7) / 8 + 3) & ~3ull;
    return SPIHT_OK;
}

extern "C" int spiht_encode_i32(spiht_ctx *ctx, const int32_t *x, int64_t c, int64_t h, int64_t w, int64_t stride_c,
                                int64_t stride_h, int64_t stride_w, int64_t ll_h, int64_t ll_w, uint64_t max_bits,
                                uint8_t *out, uint64_t out_cap, uint64_t *out_nbits, uint8_t *max_n) {
    if (!ctx || !out_nbits || !max_n || (!out && out_cap)) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (!x) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    // gather the strided view into a contiguous staging buffer (lib.rs:27 takes any strides)
    std::vector<int32_t> stage;
    const int32_t *src = x;
    uint32_t max_abs = 0;
    const bool contiguous = stride_w == 1 && stride_h == w && stride_c == h * w;
    if (!contiguous) {
        stage.resize(g.n);
        size_t t = 0;
        for (int64_t k = 0; k < c; k++)
            for (int64_t i = 0; i < h; i++) {
                const int32_t *row = x + k * stride_c + i * stride_h;
                for (int64_t j = 0; j < w; j++) stage[t++] = row[j * stride_w];
            }
        src = stage.data();
    }
    for (size_t t = 0; t < g.n; t++) {
        int32_t v = src[t];
        uint32_t m = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
        if (m > max_abs) max_abs = m;
    }
    if (max_abs >= (1u << 30)) return SPIHT_ERR_MAGNITUDE;
    uint64_t bits = bound_bits(g, max_abs);
    if (max_bits != 0) bits = std::min(bits, max_bits);
    if (bits >= 0xFFFFFF00ull * 8ull) return SPIHT_ERR_TOO_LARGE;
    const uint64_t slot = std::max<uint64_t>(4, ((bits + 7) / 8 + 3) & ~3ull);
    CHK(ensure(ctx, ctx->x, (size_t)g.n * 4));
    CHK(ensure(ctx, ctx->out, slot));
    CHK(ensure(ctx, ctx->nbits, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemcpyAsync(ctx->x.p, src, (size_t)g.n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    CHK(encode_device(ctx, g, (const int32_t *)ctx->x.p, 1, max_bits, (uint8_t *)ctx->out.p, slot,
                      (uint64_t *)ctx->nbits.p, (uint8_t *)ctx->maxn.p));
    uint64_t nbits = 0;
    uint8_t mn = 0;
    HIPCHK(hipMemcpyAsync(&nbits, ctx->nbits.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&mn, ctx->maxn.p, 1, hipMemcpyDeviceToHost, ctx->stream));
    CHK(read_err(ctx));
    *out_nbits = nbits;
    *max_n = mn;
    const uint64_t nbytes = (nbits + 7) / 8;
    if (nbytes > out_cap) return SPIHT_ERR_CAPACITY;
    if (nbytes) {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(out, ctx->out.p, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

extern "C" int spiht_decode_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h,
                                int64_t w, int64_t ll_h, int64_t ll_w, int32_t *out) {
    if (!ctx || !out || (!data && nbytes)) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (n > 30) return SPIHT_ERR_MAGNITUDE;
    if (nbytes * 8 >= 0xFFFFFF00ull) return SPIHT_ERR_TOO_LARGE;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t slot = std::max<uint64_t>(4, (nbytes + 3) & ~3ull);
    CHK(ensure(ctx, ctx->data, slot));
    CHK(ensure(ctx, ctx->nbytes, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(ensure(ctx, ctx->rec, (size_t)g.n * 4));
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemsetAsync(ctx->data.p, 0, slot, ctx->stream));
        if (nbytes) HIPCHK(hipMemcpyAsync(ctx->data.p, data, nbytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->nbytes.p, &nbytes, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->maxn.p, &n, 1, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // &nbytes / &n are stack temporaries
    }
    CHK(decode_device(ctx, g, (const uint8_t *)ctx->data.p, slot, (const uint64_t *)ctx->nbytes.p,
                      (const uint8_t *)ctx->maxn.p, 1, (int32_t *)ctx->rec.p));
    CHK(read_err(ctx));
    {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(out, ctx->rec.p, (size_t)g.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

// number of generations below the LL block that exist in the index-based tree (encoder_decoder.rs:43-75): the
// shallowest first-generation nodes are (0, ll_w) and (ll_h, 0); a node has offspring iff 2i+1 < h && 2j+1 < w
static int tree_generations(const Geom &g) {
    int best = 1;
    const int64_t cand[2][2] = {{0, g.ll_w}, {g.ll_h, 0}};
    for (auto &cd : cand) {
        int t = 1;
        while (t < 40 && 2 * (cd[0] << (t - 1)) + 1 < g.h && 2 * (cd[1] << (t - 1)) + 1 < g.w) t++;
        best = std::max(best, t);
    }
    return best;
}

extern "C" int spiht_decode_with_metadata_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c,
                                              int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, const int64_t *top_slice,
                                              const int64_t *other_slices, int64_t level, int32_t *out, int32_t *meta) {
    if (!ctx || !out || !meta || (!data && nbytes) || !top_slice || level < 0 || (level > 0 && !other_slices))
        return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (g.n >= (1u << 28)) return SPIHT_ERR_TOO_LARGE;  // list entries carry the filter in bits 28-29
    if (n > 30) return SPIHT_ERR_MAGNITUDE;
    if (nbytes * 8 >= 0xFFFFFF00ull) return SPIHT_ERR_TOO_LARGE;
    if (level > 255 || tree_generations(g) > level) return SPIHT_ERR_SHAPE;  // other_slices[depth_i] out of bounds (:603)
    std::vector<int32_t> sl(4 + (size_t)level * 12);
    for (size_t t = 0; t < sl.size(); t++) {
        const int64_t v = t < 4 ? top_slice[t] : other_slices[t - 4];
        if (v < 0 || v > 0x7FFFFFFF) return SPIHT_ERR_ARG;
        sl[t] = (int32_t)v;
    }
    for (size_t t = 4; t < sl.size(); t += 4)
        if (sl[t + 1] < sl[t] || sl[t + 3] < sl[t + 2]) return SPIHT_ERR_SHAPE;  // usize underflow in end - start (:606-608)
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t slot = std::max<uint64_t>(4, (nbytes + 3) & ~3ull);
    const uint64_t rows = nbytes * 8 + 1;
    size_t sort_bytes = 0;
    if (spiht_meta_sort_temp_bytes(rows, &sort_bytes) != 0) return SPIHT_ERR_INTERNAL;
    // trace scratch: ent[rows] u32 | 4 x u32[rows] sort buffers | act[rows] u8 | slices | sort temp
    const size_t o_ent = 0, o_k0 = align256(rows * 4), o_v0 = o_k0 + align256(rows * 4), o_k1 = o_v0 + align256(rows * 4),
                 o_v1 = o_k1 + align256(rows * 4), o_act = o_v1 + align256(rows * 4), o_sl = o_act + align256(rows),
                 o_tmp = o_sl + align256(sl.size() * 4), total = o_tmp + align256(sort_bytes);
    CHK(ensure(ctx, ctx->data, slot));
    CHK(ensure(ctx, ctx->nbytes, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(ensure(ctx, ctx->rec, (size_t)g.n * 4));
    CHK(ensure(ctx, ctx->trace, total));
    CHK(ensure(ctx, ctx->meta, rows * 32));
    char *tb = (char *)ctx->trace.p;
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemsetAsync(ctx->data.p, 0, slot, ctx->stream));
        if (nbytes) HIPCHK(hipMemcpyAsync(ctx->data.p, data, nbytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->nbytes.p, &nbytes, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->maxn.p, &n, 1, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(tb + o_sl, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(tb + o_act, TR_NONE, rows, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // stack / vector temporaries
    }
    CHK(decode_device(ctx, g, (const uint8_t *)ctx->data.p, slot, (const uint64_t *)ctx->nbytes.p,
                      (const uint8_t *)ctx->maxn.p, 1, (int32_t *)ctx->rec.p, (uint32_t *)(tb + o_ent),
                      (uint8_t *)(tb + o_act), rows));
    MetaArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.g = g;
    ma.level = (int32_t)level;
    ma.rows = rows;
    ma.tr_ent = (const uint32_t *)(tb + o_ent);
    ma.tr_act = (const uint8_t *)(tb + o_act);
    ma.data = (const uint8_t *)ctx->data.p;
    ma.slices = (const int32_t *)(tb + o_sl);
    ma.meta = (int32_t *)ctx->meta.p;
    ma.skey = (const uint32_t *)(tb + o_k1);
    ma.spos = (const uint32_t *)(tb + o_v1);
    LAUNCHCHK(spiht_launch_metadata(&ma, (uint32_t *)(tb + o_k0), (uint32_t *)(tb + o_v0), (uint32_t *)(tb + o_k1),
                                    (uint32_t *)(tb + o_v1), tb + o_tmp, sort_bytes, ctx->stream));
    CHK(read_err(ctx));
    {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(out, ctx->rec.p, (size_t)g.n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(meta, ctx->meta.p, rows * 32, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

// Progressive decoding of ONE stream to K bit budgets (ascending) from one walk: out[kk] = what decode() returns for the
// first budgets[kk] bits of the stream (a budget past the end: the whole stream).  The pattern of the reference's
// make_gif.py:46-61 (`decode(original_bytes[:byte_len])` per frame), SURVEY.md 8 f-3: the K walks of K prefixes become
// one walk with the trace of decode_with_metadata, and every tree node replays its own operations once
// (metadata.hip: k_budget_fold).  d_out: device int32 [K, c, h, w], zero-filled here.  Asynchronous apart from the upload.
extern "C" int spiht_decode_budgets_dev_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h,
                                            int64_t w, int64_t ll_h, int64_t ll_w, const uint64_t *budgets_bits, int64_t K,
                                            int32_t *d_out) {
    if (!ctx || !d_out || (!data && nbytes) || !budgets_bits || K < 1 || K > 65535) return SPIHT_ERR_ARG;
    for (int64_t k = 1; k < K; k++)
        if (budgets_bits[k] < budgets_bits[k - 1]) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (g.n >= (1u << 28)) return SPIHT_ERR_TOO_LARGE;
    if (n > 30) return SPIHT_ERR_MAGNITUDE;
    if (nbytes * 8 >= 0xFFFFFF00ull) return SPIHT_ERR_TOO_LARGE;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t slot = std::max<uint64_t>(4, (nbytes + 3) & ~3ull);
    const uint64_t rows = nbytes * 8 + 1;
    size_t sort_bytes = 0;
    if (spiht_meta_sort_temp_bytes(rows, &sort_bytes) != 0) return SPIHT_ERR_INTERNAL;
    // trace scratch: ent[rows] u32 | 4 x u32[rows] sort buffers | act[rows] u8 | budgets | sort temp
    const size_t o_ent = 0, o_k0 = align256(rows * 4), o_v0 = o_k0 + align256(rows * 4), o_k1 = o_v0 + align256(rows * 4),
                 o_v1 = o_k1 + align256(rows * 4), o_act = o_v1 + align256(rows * 4), o_bud = o_act + align256(rows),
                 o_tmp = o_bud + align256((size_t)K * 8), total = o_tmp + align256(sort_bytes);
    CHK(ensure(ctx, ctx->data, slot));
    CHK(ensure(ctx, ctx->nbytes, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(ensure(ctx, ctx->rec, (size_t)g.n * 4));
    CHK(ensure(ctx, ctx->trace, total));
    char *tb = (char *)ctx->trace.p;
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemsetAsync(ctx->data.p, 0, slot, ctx->stream));
        if (nbytes) HIPCHK(hipMemcpyAsync(ctx->data.p, data, nbytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->nbytes.p, &nbytes, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->maxn.p, &n, 1, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(tb + o_bud, budgets_bits, (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(tb + o_act, TR_NONE, rows, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // stack temporaries / the caller's arrays
    }
    HIPCHK(hipMemsetAsync(d_out, 0, (size_t)K * g.n * 4, ctx->stream));
    CHK(decode_device(ctx, g, (const uint8_t *)ctx->data.p, slot, (const uint64_t *)ctx->nbytes.p,
                      (const uint8_t *)ctx->maxn.p, 1, (int32_t *)ctx->rec.p, (uint32_t *)(tb + o_ent),
                      (uint8_t *)(tb + o_act), rows));
    MetaArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.g = g;
    ma.rows = rows;
    ma.tr_ent = (const uint32_t *)(tb + o_ent);
    ma.tr_act = (const uint8_t *)(tb + o_act);
    ma.data = (const uint8_t *)ctx->data.p;
    ma.skey = (const uint32_t *)(tb + o_k1);
    ma.spos = (const uint32_t *)(tb + o_v1);
    LAUNCHCHK(spiht_launch_budget_fold(&ma, (uint32_t *)(tb + o_k0), (uint32_t *)(tb + o_v0), (uint32_t *)(tb + o_k1),
                                       (uint32_t *)(tb + o_v1), tb + o_tmp, sort_bytes, (const uint64_t *)(tb + o_bud), (int)K,
                                       d_out, ctx->stream));
    return SPIHT_OK;
}

// ... with the K arrays brought to the host (int32 [K, c, h, w]); synchronous.
extern "C" int spiht_decode_budgets_i32(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h,
                                        int64_t w, int64_t ll_h, int64_t ll_w, const uint64_t *budgets_bits, int64_t K,
                                        int32_t *out) {
    if (!ctx || !out || K < 1) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)K * g.n * 4;
    CHK(ensure(ctx, ctx->hrec, bytes));
    CHK(spiht_decode_budgets_dev_i32(ctx, data, nbytes, n, c, h, w, ll_h, ll_w, budgets_bits, K, (int32_t *)ctx->hrec.p));
    CHK(read_err(ctx));
    {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(out, ctx->hrec.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// batched, device-resident
// ------------------------------------------------------------------------------------------------

static int batch_chunk(const Geom &g) { return (int)std::max<int64_t>(1, 65535 / g.c); }

extern "C" int spiht_encode_batch_i32(spiht_ctx *ctx, const int32_t *d_x, int64_t B, int64_t c, int64_t h, int64_t w,
                                      int64_t ll_h, int64_t ll_w, uint64_t max_bits, uint8_t *d_out,
                                      uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_max_n) {
    if (!ctx || !d_x || !d_out || !d_nbits || !d_max_n || B < 0) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const int chunk = batch_chunk(g);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        CHK(encode_device(ctx, g, d_x + (size_t)b0 * g.n, nb, max_bits, d_out + (size_t)b0 * slot_stride, slot_stride,
                          d_nbits + b0, d_max_n + b0));
    }
    return SPIHT_OK;  // asynchronous: errors surface in spiht_ctx_synchronize()
}

extern "C" int spiht_decode_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride,
                                      const uint64_t *d_nbytes, const uint8_t *d_max_n, int64_t B, int64_t c, int64_t h,
                                      int64_t w, int64_t ll_h, int64_t ll_w, int32_t *d_out) {
    if (!ctx || !d_data || !d_nbytes || !d_max_n || !d_out || B < 0) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    CHK(decode_device(ctx, g, d_data, slot_stride, d_nbytes, d_max_n, (int)B, d_out));
    return SPIHT_OK;  // asynchronous: errors surface in spiht_ctx_synchronize()
}

extern "C" int spiht_pyramid_batch_i32(spiht_ctx *ctx, const int32_t *d_x, int64_t B, int64_t c, int64_t h, int64_t w,
                                       int64_t ll_h, int64_t ll_w, uint8_t *d_dmsb, uint8_t *d_lmsb,
                                       uint32_t *d_maxabs) {
    if (!ctx || !d_x || !d_dmsb || !d_lmsb || B < 0) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (B == 0) return SPIHT_OK;
    if ((uint64_t)B * (uint64_t)g.c > 65535ull) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if (d_maxabs) {  // null: the caller has max|x| already (spiht_dwt_pyramid_batch_f64 without the pyramid)
        StageTimer t(ctx, ST_ABSMAX);
        LAUNCHCHK(spiht_launch_absmax(d_x, (int)B, g.n, d_maxabs, ctx->stream));
    }
    {
        StageTimer t(ctx, ST_PYRAMID);
        LAUNCHCHK(spiht_launch_pyramid(&g, (int)B, d_x, d_dmsb, d_lmsb, ctx->stream));
    }
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// image path
// ------------------------------------------------------------------------------------------------

extern "C" int spiht_wavelet_id(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < SPIHT_NWAVELETS; i++)
        if (!strcmp(SPIHT_WAVELETS[i].name, name)) return i;
    if (!strcmp(name, "db1")) return spiht_wavelet_id("haar");
    return -1;
}
// filter length of a wavelet of the table (pywt.Wavelet(name).dec_len); < 0: no such id
extern "C" int spiht_wavelet_taps(int wavelet) {
    return (wavelet >= 0 && wavelet < SPIHT_NWAVELETS) ? SPIHT_WAVELETS[wavelet].F : -1;
}
extern "C" int spiht_mode_id(const char *name) {
    if (!name) return -1;
    static const char *names[] = {"reflect", "symmetric", "periodic", "zero", "constant", "smooth", "antisymmetric", "antireflect",
                                  "periodization"};
    for (int i = 0; i < 9; i++)
        if (!strcmp(names[i], name)) return i;
    return -1;
}

extern "C" int spiht_geometry_mode(int64_t H, int64_t W, int wavelet, int mode, int level, int *level_used, int64_t *ll_h,
                                   int64_t *ll_w, int64_t *enc_h, int64_t *enc_w, int64_t *rec_H, int64_t *rec_W);
extern "C" int spiht_geometry(int64_t H, int64_t W, int wavelet, int level, int *level_used, int64_t *ll_h,
                              int64_t *ll_w, int64_t *enc_h, int64_t *enc_w, int64_t *rec_H, int64_t *rec_W) {
    return spiht_geometry_mode(H, W, wavelet, SPIHT_MODE_REFLECT, level, level_used, ll_h, ll_w, enc_h, enc_w, rec_H, rec_W);
}
extern "C" int spiht_geometry_mode(int64_t H, int64_t W, int wavelet, int mode, int level, int *level_used, int64_t *ll_h,
                                   int64_t *ll_w, int64_t *enc_h, int64_t *enc_w, int64_t *rec_H, int64_t *rec_W) {
    if (wavelet < 0 || wavelet >= SPIHT_NWAVELETS || mode < 0 || mode > SPIHT_MODE_PERIODIZATION) return SPIHT_ERR_ARG;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    if (level_used) *level_used = ig.L;
    if (ll_h) *ll_h = ig.ll_h;
    if (ll_w) *ll_w = ig.ll_w;
    if (enc_h) *enc_h = ig.enc_h;
    if (enc_w) *enc_w = ig.enc_w;
    if (rec_H) *rec_H = ig.rec_H;
    if (rec_W) *rec_W = ig.rec_W;
    return SPIHT_OK;
}

static int upload_mults(spiht_ctx *ctx, const double *channel_mults, int64_t c, const double **d_mults) {
    *d_mults = nullptr;
    if (!channel_mults) return SPIHT_OK;
    // the same scales as last time (every call of a codec object): nothing to upload, nothing to wait for
    if (ctx->mults.p && ctx->mults_host.size() == (size_t)c && !memcmp(ctx->mults_host.data(), channel_mults, (size_t)c * 8)) {
        *d_mults = (const double *)ctx->mults.p;
        return SPIHT_OK;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));  // queued kernels still read the old scales
    ctx->mults_host.clear();
    CHK(ensure(ctx, ctx->mults, (size_t)c * 8));
    HIPCHK(hipMemcpyAsync(ctx->mults.p, channel_mults, (size_t)c * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // caller's array may be a temporary
    ctx->mults_host.assign(channel_mults, channel_mults + c);
    *d_mults = (const double *)ctx->mults.p;
    return SPIHT_OK;
}

// pixels [planes,H,W] -> quantised packed array [planes,enc_h,enc_w]
// f32: d_img holds float pixels and the transform runs in single precision, as PyWavelets does for float32 / float16
// input (every level's input must then be at least as long as the filter: SPIHT_ERR_ARG otherwise)
// The wavelet's filters on the device for the two-pass levels: dec_lo, dec_hi, rec_lo, rec_hi (F doubles each), then dec_lo,
// dec_hi as the single-precision transform has them (F floats each).  Uploaded when the wavelet changes.
static int upload_filters(spiht_ctx *ctx, int wavelet, const double **d_filt) {
    const WaveletDef &wv = SPIHT_WAVELETS[wavelet];
    const size_t F = (size_t)wv.F, bytes = 4 * F * 8 + 2 * F * 4;
    if (ctx->filt_wavelet != wavelet || !ctx->filt.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));  // queued kernels still read the old filters
        ctx->filt_wavelet = -1;
        CHK(ensure(ctx, ctx->filt, bytes));
        std::vector<char> h(bytes);
        memcpy(h.data(), wv.dec_lo, F * 8);
        memcpy(h.data() + F * 8, wv.dec_hi, F * 8);
        memcpy(h.data() + 2 * F * 8, wv.rec_lo, F * 8);
        memcpy(h.data() + 3 * F * 8, wv.rec_hi, F * 8);
        memcpy(h.data() + 4 * F * 8, wv.dec_lo_f, F * 4);
        memcpy(h.data() + 4 * F * 8 + F * 4, wv.dec_hi_f, F * 4);
        HIPCHK(hipMemcpyAsync(ctx->filt.p, h.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // (h is a temporary)
        ctx->filt_wavelet = wavelet;
    }
    *d_filt = (const double *)ctx->filt.p;
    return SPIHT_OK;
}

static int dwt_forward(spiht_ctx *ctx, const double *d_img, int planes, int c, const ImgGeom &ig, int wavelet, int mode,
                       double q, const double *d_mults, int32_t *d_coeffs, uint32_t *d_maxabs = nullptr, bool f32 = false) {
    const WaveletDef &wv = SPIHT_WAVELETS[wavelet];
    const size_t plane_out = (size_t)ig.enc_h * ig.enc_w;
    if (f32 && ig.L == 0) return SPIHT_ERR_ARG;
    bool color = ctx->color_on && c == 3;
    if (color && f32) return SPIHT_ERR_ARG;  // the colour model change is float64 (as colour-science's)
    // the plain two-pass level: the modes that compute their extension, periodization, and filters longer than the tiled
    // kernels take (db11.., sym11.., coif4.., dmey)
    const bool twopass = mode >= SPIHT_MODE_SMOOTH || wv.F > SPIHT_MAX_TAPS;
    const double *d_filt = nullptr;
    if (twopass && ig.L > 0) CHK(upload_filters(ctx, wavelet, &d_filt));
    if (color && twopass && ig.L > 0) {
        // the two-pass level has no colour form: the colour model change as a pass of its own in front of it
        const size_t npix = (size_t)ig.hs[0] * ig.ws[0];
        CHK(ensure(ctx, ctx->img, (size_t)planes * npix * 8));
        LAUNCHCHK(spiht_launch_color3(d_img, (double *)ctx->img.p, planes / 3, npix, ctx->col_fwd.A, ctx->col_fwd.M, ctx->col_fwd.p,
                                      ctx->stream));
        d_img = (const double *)ctx->img.p;
        color = false;
    }
    if (ig.L == 0) {
        StageTimer t(ctx, ST_DWT_REST);
        if (color) {  // no transform level to carry the colour model change: a pass of its own
            CHK(ensure(ctx, ctx->a0, (size_t)planes * plane_out * 8));
            LAUNCHCHK(spiht_launch_color3(d_img, (double *)ctx->a0.p, planes / 3, plane_out, ctx->col_fwd.A, ctx->col_fwd.M,
                                          ctx->col_fwd.p, ctx->stream));
            d_img = (const double *)ctx->a0.p;
        }
        LAUNCHCHK(spiht_launch_quant_plain(d_img, d_coeffs, plane_out, planes, c, d_mults, q, d_maxabs, ctx->stream));
        return SPIHT_OK;
    }
    // zero padding cells of coeffs_to_array (thin strips; every other cell is written by a band).  The transform writes
    // band cells only, so an array it has filled once for this geometry still has its zeros the next time -- PROVIDED
    // nobody else writes into it: a caller that owns its arrays says so (option "pads_persist": the pipeline's
    // double-buffered arrays; beside a list decoder this launch of thin strips took 0.74 instead of 0.12 ms per step).
    bool pads_known = false;
    // (the mode belongs to the key: periodization packs the bands by another length rule, i.e. other strips)
    const spiht_ctx::PadKey key = {d_coeffs, planes, ig.hs[0], ig.ws[0], wv.F, ig.L, mode == SPIHT_MODE_PERIODIZATION ? 1 : 0};
    if (ctx->opt_pads_persist)
        for (const auto &k : ctx->pads_zeroed)
            pads_known = pads_known || (k.p == key.p && k.planes == key.planes && k.H == key.H && k.W == key.W && k.F == key.F && k.L == key.L &&
                                        k.mode == key.mode);
    if (!pads_known) {
        StageTimer t(ctx, ST_MEMSET);
        LAUNCHCHK(spiht_launch_zero_pads(ig.L, ig.hs, ig.ws, ig.offh, ig.offw, (int)ig.enc_h, (int)ig.enc_w, d_coeffs, planes,
                                         ctx->stream));
        // (an array seen with another geometry before has other strips: forget it -- whether or not this call may rely
        // on the promise, so that a later one that does never finds a record older than the array's last layout)
        auto &v = ctx->pads_zeroed;
        v.erase(std::remove_if(v.begin(), v.end(), [&](const spiht_ctx::PadKey &k) { return k.p == key.p; }), v.end());
        if (ctx->opt_pads_persist) {
            if (v.size() >= 8) v.erase(v.begin());
            v.push_back(key);
        }
    }
    if (ig.L >= 2) {
        CHK(ensure(ctx, ctx->a0, (size_t)planes * ig.hs[1] * ig.ws[1] * 8));
        if (ig.L >= 3) CHK(ensure(ctx, ctx->a1, (size_t)planes * ig.hs[2] * ig.ws[2] * 8));
    }
    const double *in = d_img;
    for (int l = 1; l <= ig.L; l++) {
        DwtKArgs a;
        memset(&a, 0, sizeof(a));
        a.c = c;
        a.F = wv.F;
        a.mode = mode;
        a.in_h = (int32_t)ig.hs[l - 1]; a.in_w = (int32_t)ig.ws[l - 1];
        a.out_h = (int32_t)ig.hs[l]; a.out_w = (int32_t)ig.ws[l];
        a.off_h = (int32_t)ig.offh[l]; a.off_w = (int32_t)ig.offw[l];
        a.enc_h = (int32_t)ig.enc_h; a.enc_w = (int32_t)ig.enc_w;
        a.last = (l == ig.L) ? 1 : 0;
        a.f32 = f32 ? 1 : 0;
        a.in = in;
        a.ll_out = a.last ? nullptr : (double *)((l & 1) ? ctx->a0.p : ctx->a1.p);
        a.coeffs = d_coeffs;
        a.mults = d_mults;
        a.maxabs = d_maxabs;
        a.q = q;
        if (color && l == 1) { a.color = 1; a.col = ctx->col_fwd; }
        const int Fc = std::min(wv.F, SPIHT_MAX_TAPS);  // (a longer filter goes by the device copy: two-pass level)
        memcpy(a.lo, wv.dec_lo, sizeof(double) * Fc);
        memcpy(a.hi, wv.dec_hi, sizeof(double) * Fc);
        memcpy(a.lo_f, wv.dec_lo_f, sizeof(float) * Fc);
        memcpy(a.hi_f, wv.dec_hi_f, sizeof(float) * Fc);
        if (twopass) {
            // smooth / antisymmetric / antireflect / periodization / long filters: two plain passes through an intermediate (dwt.hip:
            // k_dwt_axis_ext), a few planes at a time so that the intermediates stay under a gigabyte; in the pixels' precision
            StageTimer t(ctx, l == 1 ? ST_DWT_L1 : ST_DWT_REST);
            const size_t esz = f32 ? 4 : 8;
            const size_t n_t = (size_t)a.out_h * a.in_w, n_b = (size_t)a.out_h * a.out_w;  // elements per plane
            const size_t per_plane = (2 * n_t + 4 * n_b) * esz;
            int pc = (int)std::max<size_t>(1, ((size_t)1 << 30) / per_plane);
            pc = std::max(c, pc / c * c);  // whole images: the max|coefficient| word and the channel scale go by plane / c, plane % c
            CHK(ensure(ctx, ctx->exttmp, per_plane * (size_t)std::min(pc, planes)));
            for (int p0 = 0; p0 < planes; p0 += pc) {
                const int np = std::min(pc, planes - p0);
                DwtKArgs b = a;
                b.in = (const double *)((const char *)a.in + (size_t)p0 * a.in_h * a.in_w * esz);
                if (b.ll_out) b.ll_out = (double *)((char *)a.ll_out + (size_t)p0 * n_b * esz);
                b.coeffs = a.coeffs + (size_t)p0 * a.enc_h * a.enc_w;
                if (b.maxabs) b.maxabs = a.maxabs + p0 / c;
                char *base = (char *)ctx->exttmp.p;
                void *t_lo = base, *t_hi = base + (size_t)np * n_t * esz;
                char *bb = base + 2 * (size_t)np * n_t * esz;
                LAUNCHCHK(spiht_launch_dwt_level_ext(&b, np, t_lo, t_hi, bb, bb + (size_t)np * n_b * esz, bb + 2 * (size_t)np * n_b * esz,
                                                     bb + 3 * (size_t)np * n_b * esz, d_filt, ctx->stream));
            }
        } else {
            StageTimer t(ctx, l == 1 ? ST_DWT_L1 : ST_DWT_REST);
            LAUNCHCHK(spiht_launch_dwt_level(&a, planes, ctx->stream));
        }
        in = a.ll_out;
    }
    return SPIHT_OK;
}

// packed int32 array -> pixels [planes, rec_H, rec_W]
// l_hi .. l_lo: the levels to run, coarsest first (ig.L .. 1 = all).  A run that stops above level 1 leaves its last
// approximation in d_out ([planes, 2*hs[l_lo]-F+2, 2*ws[l_lo]-F+2]); a run that starts below ig.L takes that array as
// d_a_in.
// Geometry of the L1Flags words of an image geometry (common.h): false when they do not apply (fewer than two levels: the
// level-1 approximation then comes out of the packed array too; a filter whose halo exceeds a tile)
static bool l1flags_geometry(const ImgGeom &ig, int F, L1Flags *fl) {
    memset(fl, 0, sizeof(*fl));
    if (ig.per) return false;  // (periodization runs the plain two-pass inverse: no tiles)
    if (ig.L < 2 || F / 2 - 1 > IW_TH / 2 || F / 2 - 1 > IW_TW / 2) return false;
    fl->off_h = (int32_t)ig.offh[1]; fl->off_w = (int32_t)ig.offw[1];
    fl->band_h = (int32_t)ig.hs[1]; fl->band_w = (int32_t)ig.ws[1];
    fl->hf1 = F / 2 - 1;
    fl->gx = (int32_t)((2 * ig.ws[1] - F + 2 + IW_TW - 1) / IW_TW);
    fl->gy = (int32_t)((2 * ig.hs[1] - F + 2 + IW_TH - 1) / IW_TH);
    return fl->gx > 0 && fl->gy > 0;
}

// d_flags: L1Flags words [planes, gy, gx] the decoder of d_rec left (nullptr: every level-1 tile reads its detail bands)
static int dwt_inverse(spiht_ctx *ctx, const int32_t *d_rec, int planes, int c, const ImgGeom &ig, int wavelet, double q,
                       const double *d_mults, double *d_out, int l_hi = -1, int l_lo = 1, const double *d_a_in = nullptr,
                       const uint32_t *d_flags = nullptr) {
    const WaveletDef &wv = SPIHT_WAVELETS[wavelet];
    const int F = wv.F;
    if (l_hi < 0) l_hi = ig.L;
    const bool color = ctx->color_on && c == 3;
    if (ig.L == 0) {
        StageTimer t(ctx, ST_IDWT_REST);
        LAUNCHCHK(spiht_launch_dequant_plain(d_rec, d_out, (size_t)ig.enc_h * ig.enc_w, planes, c, d_mults, q, ctx->stream));
        if (color)
            LAUNCHCHK(spiht_launch_color3(d_out, d_out, planes / 3, (size_t)ig.enc_h * ig.enc_w, ctx->col_inv.A, ctx->col_inv.M,
                                          ctx->col_inv.p, ctx->stream));
        return SPIHT_OK;
    }
    // intermediate approximations ping-pong between a0/a1; sizes 2*band-F+2
    size_t maxa = 0;
    const int Fg = ig.per ? 2 : F;  // (the length rule's filter length: periodization gives back 2 n samples)
    for (int l = l_hi; l > l_lo; l--) maxa = std::max(maxa, (size_t)(2 * ig.hs[l] - Fg + 2) * (size_t)(2 * ig.ws[l] - Fg + 2));
    if (maxa) {
        CHK(ensure(ctx, ctx->a0, maxa * planes * 8));
        CHK(ensure(ctx, ctx->a1, maxa * planes * 8));
    }
    const double *a_in = d_a_in;
    int64_t ah = ig.ll_h, aw = ig.ll_w;
    if (l_hi < ig.L) { ah = 2 * ig.hs[l_hi + 1] - Fg + 2; aw = 2 * ig.ws[l_hi + 1] - Fg + 2; }
    for (int l = l_hi; l >= l_lo; l--) {
        IdwtKArgs a;
        memset(&a, 0, sizeof(a));
        a.c = c;
        a.F = F;
        a.band_h = (int32_t)ig.hs[l]; a.band_w = (int32_t)ig.ws[l];
        a.out_h = (int32_t)(2 * ig.hs[l] - F + 2); a.out_w = (int32_t)(2 * ig.ws[l] - F + 2);
        a.a_h = (int32_t)ah; a.a_w = (int32_t)aw;
        // waverec2 trim rule: the running approximation may be exactly one longer than the band
        if (!((ah == ig.hs[l] || ah == ig.hs[l] + 1) && (aw == ig.ws[l] || aw == ig.ws[l] + 1))) return SPIHT_ERR_ARG;
        if (ig.per) { a.out_h = (int32_t)(2 * ig.hs[l]); a.out_w = (int32_t)(2 * ig.ws[l]); }
        a.off_h = (int32_t)ig.offh[l]; a.off_w = (int32_t)ig.offw[l];
        a.enc_h = (int32_t)ig.enc_h; a.enc_w = (int32_t)ig.enc_w;
        a.first = (l == ig.L) ? 1 : 0;
        a.a_in = a_in;
        a.rec = d_rec;
        a.out = (l == l_lo) ? d_out : (double *)((l & 1) ? ctx->a1.p : ctx->a0.p);
        a.mults = d_mults;
        a.q = q;
        if (color && l == 1) { a.color = 1; a.col = ctx->col_inv; }
        if (l == 1 && !a.first && !a.color) a.flags = d_flags;
        memcpy(a.lo, wv.rec_lo, sizeof(double) * std::min(F, SPIHT_MAX_TAPS));
        memcpy(a.hi, wv.rec_hi, sizeof(double) * std::min(F, SPIHT_MAX_TAPS));
        if (ig.per || F > SPIHT_MAX_TAPS) {
            // periodization / a filter longer than the tiled kernels take: two plain passes through an intermediate (dwt.hip:
            // k_idwt_axis_per), a few planes at a time; the colour model of the picture as a pass of its own behind level 1
            const double *d_filt;
            CHK(upload_filters(ctx, wavelet, &d_filt));
            StageTimer t(ctx, l == 1 ? ST_IDWT_L1 : ST_IDWT_REST);
            a.color = 0;
            const size_t per_plane = (size_t)2 * a.band_h * a.out_w * 8;
            int pc = (int)std::max<size_t>(1, ((size_t)1 << 30) / per_plane);
            pc = std::max(c, pc / c * c);
            CHK(ensure(ctx, ctx->exttmp, per_plane * (size_t)std::min(pc, planes)));
            for (int p0 = 0; p0 < planes; p0 += pc) {
                const int np = std::min(pc, planes - p0);
                IdwtKArgs b = a;
                if (b.a_in) b.a_in = a.a_in + (size_t)p0 * a.a_h * a.a_w;
                b.rec = a.rec + (size_t)p0 * a.enc_h * a.enc_w;
                b.out = a.out + (size_t)p0 * a.out_h * a.out_w;
                double *t_lo = (double *)ctx->exttmp.p, *t_hi = t_lo + (size_t)np * a.band_h * a.out_w;
                LAUNCHCHK(spiht_launch_idwt_level_per(&b, np, t_lo, t_hi, d_filt, ig.per ? 1 : 0, ctx->stream));
            }
            if (color && l == 1)
                LAUNCHCHK(spiht_launch_color3(a.out, a.out, planes / 3, (size_t)a.out_h * a.out_w, ctx->col_inv.A, ctx->col_inv.M,
                                              ctx->col_inv.p, ctx->stream));
        } else {
            StageTimer t(ctx, l == 1 ? ST_IDWT_L1 : ST_IDWT_REST);
            LAUNCHCHK(spiht_launch_idwt_level(&a, planes, ctx->stream, ctx->tilectr.dev ? &ctx->tilectr : nullptr));
        }
        a_in = a.out;
        ah = a.out_h;
        aw = a.out_w;
    }
    return SPIHT_OK;
}

static int check_img_args(int wavelet, int mode, int64_t B, int64_t c, int64_t H, int64_t W) {
    if (wavelet < 0 || wavelet >= SPIHT_NWAVELETS || mode < 0 || mode > SPIHT_MODE_PERIODIZATION) return SPIHT_ERR_ARG;
    if (B < 0 || c < 1 || H < 1 || W < 1) return SPIHT_ERR_ARG;
    if (H > (1 << 24) || W > (1 << 24)) return SPIHT_ERR_TOO_LARGE;
    return SPIHT_OK;
}

static int dwt_quant_batch(spiht_ctx *ctx, const void *d_img_v, bool f32, int64_t B, int64_t c, int64_t H, int64_t W,
                           int wavelet, int mode, int level, double q_scale, const double *channel_mults, int32_t *d_coeffs) {
    const double *d_img = (const double *)d_img_v;
    const size_t esz = f32 ? 4 : 8;
    if (!ctx || !d_img || !d_coeffs) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int chunk = (int)std::max<int64_t>(1, 65535 / c);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        CHK(dwt_forward(ctx, (const double *)((const char *)d_img + (size_t)b0 * c * H * W * esz), nb * (int)c, (int)c, ig, wavelet,
                        mode, q_scale, d_mults, d_coeffs + (size_t)b0 * c * ig.enc_h * ig.enc_w, nullptr, f32));
    }
    return SPIHT_OK;
}
extern "C" int spiht_dwt_quant_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                         int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                         int32_t *d_coeffs) {
    return dwt_quant_batch(ctx, d_img, false, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, d_coeffs);
}
extern "C" int spiht_dwt_quant_batch_f32(spiht_ctx *ctx, const float *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                         int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                         int32_t *d_coeffs) {
    return dwt_quant_batch(ctx, d_img, true, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, d_coeffs);
}

static int dequant_idwt_batch(spiht_ctx *ctx, const int32_t *d_rec, const uint32_t *d_flags, int64_t B, int64_t c, int64_t H,
                              int64_t W, int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                              double *d_img_out) {
    if (!ctx || !d_rec || !d_img_out) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int chunk = (int)std::max<int64_t>(1, 65535 / c);
    L1Flags fl;
    const bool flagged = d_flags && !(ctx->color_on && c == 3) && l1flags_geometry(ig, SPIHT_WAVELETS[wavelet].F, &fl);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        CHK(dwt_inverse(ctx, d_rec + (size_t)b0 * c * ig.enc_h * ig.enc_w, nb * (int)c, (int)c, ig, wavelet, q_scale,
                        d_mults, d_img_out + (size_t)b0 * c * ig.rec_H * ig.rec_W, -1, 1, nullptr,
                        flagged ? d_flags + (size_t)b0 * c * fl.gy * fl.gx : nullptr));
    }
    return SPIHT_OK;
}
extern "C" int spiht_dequant_idwt_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, int64_t B, int64_t c, int64_t H,
                                            int64_t W, int wavelet, int mode, int level, double q_scale,
                                            const double *channel_mults, double *d_img_out) {
    return dequant_idwt_batch(ctx, d_rec, nullptr, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, d_img_out);
}
extern "C" int spiht_dequant_idwt_flags_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const uint32_t *d_flags, int64_t B,
                                                  int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level,
                                                  double q_scale, const double *channel_mults, double *d_img_out) {
    return dequant_idwt_batch(ctx, d_rec, d_flags, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, d_img_out);
}

// The inverse transform in two parts, so that a pipelined caller can queue the coarse levels (a quarter of the bytes,
// six small launches at 1080p) where no list decoder shares the GPU and only level 1 beside it (csrc/pipeline.cpp):
//   coarse: levels level..2 -> d_approx [B*c, 2*hs[2]-F+2, 2*ws[2]-F+2] (float64), the approximation level 1 starts from
//   level1: d_rec + d_approx -> pixels.  With fewer than two levels the coarse part does nothing and d_approx is not read.
static int idwt_part(spiht_ctx *ctx, const int32_t *d_rec, double *d_approx, int64_t B, int64_t c, int64_t H, int64_t W,
                     int wavelet, int mode, int level, double q_scale, const double *channel_mults, double *d_img_out,
                     const uint32_t *d_flags = nullptr) {
    if (!ctx || !d_rec || (!d_approx && !d_img_out)) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    if (ig.per) return SPIHT_ERR_ARG;  // (the two-part inverse is a schedule experiment of the tiled kernels)
    const bool coarse = d_img_out == nullptr;
    if (coarse && ig.L < 2) return SPIHT_OK;
    if (!coarse && ig.L >= 2 && !d_approx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int F = SPIHT_WAVELETS[wavelet].F;
    const size_t a_plane = ig.L >= 2 ? (size_t)(2 * ig.hs[2] - F + 2) * (size_t)(2 * ig.ws[2] - F + 2) : 0;
    const int chunk = (int)std::max<int64_t>(1, 65535 / c);
    L1Flags fl;
    const bool flagged = d_flags && !(ctx->color_on && c == 3) && l1flags_geometry(ig, F, &fl);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (int)std::min<int64_t>(chunk, B - b0);
        const int32_t *rec = d_rec + (size_t)b0 * c * ig.enc_h * ig.enc_w;
        double *ap = d_approx ? d_approx + (size_t)b0 * c * a_plane : nullptr;
        if (coarse)
            CHK(dwt_inverse(ctx, rec, nb * (int)c, (int)c, ig, wavelet, q_scale, d_mults, ap, ig.L, 2, nullptr));
        else
            CHK(dwt_inverse(ctx, rec, nb * (int)c, (int)c, ig, wavelet, q_scale, d_mults,
                            d_img_out + (size_t)b0 * c * ig.rec_H * ig.rec_W, std::min(ig.L, 1), 1, ig.L >= 2 ? ap : nullptr,
                            flagged ? d_flags + (size_t)b0 * c * fl.gy * fl.gx : nullptr));
    }
    return SPIHT_OK;
}
extern "C" int spiht_idwt_coarse_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, int64_t B, int64_t c, int64_t H, int64_t W,
                                           int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                           double *d_approx) {
    if (!d_approx) return SPIHT_ERR_ARG;
    return idwt_part(ctx, d_rec, d_approx, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, nullptr);
}
extern "C" int spiht_idwt_level1_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const double *d_approx, int64_t B, int64_t c,
                                           int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                           const double *channel_mults, double *d_img_out) {
    if (!d_img_out) return SPIHT_ERR_ARG;
    return idwt_part(ctx, d_rec, const_cast<double *>(d_approx), B, c, H, W, wavelet, mode, level, q_scale, channel_mults,
                     d_img_out);
}

// ... reading the decoder's occupancy words of the level-1 tiles (spiht_decode_lists_flags_batch_i32; NULL: reads everything)
extern "C" int spiht_idwt_level1_flags_batch_f64(spiht_ctx *ctx, const int32_t *d_rec, const double *d_approx, const uint32_t *d_flags,
                                                 int64_t B, int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level,
                                                 double q_scale, const double *channel_mults, double *d_img_out) {
    if (!d_img_out) return SPIHT_ERR_ARG;
    return idwt_part(ctx, d_rec, const_cast<double *>(d_approx), B, c, H, W, wavelet, mode, level, q_scale, channel_mults,
                     d_img_out, d_flags);
}

extern "C" int spiht_idwt_approx_shape(int64_t H, int64_t W, int wavelet, int level, int64_t *a_h, int64_t *a_w) {
    if (wavelet < 0 || wavelet >= SPIHT_NWAVELETS || !a_h || !a_w) return SPIHT_ERR_ARG;
    ImgGeom ig;
    const int F = SPIHT_WAVELETS[wavelet].F;
    CHK(img_geometry(H, W, F, level, &ig));
    *a_h = ig.L >= 2 ? 2 * ig.hs[2] - F + 2 : 0;
    *a_w = ig.L >= 2 ? 2 * ig.ws[2] - F + 2 : 0;
    return SPIHT_OK;
}

static int encode_image_batch(spiht_ctx *ctx, const void *d_img_v, bool f32, int64_t B, int64_t c, int64_t H, int64_t W,
                              int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                              uint64_t max_bits, uint8_t *d_out, uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_max_n,
                              int32_t *d_coeffs) {
    const double *d_img = (const double *)d_img_v;
    const size_t esz = f32 ? 4 : 8;
    if (!ctx || !d_img || !d_out || !d_nbits || !d_max_n) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    Geom g;
    CHK(make_geom(c, ig.enc_h, ig.enc_w, ig.ll_h, ig.ll_w, &g));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int chunk = batch_chunk(g);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        int32_t *co = d_coeffs ? d_coeffs + (size_t)b0 * g.n : nullptr;
        if (!co) {
            CHK(ensure(ctx, ctx->coeffs, (size_t)nb * g.n * 4));
            co = (int32_t *)ctx->coeffs.p;
        }
        CHK(ensure(ctx, ctx->maxabs, (size_t)nb * 4));
        HIPCHK(hipMemsetAsync(ctx->maxabs.p, 0, (size_t)nb * 4, ctx->stream));
        CHK(dwt_forward(ctx, (const double *)((const char *)d_img + (size_t)b0 * c * H * W * esz), nb * (int)c, (int)c, ig, wavelet,
                        mode, q_scale, d_mults, co, (uint32_t *)ctx->maxabs.p, f32));
        CHK(encode_device(ctx, g, co, nb, max_bits, d_out + (size_t)b0 * slot_stride, slot_stride, d_nbits + b0,
                          d_max_n + b0, true));
    }
    return SPIHT_OK;  // asynchronous: errors surface in spiht_ctx_synchronize()
}
extern "C" int spiht_encode_image_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H,
                                            int64_t W, int wavelet, int mode, int level, double q_scale,
                                            const double *channel_mults, uint64_t max_bits, uint8_t *d_out,
                                            uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_max_n,
                                            int32_t *d_coeffs) {
    return encode_image_batch(ctx, d_img, false, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, d_out,
                              slot_stride, d_nbits, d_max_n, d_coeffs);
}
extern "C" int spiht_encode_image_batch_f32(spiht_ctx *ctx, const float *d_img, int64_t B, int64_t c, int64_t H,
                                            int64_t W, int wavelet, int mode, int level, double q_scale,
                                            const double *channel_mults, uint64_t max_bits, uint8_t *d_out,
                                            uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_max_n,
                                            int32_t *d_coeffs) {
    return encode_image_batch(ctx, d_img, true, B, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, d_out,
                              slot_stride, d_nbits, d_max_n, d_coeffs);
}

extern "C" int spiht_decode_image_batch_f64(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride,
                                            const uint64_t *d_nbytes, const uint8_t *d_max_n, int64_t B, int64_t c,
                                            int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                            const double *channel_mults, double *d_img_out, int32_t *d_rec) {
    if (!ctx || !d_data || !d_nbytes || !d_max_n || !d_img_out) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    Geom g;
    CHK(make_geom(c, ig.enc_h, ig.enc_w, ig.ll_h, ig.ll_w, &g));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int chunk = batch_chunk(g);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        int32_t *rec = d_rec ? d_rec + (size_t)b0 * g.n : nullptr;
        // the decoder tells the inverse transform which level-1 tiles hold anything (common.h: L1Flags)
        L1Flags fl;
        const bool flagged = ctx->opt_l1_flags && !(ctx->color_on && c == 3) && l1flags_geometry(ig, SPIHT_WAVELETS[wavelet].F, &fl);
        if (flagged) {
            CHK(ensure(ctx, ctx->l1flags, (size_t)nb * c * fl.gy * fl.gx * 4));
            fl.p = (uint32_t *)ctx->l1flags.p;
        }
        if (rec) {
            CHK(decode_device(ctx, g, d_data + (size_t)b0 * slot_stride, slot_stride, d_nbytes + b0, d_max_n + b0, nb, rec,
                              nullptr, nullptr, 0, true, nullptr, flagged ? &fl : nullptr));
            CHK(dwt_inverse(ctx, rec, nb * (int)c, (int)c, ig, wavelet, q_scale, d_mults,
                            d_img_out + (size_t)b0 * c * ig.rec_H * ig.rec_W, -1, 1, nullptr, flagged ? fl.p : nullptr));
            continue;
        }
        // Internal coefficient array: it is all zero on entry and is left all zero -- after the inverse transform the
        // cells the decoder wrote are cleared again through its own LSP lists (about 1 % of the array) instead of
        // zero-filling 26 MB per 1080p image before every decode.
        const size_t old_cap = ctx->recz.cap;  // (the allocator may hand back the same address for a larger buffer)
        CHK(ensure(ctx, ctx->recz, (size_t)nb * g.n * 4));
        if (ctx->recz.cap != old_cap || !ctx->recz_clean) {
            StageTimer t(ctx, ST_MEMSET);
            HIPCHK(hipMemsetAsync(ctx->recz.p, 0, ctx->recz.cap, ctx->stream));
            ctx->recz_clean = true;
        }
        rec = (int32_t *)ctx->recz.p;
        DecArgs da;
        CHK(decode_device(ctx, g, d_data + (size_t)b0 * slot_stride, slot_stride, d_nbytes + b0, d_max_n + b0, nb, rec, nullptr,
                          nullptr, 0, false, &da, flagged ? &fl : nullptr));
        CHK(dwt_inverse(ctx, rec, nb * (int)c, (int)c, ig, wavelet, q_scale, d_mults,
                        d_img_out + (size_t)b0 * c * ig.rec_H * ig.rec_W, -1, 1, nullptr, flagged ? fl.p : nullptr));
        if (da.nslots >= nb) {
            StageTimer t(ctx, ST_MEMSET);
            LAUNCHCHK(spiht_launch_unscatter(&da, ctx->stream));
        } else {
            ctx->recz_clean = false;  // slots were reused inside the launch: the lists of earlier images are gone
        }
    }
    return SPIHT_OK;  // asynchronous: errors surface in spiht_ctx_synchronize()
}

// ------------------------------------------------------------------------------------------------
// the drop-in calls on host arrays: encode_image / decode_image / decode_from_rec_arr of the reference's wrapper
// (spiht_wrapper.py:142-216, 259-281) as one C call each.  Pixels, stream and coefficient array live in the
// context's grow-only device buffers -- no allocation per call after the first of a given size.
// ------------------------------------------------------------------------------------------------
static int encode_image_host(spiht_ctx *ctx, const void *img, bool f32, int64_t c, int64_t H, int64_t W, int wavelet,
                             int mode, int level, double q_scale, const double *channel_mults, uint64_t max_bits,
                             uint8_t *out, uint64_t out_cap, uint64_t *out_nbits, uint8_t *max_n) {
    if (!ctx || !img || !out_nbits || !max_n || (!out && out_cap)) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, 1, c, H, W));
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    Geom g;
    CHK(make_geom(c, ig.enc_h, ig.enc_w, ig.ll_h, ig.ll_w, &g));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t bits = bound_bits(g, 0x3FFFFFFFu);
    if (max_bits != 0) bits = std::min(bits, max_bits);
    if (bits >= 0xFFFFFF00ull * 8ull) return SPIHT_ERR_TOO_LARGE;
    const uint64_t slot = std::max<uint64_t>(4, ((bits + 7) / 8 + 3) & ~3ull);
    const size_t img_bytes = (size_t)c * H * W * (f32 ? 4 : 8);
    CHK(ensure(ctx, ctx->himg, img_bytes));
    CHK(ensure(ctx, ctx->out, slot));
    CHK(ensure(ctx, ctx->nbits, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemcpyAsync(ctx->himg.p, img, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    CHK(encode_image_batch(ctx, ctx->himg.p, f32, 1, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits,
                           (uint8_t *)ctx->out.p, slot, (uint64_t *)ctx->nbits.p, (uint8_t *)ctx->maxn.p, nullptr));
    uint64_t nbits = 0;
    uint8_t mn = 0;
    HIPCHK(hipMemcpyAsync(&nbits, ctx->nbits.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&mn, ctx->maxn.p, 1, hipMemcpyDeviceToHost, ctx->stream));
    CHK(read_err(ctx));  // synchronises
    *out_nbits = nbits;
    *max_n = mn;
    const uint64_t nbytes = (nbits + 7) / 8;
    if (nbytes > out_cap) return SPIHT_ERR_CAPACITY;
    if (nbytes) {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(out, ctx->out.p, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}
extern "C" int spiht_encode_image_host_f64(spiht_ctx *ctx, const double *img, int64_t c, int64_t H, int64_t W, int wavelet,
                                           int mode, int level, double q_scale, const double *channel_mults,
                                           uint64_t max_bits, uint8_t *out, uint64_t out_cap, uint64_t *out_nbits,
                                           uint8_t *max_n) {
    return encode_image_host(ctx, img, false, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, out, out_cap,
                             out_nbits, max_n);
}
extern "C" int spiht_encode_image_host_f32(spiht_ctx *ctx, const float *img, int64_t c, int64_t H, int64_t W, int wavelet,
                                           int mode, int level, double q_scale, const double *channel_mults,
                                           uint64_t max_bits, uint8_t *out, uint64_t out_cap, uint64_t *out_nbits,
                                           uint8_t *max_n) {
    return encode_image_host(ctx, img, true, c, H, W, wavelet, mode, level, q_scale, channel_mults, max_bits, out, out_cap,
                             out_nbits, max_n);
}

extern "C" int spiht_decode_image_host_f64(spiht_ctx *ctx, const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c,
                                           int64_t H, int64_t W, int wavelet, int mode, int level, double q_scale,
                                           const double *channel_mults, double *img_out) {
    if (!ctx || !img_out || (!data && nbytes)) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, 1, c, H, W));
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    if (n > 30) return SPIHT_ERR_MAGNITUDE;
    if (nbytes * 8 >= 0xFFFFFF00ull) return SPIHT_ERR_TOO_LARGE;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t slot = std::max<uint64_t>(4, (nbytes + 3) & ~3ull);
    const size_t out_bytes = (size_t)c * ig.rec_H * ig.rec_W * 8;
    CHK(ensure(ctx, ctx->data, slot));
    CHK(ensure(ctx, ctx->nbytes, 8));
    CHK(ensure(ctx, ctx->maxn, 4));
    CHK(ensure(ctx, ctx->himg, out_bytes));
    CHK(clear_err(ctx));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemsetAsync((char *)ctx->data.p + (slot - 4), 0, 4, ctx->stream));  // the bytes past the stream in its last word
        if (nbytes) HIPCHK(hipMemcpyAsync(ctx->data.p, data, nbytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->nbytes.p, &nbytes, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->maxn.p, &n, 1, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // &nbytes / &n are stack temporaries
    }
    CHK(spiht_decode_image_batch_f64(ctx, (const uint8_t *)ctx->data.p, slot, (const uint64_t *)ctx->nbytes.p,
                                     (const uint8_t *)ctx->maxn.p, 1, c, H, W, wavelet, mode, level, q_scale, channel_mults,
                                     (double *)ctx->himg.p, nullptr));
    CHK(read_err(ctx));
    {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(img_out, ctx->himg.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

extern "C" int spiht_dequant_idwt_host_f64(spiht_ctx *ctx, const int32_t *rec, int64_t c, int64_t H, int64_t W, int wavelet,
                                           int mode, int level, double q_scale, const double *channel_mults,
                                           double *img_out) {
    if (!ctx || !rec || !img_out) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, 1, c, H, W));
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t rec_bytes = (size_t)c * ig.enc_h * ig.enc_w * 4, out_bytes = (size_t)c * ig.rec_H * ig.rec_W * 8;
    CHK(ensure(ctx, ctx->hrec, rec_bytes));
    CHK(ensure(ctx, ctx->himg, out_bytes));
    {
        StageTimer t(ctx, ST_H2D);
        HIPCHK(hipMemcpyAsync(ctx->hrec.p, rec, rec_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    CHK(spiht_dequant_idwt_batch_f64(ctx, (const int32_t *)ctx->hrec.p, 1, c, H, W, wavelet, mode, level, q_scale,
                                     channel_mults, (double *)ctx->himg.p));
    {
        StageTimer t(ctx, ST_D2H);
        HIPCHK(hipMemcpyAsync(img_out, ctx->himg.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// the two halves of each direction on their own, so a caller can run the HBM-bound half of one batch on one
// context while another context list-codes a different batch (bench.py)
// ------------------------------------------------------------------------------------------------

extern "C" int spiht_dwt_pyramid_batch_f64(spiht_ctx *ctx, const double *d_img, int64_t B, int64_t c, int64_t H, int64_t W,
                                           int wavelet, int mode, int level, double q_scale, const double *channel_mults,
                                           int32_t *d_coeffs, uint8_t *d_dmsb, uint8_t *d_lmsb, uint32_t *d_maxabs) {
    if (!ctx || !d_img || !d_coeffs || !d_maxabs || (!d_dmsb) != (!d_lmsb)) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    if (B == 0) return SPIHT_OK;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    Geom g;
    CHK(make_geom(c, ig.enc_h, ig.enc_w, ig.ll_h, ig.ll_w, &g));
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const double *d_mults;
    CHK(upload_mults(ctx, channel_mults, c, &d_mults));
    const int chunk = batch_chunk(g);
    HIPCHK(hipMemsetAsync(d_maxabs, 0, (size_t)B * 4, ctx->stream));
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        int32_t *co = d_coeffs + (size_t)b0 * g.n;
        CHK(dwt_forward(ctx, d_img + (size_t)b0 * c * H * W, nb * (int)c, (int)c, ig, wavelet, mode, q_scale, d_mults, co,
                        d_maxabs + b0, false));
        if (!d_dmsb) continue;  // transform + max|coefficient| only: the pyramid is queued elsewhere (spiht_pyramid_batch_i32)
        StageTimer t(ctx, ST_PYRAMID);
        LAUNCHCHK(spiht_launch_pyramid(&g, nb, co, d_dmsb + (size_t)b0 * g.n, d_lmsb + (size_t)b0 * g.n, ctx->stream));
    }
    return SPIHT_OK;
}

extern "C" int spiht_encode_lists_batch_i32(spiht_ctx *ctx, const int32_t *d_x, const uint8_t *d_dmsb,
                                            const uint8_t *d_lmsb, const uint32_t *d_maxabs, int64_t B, int64_t c,
                                            int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, uint64_t max_bits_in,
                                            uint8_t *d_out, uint64_t slot_stride, uint64_t *d_nbits, uint8_t *d_max_n) {
    if (!ctx || !d_x || !d_dmsb || !d_lmsb || !d_maxabs || !d_out || !d_nbits || !d_max_n || B < 0) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    if (B == 0) return SPIHT_OK;
    if (slot_stride % 4 != 0) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t max_bits = max_bits_in == 0 ? SPIHT_MAX_BITS_UNLIMITED : max_bits_in;
    ListCaps caps;
    list_caps(g, std::min<uint64_t>(max_bits, slot_stride * 8), &caps, nullptr);
    const int chunk = batch_chunk(g);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        int nb = (int)std::min<int64_t>(chunk, B - b0);
        int nslots = 0;
        ListPtrs lp;
        CHK(alloc_lists(ctx, caps, std::min(nb, ctx->num_cu), false, &nslots, &lp));
        {
            StageTimer t(ctx, ST_MEMSET);
            HIPCHK(hipMemsetAsync(d_out + (size_t)b0 * slot_stride, 0, (size_t)nb * slot_stride, ctx->stream));
        }
        CHK(encode_lists_device(ctx, g, d_x + (size_t)b0 * g.n, d_dmsb + (size_t)b0 * g.n, d_lmsb + (size_t)b0 * g.n,
                                d_maxabs + b0, nb, max_bits, caps, nslots, lp, d_out + (size_t)b0 * slot_stride, slot_stride,
                                d_nbits + b0, d_max_n + b0));
    }
    return SPIHT_OK;
}

static int decode_lists_batch(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride, const uint64_t *d_nbytes,
                              const uint8_t *d_max_n, int64_t B, const Geom &g, int32_t *d_out_zeroed, const L1Flags *fl) {
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    DecArgs da;
    CHK(decode_device(ctx, g, d_data, slot_stride, d_nbytes, d_max_n, (int)B, d_out_zeroed, nullptr, nullptr, 0, false, &da, fl));
    ctx->last_dec = da;
    ctx->last_dec_valid = da.nslots >= (int)B;  // otherwise slots were reused inside the launch
    return SPIHT_OK;
}
extern "C" int spiht_decode_lists_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride,
                                            const uint64_t *d_nbytes, const uint8_t *d_max_n, int64_t B, int64_t c,
                                            int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, int32_t *d_out_zeroed) {
    if (!ctx || !d_data || !d_nbytes || !d_max_n || !d_out_zeroed || B < 0) return SPIHT_ERR_ARG;
    Geom g;
    CHK(make_geom(c, h, w, ll_h, ll_w, &g));
    return decode_lists_batch(ctx, d_data, slot_stride, d_nbytes, d_max_n, B, g, d_out_zeroed, nullptr);
}
// The same for the coefficient arrays of B images of H x W pixels (the geometry spiht_geometry gives), leaving in d_flags
// (spiht_l1_flags_words(...) x B words; zero-filled by this call) the occupancy of the inverse transform's level-1 tiles
// for spiht_dequant_idwt_flags_batch_f64.  d_flags == NULL or a geometry without flags: as spiht_decode_lists_batch_i32.
extern "C" int spiht_decode_lists_flags_batch_i32(spiht_ctx *ctx, const uint8_t *d_data, uint64_t slot_stride,
                                                  const uint64_t *d_nbytes, const uint8_t *d_max_n, int64_t B, int64_t c,
                                                  int64_t H, int64_t W, int wavelet, int mode, int level, int32_t *d_out_zeroed,
                                                  uint32_t *d_flags) {
    if (!ctx || !d_data || !d_nbytes || !d_max_n || !d_out_zeroed || B < 0) return SPIHT_ERR_ARG;
    CHK(check_img_args(wavelet, mode, B, c, H, W));
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    Geom g;
    CHK(make_geom(c, ig.enc_h, ig.enc_w, ig.ll_h, ig.ll_w, &g));
    L1Flags fl;
    const bool flagged = d_flags && l1flags_geometry(ig, SPIHT_WAVELETS[wavelet].F, &fl);
    if (flagged) fl.p = d_flags;
    return decode_lists_batch(ctx, d_data, slot_stride, d_nbytes, d_max_n, B, g, d_out_zeroed, flagged ? &fl : nullptr);
}

// Puts the zeros back into the array the context's last spiht_decode_lists_batch_i32 scattered into (after its
// consumer, the inverse transform, has read it): clears exactly the cells that call wrote, through the decoder's
// lists, instead of a zero-fill of B*c*h*w*4 bytes before the next decode.  Falls back to that zero-fill when the
// lists are gone (another list-coding call on this context in between, slots reused within the launch, a latched
// device error).  Queued on the context's stream like everything else.
extern "C" int spiht_unscatter_lists_batch_i32(spiht_ctx *ctx, int32_t *d_out, int64_t B, int64_t c, int64_t h, int64_t w) {
    if (!ctx || !d_out || B < 0 || c < 1 || h < 1 || w < 1) return SPIHT_ERR_ARG;
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    StageTimer t(ctx, ST_MEMSET);
    const DecArgs &da = ctx->last_dec;
    if (ctx->last_dec_valid && da.out == d_out && da.B == (int)B && (int64_t)da.g.n == c * h * w) {
        LAUNCHCHK(spiht_launch_unscatter(&da, ctx->stream));
    } else {
        HIPCHK(hipMemsetAsync(d_out, 0, (size_t)B * (size_t)(c * h * w) * 4, ctx->stream));
    }
    return SPIHT_OK;
}

// Width of the list decoder's workgroups on this context.  12 wavefronts (default) walk one stream fastest; 8 take 4 %
// longer alone but leave the HBM-bound kernels that share the CUs with the decoder more room -- the setting of the
// list-coding contexts of the pipelined schedule (csrc/pipeline.cpp).  Same output either way.
// The context's mutex for a SEQUENCE of calls (it is recursive: the calls inside take it again).  What needs it: a
// setting that is state of the context and must hold for exactly the calls of one caller -- the colour model
// (spiht_ctx_set_color3 ... image calls ... clear): without it another thread's call on the same context could run
// between the set and the clear and be coded in the wrong colour model.  Unlock from the thread that locked.
extern "C" int spiht_ctx_lock(spiht_ctx *ctx) {
    if (!ctx) return SPIHT_ERR_ARG;
    ctx->mu.lock();
    return SPIHT_OK;
}
extern "C" int spiht_ctx_unlock(spiht_ctx *ctx) {
    if (!ctx) return SPIHT_ERR_ARG;
    ctx->mu.unlock();
    return SPIHT_OK;
}

// Switches of this library's own making (nothing of the reference): "l1_flags" (default 1) the list decoder flags the occupied level-1 tiles for the inverse transform of the image-level
// decode calls.  Results are the same bits whatever the setting.  Unknown name / value: SPIHT_ERR_ARG.
extern "C" int spiht_ctx_set_option(spiht_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name || value < 0) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const bool b01 = value == 0 || value == 1;
    if (!strcmp(name, "l1_flags") && b01) ctx->opt_l1_flags = value != 0;
    else if (!strcmp(name, "idwt_groups") && value <= 8) ctx->tilectr.wg_per_cu = (int32_t)value;
    else if (!strcmp(name, "wide_encode") && value <= 3) ctx->opt_wide_encode = (int)value;
    else if (!strcmp(name, "wide_groups") && value <= 256) ctx->opt_wide_g = (int)value;
    else if (!strcmp(name, "wide_solo") && value <= (1 << 30)) ctx->opt_wide_solo = (int)value;
    else if (!strcmp(name, "pads_persist") && b01) ctx->opt_pads_persist = value != 0;
    else return SPIHT_ERR_ARG;
    return SPIHT_OK;
}
extern "C" int spiht_ctx_get_option(spiht_ctx *ctx, const char *name, int64_t *value) {
    if (!ctx || !name || !value) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!strcmp(name, "l1_flags")) *value = ctx->opt_l1_flags;
    else if (!strcmp(name, "idwt_groups")) *value = ctx->tilectr.wg_per_cu;
    else if (!strcmp(name, "wide_encode")) *value = ctx->opt_wide_encode;
    else if (!strcmp(name, "wide_groups")) *value = ctx->opt_wide_g;
    else if (!strcmp(name, "wide_solo")) *value = ctx->opt_wide_solo;
    else if (!strcmp(name, "pads_persist")) *value = ctx->opt_pads_persist;
    else if (!strcmp(name, "num_cu")) *value = ctx->num_cu;                    // (read-only: what the device reports)
    else if (!strcmp(name, "lds_per_cu")) *value = ctx->tilectr.lds_per_cu;
    else return SPIHT_ERR_ARG;
    return SPIHT_OK;
}

// The last encode call of this context that ran the several-CUs-per-image encoder: how many images (groups of workgroups)
// it had and how many of those groups gave up for lack of residency and were coded by the single-workgroup kernel behind
// it instead (csrc/encode_wide.hip).  Waits for the context's stream.  No such call yet: 0, 0.
extern "C" int spiht_ctx_wide_stats(spiht_ctx *ctx, uint32_t *groups, uint32_t *gave_up) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int n = ctx->widebuf.p ? ctx->wide_last_groups : 0;
    std::vector<WideCtl> h((size_t)n);
    if (n) HIPCHK(hipMemcpy(h.data(), ctx->widebuf.p, (size_t)n * sizeof(WideCtl), hipMemcpyDeviceToHost));
    uint32_t gu = 0;
    for (const WideCtl &c : h) gu += (c.bad & 2u) ? 1u : 0u;
    if (groups) *groups = (uint32_t)n;
    if (gave_up) *gave_up = gu;
    return SPIHT_OK;
}

// Size, in 32-bit words per image, of the occupancy words the decoder leaves for the inverse transform's level 1
// (common.h: L1Flags): 0 when they do not apply to this geometry (fewer than two levels).
extern "C" int spiht_l1_flags_words(int64_t c, int64_t H, int64_t W, int wavelet, int mode, int level, uint64_t *words_per_image) {
    if (!words_per_image || wavelet < 0 || wavelet >= SPIHT_NWAVELETS || c < 1 || mode < 0 || mode > SPIHT_MODE_PERIODIZATION)
        return SPIHT_ERR_ARG;
    ImgGeom ig;
    CHK(img_geometry(H, W, SPIHT_WAVELETS[wavelet].F, level, &ig, mode));
    L1Flags fl;
    *words_per_image = l1flags_geometry(ig, SPIHT_WAVELETS[wavelet].F, &fl) ? (uint64_t)c * fl.gy * fl.gx : 0;
    return SPIHT_OK;
}

extern "C" int spiht_ctx_set_decoder_waves(spiht_ctx *ctx, int waves) {
    if (!ctx || (waves != 8 && waves != 12)) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->dec_waves = waves;
    return SPIHT_OK;
}

// Colour model of the coded picture, fused into the transform: while set, every image-level entry point of
// this context takes and returns pixels in the caller's colour model (RGB) and codes them in the other one -- level 1 of the
// forward transform converts on its loads (w = M_f * spow(A_f * u, p_f) per pixel), level 1 of the inverse transform on its
// stores (A_i, M_i, p_i: the way back) -- for 3-channel float64 images.  A_f == NULL clears it.  Row-major 3x3 host arrays.
extern "C" int spiht_ctx_set_color3(spiht_ctx *ctx, const double *A_f, const double *M_f, double p_f, const double *A_i,
                                    const double *M_i, double p_i) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!A_f) { ctx->color_on = false; return SPIHT_OK; }
    if (!M_f || !A_i || !M_i) return SPIHT_ERR_ARG;
    for (int i = 0; i < 9; i++) {
        ctx->col_fwd.A[i] = A_f[i]; ctx->col_fwd.M[i] = M_f[i];
        ctx->col_inv.A[i] = A_i[i]; ctx->col_inv.M[i] = M_i[i];
    }
    ctx->col_fwd.p = p_f;
    ctx->col_inv.p = p_i;
    ctx->color_on = true;
    return SPIHT_OK;
}

// what spiht_ctx_set_color3 last set (on == 0: nothing is set and the arrays are left alone); any output pointer may be NULL
extern "C" int spiht_ctx_get_color3(spiht_ctx *ctx, int *on, double *A_f, double *M_f, double *p_f, double *A_i, double *M_i,
                                    double *p_i) {
    if (!ctx || !on) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    *on = ctx->color_on ? 1 : 0;
    if (!ctx->color_on) return SPIHT_OK;
    if (A_f) memcpy(A_f, ctx->col_fwd.A, 72);
    if (M_f) memcpy(M_f, ctx->col_fwd.M, 72);
    if (A_i) memcpy(A_i, ctx->col_inv.A, 72);
    if (M_i) memcpy(M_i, ctx->col_inv.M, 72);
    if (p_f) *p_f = ctx->col_fwd.p;
    if (p_i) *p_i = ctx->col_inv.p;
    return SPIHT_OK;
}

// Colour model change of B three-channel float64 images [B,3,npix] on the device: per pixel w = M * spow(A * u, p) with
// spow(x, p) = sign(x)|x|^p (the RGB <-> IPT shape; A, M row-major 3x3 host arrays).  d_out may equal d_in.
extern "C" int spiht_color3_batch_f64(spiht_ctx *ctx, const double *d_in, double *d_out, int64_t B, int64_t npix,
                                      const double *A, const double *M, double p) {
    if (!ctx || !d_in || !d_out || !A || !M || B < 0 || npix < 1 || B > 65535) return SPIHT_ERR_ARG;
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    LAUNCHCHK(spiht_launch_color3(d_in, d_out, (int)B, (size_t)npix, A, M, p, ctx->stream));
    return SPIHT_OK;
}

// ------------------------------------------------------------------------------------------------
// events: ordering between contexts at a finer grain than spiht_ctx_wait_on
// ------------------------------------------------------------------------------------------------
struct spiht_event {
    int device;
    hipEvent_t ev;
};

// the context's HIP stream (a hipStream_t), for callers that put their own work (e.g. an RCCL collective) in order
// with the library's
extern "C" int spiht_ctx_stream(spiht_ctx *ctx, void **stream) {
    if (!ctx || !stream) return SPIHT_ERR_ARG;
    *stream = (void *)ctx->stream;
    return SPIHT_OK;
}

extern "C" int spiht_event_create(spiht_ctx *ctx, spiht_event **out) {
    if (!ctx || !out) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    spiht_event *e = new spiht_event;
    e->device = ctx->device;
    hipError_t r = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (r != hipSuccess) {
        delete e;
        g_hip_err = std::string("hipEventCreateWithFlags: ") + hipGetErrorString(r);
        return SPIHT_ERR_HIP;
    }
    *out = e;
    return SPIHT_OK;
}
extern "C" void spiht_event_destroy(spiht_event *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipEventDestroy(e->ev);
    delete e;
}
// marks the point reached by the work queued on ctx so far
extern "C" int spiht_event_record(spiht_event *e, spiht_ctx *ctx) {
    if (!e || !ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventRecord(e->ev, ctx->stream));
    return SPIHT_OK;
}
// work queued on ctx after this call starts only when the recorded point has been reached
extern "C" int spiht_ctx_wait_event(spiht_ctx *ctx, spiht_event *e) {
    if (!e || !ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamWaitEvent(ctx->stream, e->ev, 0));
    return SPIHT_OK;
}

extern "C" int spiht_launch_gate(const uint32_t *counter, uint32_t target, uint64_t ticks, hipStream_t st);
// Placement between two contexts' kernels without a timer.  The large levels of the inverse transform run as persistent
// workgroups (a fixed number per CU) that count themselves in as they start; a ticket says what that count will read once
// every such launch queued on `ctx` SO FAR has all its workgroups on the CUs ...
extern "C" int spiht_ctx_resident_ticket(spiht_ctx *ctx, const void **d_counter, uint32_t *target) {
    if (!ctx || !d_counter || !target) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    *d_counter = ctx->tilectr.dev ? (const void *)(ctx->tilectr.dev + TILECTR_STARTED) : nullptr;
    *target = ctx->tilectr.started;
    return SPIHT_OK;
}
// ... and this queues, on another context's stream, a one-wavefront kernel that waits for that count (at most timeout_us
// microseconds, <= 10 000: it must not hold its stream for ever if the launch it waits for never comes): the work queued
// behind it starts when those workgroups are resident.  d_counter NULL: nothing to wait for.
extern "C" int spiht_ctx_wait_resident(spiht_ctx *ctx, const void *d_counter, uint32_t target, uint32_t timeout_us) {
    if (!ctx || timeout_us > 10000u) return SPIHT_ERR_ARG;
    if (!d_counter) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    LAUNCHCHK(spiht_launch_gate((const uint32_t *)d_counter, target, (uint64_t)timeout_us * 2400u, ctx->stream));  // (s_memtime: shader clocks)
    return SPIHT_OK;
}

extern "C" int spiht_nbits_to_nbytes(spiht_ctx *ctx, const uint64_t *d_nbits, int64_t B, uint64_t *d_nbytes) {
    if (!ctx || !d_nbits || !d_nbytes || B < 0) return SPIHT_ERR_ARG;
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    LAUNCHCHK(spiht_launch_nbits_to_nbytes(d_nbits, (int)B, d_nbytes, ctx->stream));
    return SPIHT_OK;
}

#ifdef SPIHT_DIAG  // diagnostics of tools/corun*.py, not part of the product library (tools/build_variant.sh diag -DSPIHT_DIAG)
// diagnostic: copy the device error/debug words (64 x uint32) to the host
extern "C" int spiht_launch_spin(int blocks, int threads, uint64_t ticks, uint32_t lds_bytes, uint32_t *sink, hipStream_t st);
// diagnostic, not part of the ABI header: occupy the GPU with `blocks` idle workgroups for `ticks` clock ticks
extern "C" int spiht_debug_spin(spiht_ctx *ctx, int blocks, int threads, uint64_t ticks, uint32_t lds_bytes) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    LAUNCHCHK(spiht_launch_spin(blocks, threads, ticks, lds_bytes, (uint32_t *)ctx->err.p + 8, ctx->stream));
    return SPIHT_OK;
}

extern "C" int spiht_launch_spin_kind(int blocks, int threads, uint64_t ticks, uint32_t lds_bytes, uint32_t *sink, int mode, hipStream_t st);
// diagnostic: ... with workgroups that do one kind of work (pyramid.hip: k_spin_kind)
extern "C" int spiht_debug_spin_kind(spiht_ctx *ctx, int blocks, int threads, uint64_t ticks, uint32_t lds_bytes, int mode) {
    if (!ctx) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    LAUNCHCHK(spiht_launch_spin_kind(blocks, threads, ticks, lds_bytes, (uint32_t *)ctx->err.p + 8, mode, ctx->stream));
    return SPIHT_OK;
}

extern "C" int spiht_debug_words(spiht_ctx *ctx, uint32_t *out64) {
    if (!ctx || !out64) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out64, ctx->err.p, 256, hipMemcpyDeviceToHost));
    return SPIHT_OK;
}
extern "C" int spiht_debug_words_ext(spiht_ctx *ctx, uint32_t *out2048) {
    if (!ctx || !out2048) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out2048, ctx->err.p, 8192, hipMemcpyDeviceToHost));
    return SPIHT_OK;
}
#endif  // SPIHT_DIAG

// ------------------------------------------------------------------------------------------------
// multi-GPU: the one exchange of the path -- an all-gather of the stream slots, bit counts and start planes between
// encode and decode -- on RCCL, queued on the context's stream like every kernel.  librccl is loaded
// on first use (a single-GPU process never touches it); no link-time dependency.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetLastError)(ncclComm_t) = nullptr;  // optional (newer RCCL)
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static std::string g_rccl_name;   // the soname that was loaded, or what was tried and why each failed

static int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return SPIHT_OK;
    const char *names[] = {getenv("SPIHT_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *h = nullptr;
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) { g_rccl_name = n; break; }
        const char *e = dlerror();
        tried += std::string(tried.empty() ? "" : "; ") + n + ": " + (e ? e : "?");
    }
    if (!h) {
        g_rccl_name = "not loaded (" + tried + ")";
        g_hip_err = "cannot load librccl, tried " + tried;
        return SPIHT_ERR_HIP;
    }
    RcclApi a;
    a.h = h;
#define RSYM(field, name)                                                                \
    do {                                                                                  \
        *(void **)(&a.field) = dlsym(h, name);                                            \
        if (!a.field) { g_hip_err = std::string("librccl lacks ") + name; dlclose(h); return SPIHT_ERR_HIP; } \
    } while (0)
    RSYM(GetUniqueId, "ncclGetUniqueId");
    RSYM(CommInitRank, "ncclCommInitRank");
    RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(AllGather, "ncclAllGather");
    RSYM(AllReduce, "ncclAllReduce");
    RSYM(GroupStart, "ncclGroupStart");
    RSYM(GroupEnd, "ncclGroupEnd");
    RSYM(GetErrorString, "ncclGetErrorString");
    RSYM(GetVersion, "ncclGetVersion");
#undef RSYM
    *(void **)(&a.GetLastError) = dlsym(h, "ncclGetLastError");
    g_rccl = a;
    return SPIHT_OK;
}

#define NCCLCHK(expr)                                                                      \
    do {                                                                                   \
        ncclResult_t _r = (expr);                                                          \
        if (_r != ncclSuccess) {                                                           \
            g_hip_err = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);             \
            return SPIHT_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

struct spiht_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    void *scratch = nullptr;  // 16 device bytes for the barrier / the scalar reductions
};

static_assert(SPIHT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

// Which RCCL library serves this process: the soname dlopen took, or -- when none could be loaded -- every candidate
// tried with the loader's reason.  Empty before the first spiht_comm_* call.
extern "C" const char *spiht_rccl_library(void) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return g_rccl_name.c_str();
}

extern "C" int spiht_comm_unique_id(uint8_t *id) {
    if (!id) return SPIHT_ERR_ARG;
    CHK(rccl_load());
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SPIHT_COMM_ID_BYTES);
    return SPIHT_OK;
}

extern "C" int spiht_comm_create(spiht_ctx *ctx, const uint8_t *id, int world, int rank, spiht_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return SPIHT_ERR_ARG;
    *out = nullptr;
    CHK(rccl_load());
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, SPIHT_COMM_ID_BYTES);
    spiht_comm *c = new spiht_comm();
    c->world = world;
    c->rank = rank;
    c->device = ctx->device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        // the result's name and, where this RCCL has it, the text of the last error it logged (the actual reason)
        g_hip_err = "ncclCommInitRank(world " + std::to_string(world) + ", rank " + std::to_string(rank) + ", device " +
                    std::to_string(ctx->device) + ", " + g_rccl_name + "): " + g_rccl.GetErrorString(r);
        if (g_rccl.GetLastError) {
            const char *le = g_rccl.GetLastError(nullptr);
            if (le && *le) g_hip_err += std::string(" -- ") + le;
        }
        delete c;
        return SPIHT_ERR_HIP;
    }
    if (hipMalloc(&c->scratch, 16) != hipSuccess) {
        (void)hipGetLastError();
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return SPIHT_ERR_NOMEM;
    }
    *out = c;
    return SPIHT_OK;
}

extern "C" void spiht_comm_destroy(spiht_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int spiht_comm_info(spiht_comm *c, int *world, int *rank, int *rccl_version) {
    if (!c) return SPIHT_ERR_ARG;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) {
        *rccl_version = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(rccl_version);
    }
    return SPIHT_OK;
}

// Rank r's B slots / bit counts / start planes land in rows [r*B, (r+1)*B) of the gathered arrays on every rank
// (spiht_amd/dist.py: rank-major).  Three all-gathers in one RCCL group, queued on the context's stream: ordered
// after the encoder kernels queued before and before the decoder kernels queued after; the host does not block.
extern "C" int spiht_gather_streams(spiht_ctx *ctx, spiht_comm *c, const uint8_t *d_slots, const uint64_t *d_nbits,
                                    const uint8_t *d_max_n, int64_t B, uint64_t slot_stride, uint8_t *d_all_slots,
                                    uint64_t *d_all_nbits, uint8_t *d_all_max_n) {
    if (!ctx || !c || !d_slots || !d_nbits || !d_max_n || !d_all_slots || !d_all_nbits || !d_all_max_n || B < 0)
        return SPIHT_ERR_ARG;
    if (c->device != ctx->device) return SPIHT_ERR_ARG;
    if (B == 0) return SPIHT_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    StageTimer t(ctx, ST_GATHER);  // (the exchange's own time on the stream: a first multi-GPU run explains itself)
    NCCLCHK(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.AllGather(d_slots, d_all_slots, (size_t)B * slot_stride, ncclUint8, c->comm, ctx->stream);
    ncclResult_t r2 = g_rccl.AllGather(d_nbits, d_all_nbits, (size_t)B, ncclUint64, c->comm, ctx->stream);
    ncclResult_t r3 = g_rccl.AllGather(d_max_n, d_all_max_n, (size_t)B, ncclUint8, c->comm, ctx->stream);
    NCCLCHK(g_rccl.GroupEnd());
    NCCLCHK(r1);
    NCCLCHK(r2);
    NCCLCHK(r3);
    return SPIHT_OK;
}

// Where rank r's rows start in the gathered arrays (rank-major: rows [r*B, (r+1)*B)), in bytes from each array's start --
// what a rank's decoder reads after spiht_gather_streams.  Pure arithmetic (no device); spiht_pipeline_submit_gather uses it.
extern "C" int spiht_gather_row_offsets(int rank, int world, int64_t B, uint64_t slot_stride, uint64_t *off_slots, uint64_t *off_nbits,
                                        uint64_t *off_max_n) {
    if (world < 1 || rank < 0 || rank >= world || B < 0 || !off_slots || !off_nbits || !off_max_n) return SPIHT_ERR_ARG;
    if (slot_stride && (uint64_t)world * (uint64_t)B > UINT64_MAX / slot_stride) return SPIHT_ERR_TOO_LARGE;
    *off_slots = (uint64_t)rank * (uint64_t)B * slot_stride;
    *off_nbits = (uint64_t)rank * (uint64_t)B * sizeof(uint64_t);
    *off_max_n = (uint64_t)rank * (uint64_t)B;
    return SPIHT_OK;
}

// max over ranks of a host double (timing: the job's time is its slowest rank's); blocks until done
extern "C" int spiht_comm_allreduce_max_f64(spiht_ctx *ctx, spiht_comm *c, double *value) {
    if (!ctx || !c || !value || c->device != ctx->device) return SPIHT_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(c->scratch, value, 8, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(g_rccl.AllReduce(c->scratch, c->scratch, 1, ncclFloat64, ncclMax, c->comm, ctx->stream));
    HIPCHK(hipMemcpyAsync(value, c->scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}

// every rank has reached this call and everything queued on its context's stream before it has finished
extern "C" int spiht_comm_barrier(spiht_ctx *ctx, spiht_comm *c) {
    double one = 1.0;
    return spiht_comm_allreduce_max_f64(ctx, c, &one);
}

// ------------------------------------------------------------------------------------------------
// page-locked host memory for the arrays the drop-in calls RETURN (spiht_wrapper.py:192-216 returns a new ndarray): a
// device -> host copy into such memory is one DMA at the link's speed; into fresh pageable memory it is staged copies
// plus a page fault per 4 KB (measured on a 1080p RGB picture, 49.8 MB: 4.2 ms against 1 ms).  Pooled, because pinning
// is what costs: a freed buffer waits for the next request of about its size.
// ------------------------------------------------------------------------------------------------
namespace {
struct HostPool {
    std::mutex mu;
    struct Buf { void *p; size_t cap; };
    std::vector<Buf> free_list;            // oldest first
    std::vector<Buf> live;                 // handed out
    size_t pooled = 0, out = 0;
    static constexpr size_t kMaxPooled = (size_t)1 << 30, kMaxOut = (size_t)4 << 30;
};
HostPool g_host_pool;
bool g_host_pool_exiting = false;  // process exit: buffers are left to the system (the HIP runtime may be gone already)
}  // namespace

// bytes of page-locked host memory (usable from every device).  SPIHT_ERR_NOMEM when the allocation fails or more than
// 4 GiB are handed out already: the caller then takes ordinary memory (the calls accept any host pointer).
extern "C" int spiht_host_alloc(uint64_t bytes, void **h_ptr) {
    if (!h_ptr || bytes == 0) return SPIHT_ERR_ARG;
    *h_ptr = nullptr;
    HostPool &hp = g_host_pool;
    static const int registered = atexit([] { g_host_pool_exiting = true; });
    (void)registered;
    std::lock_guard<std::mutex> lk(hp.mu);
    if (hp.out + bytes > HostPool::kMaxOut) return SPIHT_ERR_NOMEM;
    int best = -1;
    for (size_t i = 0; i < hp.free_list.size(); i++)
        if (hp.free_list[i].cap >= bytes && hp.free_list[i].cap <= bytes + bytes / 4 + 4096 &&
            (best < 0 || hp.free_list[i].cap < hp.free_list[(size_t)best].cap)) best = (int)i;
    HostPool::Buf b;
    if (best >= 0) {
        b = hp.free_list[(size_t)best];
        hp.free_list.erase(hp.free_list.begin() + best);
        hp.pooled -= b.cap;
    } else {
        b.cap = (size_t)bytes;
        if (hipHostMalloc(&b.p, b.cap, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            // (room may be what is missing: the pool gives its buffers back and the request is tried once more)
            for (auto &f : hp.free_list) (void)hipHostFree(f.p);
            hp.free_list.clear();
            hp.pooled = 0;
            if (hipHostMalloc(&b.p, b.cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return SPIHT_ERR_NOMEM; }
        }
    }
    hp.live.push_back(b);
    hp.out += b.cap;
    *h_ptr = b.p;
    return SPIHT_OK;
}
extern "C" int spiht_host_free(void *h_ptr) {
    if (!h_ptr) return SPIHT_OK;
    HostPool &hp = g_host_pool;
    std::lock_guard<std::mutex> lk(hp.mu);
    for (size_t i = 0; i < hp.live.size(); i++) {
        if (hp.live[i].p != h_ptr) continue;
        const HostPool::Buf b = hp.live[i];
        hp.live.erase(hp.live.begin() + (long)i);
        hp.out -= b.cap;
        hp.free_list.push_back(b);
        hp.pooled += b.cap;
        while (!g_host_pool_exiting && (hp.pooled > HostPool::kMaxPooled || hp.free_list.size() > 8)) {  // the oldest goes back to the system
            (void)hipHostFree(hp.free_list.front().p);
            hp.pooled -= hp.free_list.front().cap;
            hp.free_list.erase(hp.free_list.begin());
        }
        return SPIHT_OK;
    }
    return SPIHT_ERR_ARG;  // not one of ours
}

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int spiht_dev_alloc(spiht_ctx *ctx, uint64_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 4);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return SPIHT_ERR_NOMEM;
    }
    return SPIHT_OK;
}
extern "C" int spiht_dev_free(spiht_ctx *ctx, void *d_ptr) {
    if (!ctx) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        forget_pads(ctx, d_ptr);
    }
    HIPCHK(hipFree(d_ptr));
    return SPIHT_OK;
}
extern "C" int spiht_dev_upload(spiht_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!ctx || (!d_dst && bytes) || (!h_src && bytes)) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}
extern "C" int spiht_dev_download(spiht_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!ctx || (!h_dst && bytes) || (!d_src && bytes)) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPIHT_OK;
}
extern "C" int spiht_dev_copy(spiht_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes) {
    if (!ctx || (!d_dst && bytes) || (!d_src && bytes)) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SPIHT_OK;
}
extern "C" int spiht_dev_memset(spiht_ctx *ctx, void *d_dst, int value, uint64_t bytes) {
    if (!ctx || (!d_dst && bytes)) return SPIHT_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return SPIHT_OK;
}
