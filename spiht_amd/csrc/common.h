// Shared host/device definitions for libspiht_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPIHT_MAX_LEVELS 32
#define SPIHT_MAX_TAPS 20

// Exact unsigned division by an invariant divisor d >= 2 (Granlund-Montgomery round-up form):
//   q = (t + ((n - t) >> 1)) >> sh,  t = mulhi(m, n)
struct FastDiv {
    uint32_t m, sh, d, pad;
};

static inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.pad = 0;
    if (d < 2) { f.m = 0; f.sh = 0; return f; }  // callers never divide by < 2
    uint32_t l = 0;
    while ((1ull << l) < d) l++;  // ceil(log2 d)
    uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.m = (uint32_t)m;
    f.sh = l - 1;
    return f;
}

#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> f.sh;
}
#endif

// Geometry of one coefficient array [c,h,w] with an ll_h x ll_w root block.
struct Geom {
    int32_t c, h, w, ll_h, ll_w;
    uint32_t hw;  // h*w
    uint32_t n;   // c*h*w  (< 2^30)
    uint32_t pad;
    FastDiv div_w, div_hw;
};

// List entry: bit 31 = type A (1) / type B (0) for LIS entries; bit 30 = "leaf": a type-A entry whose node
// has no offspring (the reference still queues it when its parent's B entry fires; it emits a 0 in every
// plane and can never fire, encoder_decoder.rs:229-239); low 30 bits = linear index k*hw + i*w + j.
#define ENT_A 0x80000000u
#define ENT_LEAF 0x40000000u
#define ENT_IDX 0x3FFFFFFFu

// Per-slot scratch of the list coder.  A "slot" serves one image at a time.
struct ListCaps {
    uint32_t lip, lsp, lis;  // capacities in entries
    uint32_t pad;
};

// One image on several CUs (encode_wide.hip): per group of workgroups a control block and the descriptors of the chunk scan.
struct WideCtl {          // zero-filled before every launch
    uint32_t bar_count, bar_gen;  // the group's barrier: arrivals (monotonic), released epoch
    uint32_t go;          // workgroup 0 has handed over: st[] is valid
    uint32_t bad;         // 1: a list capacity exceeded; 2: a wait ran out of time -- the group's workgroups were not all
                          // resident (other kernels held the CUs): everyone leaves, k_encode codes the image again
    uint32_t st[12];      // n, lip_len, lsp_len, lis_len, bitpos lo / hi, LIP buffer, LIS buffer, done
    uint32_t tot[8];      // totals of the pass just finished, two sets (pass number & 1)
    uint32_t bad_at;      // ~(pass number) of the first pass in which a capacity was exceeded (atomicMax), 0: none
    uint32_t pad[3];
};
struct WideArgs {
    WideCtl *ctl;         // [groups]
    uint64_t *desc;       // [groups][2][maxchunks][4] words [pass number | count]: aggregates, inclusive prefixes; zero-filled
    uint32_t maxchunks;
    uint32_t G;           // workgroups per image
    uint32_t solo;        // entries on the three lists together up to which a plane is coded by workgroup 0 alone
    uint32_t pad;
};

struct EncArgs {
    Geom g;
    ListCaps caps;
    int32_t B;
    int32_t nslots;
    const int32_t *x;        // [B, n]
    const uint8_t *dmsb;     // [B, n]
    const uint8_t *lmsb;     // [B, n]
    const uint32_t *maxabs;  // [B]
    uint64_t max_bits;       // already mapped: 0 -> unlimited
    uint8_t *out;            // [B, slot_stride]
    uint64_t slot_stride;    // bytes, multiple of 4
    uint64_t *out_nbits;     // [B]
    uint8_t *out_maxn;       // [B]
    // scratch, per slot
    uint32_t *lip0, *lip1, *lsp, *lis0, *lis1, *lis2;
    uint32_t *err;           // device error word
    const WideCtl *redo;     // k_encode behind k_encode_wide: codes only the images whose group gave up (bad & 2); else null
    float log2_thresh[32];   // log2_thresh[k]: smallest float m < 2^k with (u8)log2f(m) == k (host libm), or 2^k
};

// Tile of the inverse level-1 kernels (dwt.hip), in output positions; in band positions half of it.
#define IW_TH 24    // output rows per tile (two halves, one per half of the workgroup; a multiple of 4)
#define IW_TW 128   // output cols per tile, one thread per column per half

// Which tiles of the inverse transform's level 1 have anything but zeros in their detail bands: the list decoder knows
// every cell it writes, so it sets one word per (plane, tile) whose staged band region -- the tile's IW_TH/2 x IW_TW/2
// band positions plus the halo of F/2 - 1 rows / columns behind them -- contains a decoded cell of a level-1 band; the
// inverse level-1 kernel then does not read the three int32 detail bands of a tile whose word is still zero (at 0.5 bpp
// nearly all of them).  Conservative by construction: a set word only means "read".  p == nullptr: no flags.
struct L1Flags {
    uint32_t *p;              // [planes, gy, gx] (planes = B*c of the launch), zero-filled before the decoder runs
    int32_t off_h, off_w;     // offsets of the level-1 detail bands in the packed array
    int32_t band_h, band_w;   // their size
    int32_t hf1;              // F/2 - 1: band rows / columns of halo a tile stages beyond its own
    int32_t gx, gy;           // tiles per plane
    int32_t pad;
};

struct DecArgs {
    Geom g;
    ListCaps caps;
    int32_t B;
    int32_t nslots;
    const uint8_t *data;      // [B, slot_stride]
    uint64_t slot_stride;     // bytes, multiple of 4
    const uint64_t *nbytes;   // [B]
    const uint8_t *max_n;     // [B]
    int32_t *out;             // [B, n], zero-filled before launch
    uint32_t *lip0, *lip1, *lsp_idx, *lis0, *lis1, *lis2;
    int32_t *lsp_val;
    uint32_t *err;
    uint32_t *lsp_count;      // [nslots] or null: final LSP length of the image a slot decoded (k_unscatter)
    L1Flags fl;               // occupancy of the inverse transform's level-1 tiles (fl.p null: not wanted)
    // decode_with_metadata only (k_decode<true>): one trace record per stream position 0..nbits (the last one is
    // the operation that was waiting for a bit when the stream ended)
    uint32_t *tr_ent;         // [B, tr_stride]  entry: node index | filter << 28 (| ENT_A / ENT_LEAF, ignored)
    uint8_t *tr_act;          // [B, tr_stride]  action 0..6 | plane << 3;  TR_NONE where no operation starts
    uint64_t tr_stride;       // records per image (>= 8*nbytes + 1)
};

// decode_with_metadata: list entries also carry the filter (encoder_decoder.rs:457-462) in bits 28-29, because
// on trees with duplicated nodes (odd ll_h / ll_w) the same node is reached under two different filters; the
// index is then limited to 28 bits.
#define ENT_IDX_META 0x0FFFFFFFu
#define ENT_FILT_SHIFT 28
#define TR_NONE 0xFFu

// k_meta_rows / k_meta_fold (metadata.hip)
struct MetaArgs {
    Geom g;
    int32_t level;            // number of detail levels (Slices.other_slices.len())
    int32_t pad;
    uint64_t rows;            // 8*nbytes + 1
    const uint32_t *tr_ent;   // [rows]
    const uint8_t *tr_act;    // [rows]
    const uint8_t *data;      // stream bytes
    const int32_t *slices;    // device: top {start_i,end_i,start_j,end_j}, then [level][3][4]
    int32_t *meta;            // [rows, 8]
    const uint32_t *skey;     // sorted node index per record (fold)
    const uint32_t *spos;     // record position, sorted by (node index, position)
};

struct PyrArgs {
    Geom g;
    int32_t B;
    int32_t round;            // 1..: index-doubling depth handled by this launch
    const int32_t *x;
    uint8_t *dmsb, *lmsb;
    uint32_t *maxabs;
};

// Colour model change fused into level 1 of the transform (dwt.hip): per pixel w = M * spow(A * u, p), spow(x, p) =
// sign(x)|x|^p -- the shape of RGB <-> IPT (spiht/color_models.py:6-13 -> colour-science; here the published matrices)
struct Color3 {
    double A[9], M[9], p;
};

// One forward DWT level (dwt.hip)
struct DwtKArgs {
    int32_t c;             // channels (plane index = b*c + k)
    int32_t F, mode;
    int32_t in_h, in_w, out_h, out_w;
    int32_t off_h, off_w, enc_h, enc_w;
    int32_t last;          // coarsest level: LL is quantised into the packed array too
    int32_t planes;        // B*c (set by the launcher)
    int32_t f32;           // single-precision level (k_dwt_level_f32): `in` / `ll_out` then point to float arrays
    int32_t ov_h, ov_w;    // first output row / column that hangs over the bottom / right end of the input far enough for
                           // PyWavelets' overhang order to differ from ascending order; out_h / out_w: none (launcher)
    const double *in;      // [planes, in_h, in_w]
    double *ll_out;        // [planes, out_h, out_w]
    int32_t *coeffs;       // [planes, enc_h, enc_w]
    const double *mults;   // device [c] or null
    uint32_t *maxabs;      // device [B] or null: atomicMax of |quantised coefficient| per image
    double q;
    double lo[SPIHT_MAX_TAPS], hi[SPIHT_MAX_TAPS];  // dec_lo, dec_hi
    float lo_f[SPIHT_MAX_TAPS], hi_f[SPIHT_MAX_TAPS];  // ... as PyWavelets' single-precision transform has them (f32 levels)
    int32_t color, pad2;   // level 1 of a 3-channel image with the colour model change on its loads (k_dwt1_color)
    Color3 col;
};

// One inverse DWT level (dwt.hip)
// Tile hand-out of the persistent inverse-transform kernel (k_idwt_level_pf): one counter per XCD in device memory,
// never reset -- the launcher knows how many numbers a launch draws from each (tiles + 2 per workgroup) and hands the
// kernel the counter values it starts from (all arithmetic modulo 2^32).
struct TileCtr {
    uint32_t *dev;      // 8 counters, 32 words apart (a memory line each), and a ninth (word 8 * 32): persistent workgroups
                        // that have started, ever; zero when allocated
    uint32_t base[8];   // host-side: value of each counter before the next launch
    int32_t wg_per_cu;  // persistent workgroups per CU of the next launches (0: the kernel's default, IWP_WG)
    int32_t num_cu;     // of the context's device
    int32_t lds_per_cu; // bytes of LDS a CU has (hipDeviceProp_t::maxSharedMemoryPerMultiProcessor)
    uint32_t started;   // host-side: what the ninth counter reads once every launch queued so far has all its workgroups on the CUs
};
#define TILECTR_WORDS (9 * 32)
#define TILECTR_STARTED (8 * 32)
struct TileBase { uint32_t v[8]; };

struct IdwtKArgs {
    int32_t c;
    int32_t F;
    int32_t band_h, band_w;    // detail band size (= approximation size used)
    int32_t out_h, out_w;      // 2*band - F + 2
    int32_t a_h, a_w;          // stored size of the incoming approximation (>= band; trim rule)
    int32_t off_h, off_w, enc_h, enc_w;
    int32_t first;             // coarsest level: approximation comes from rec[0:band_h, 0:band_w]
    int32_t planes;            // B*c (set by the launcher)
    const double *a_in;        // [planes, a_h, a_w]
    const int32_t *rec;        // [planes, enc_h, enc_w]
    const uint32_t *flags;     // [planes, gy, gx] or null: L1Flags words of this level's tiles (level 1 only)
    double *out;               // [planes, out_h, out_w]
    const double *mults;
    double q;
    double lo[SPIHT_MAX_TAPS], hi[SPIHT_MAX_TAPS];  // rec_lo, rec_hi
    int32_t color, pad2;       // level 1 of a 3-channel image with the colour model change on its stores (k_idwt1_color)
    Color3 col;
};
