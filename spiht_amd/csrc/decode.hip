// SPIHT list decoder (gfx950): one workgroup of DEC_NW wavefronts owns one image at a time.
//
// Reproduces /root/reference/src/encoder_decoder.rs:307-454 on all 8*nbytes bits of the stream
// (src/lib.rs:38 hands the pad bits to the decoder as data).  Unlike the encoder, the position of a list
// entry's bits depends on every bit decoded before it.  What is truly serial is only WHERE each entry's bits
// start; everything else (values, list appends) is data parallel once that is known.
//   * LIP pass: tokens are '0' | '1 s'.  No serial walker: every lane takes one 64-bit stream window, the token-start
//     mask of a window is computed in O(1) for both possible carry-in states with the carry trick used for escaped
//     characters in SIMD JSON parsers (runs of ones pair up from their first bit), the carry functions and the token /
//     LSP / LIP counts are combined with block scans, and the windows' entries are then emitted by all wavefronts.
//   * LIS pass (generation by generation = the reference's FIFO order): unfired entries and fired B entries take one
//     bit; only a fired type-A entry (1 + 4..8 bits) shifts what follows.  Wavefront 0, the sequencer, hops from fired
//     A to fired A with a hand-written scalar loop (find-first-set on `bits & type-mask`, token length by v_readlane)
//     and publishes, per 64-bit window, the mask of fired entries through an LDS ring.  Wavefront 1, the helper, runs
//     ahead of it in batches of 32 windows, one lane per window: each window's bits, its per-position token lengths
//     (three bit planes, worked out with bitwise operations on the window's word) and the whole walk of the window
//     under the hypothesis "all entries are type A" for each of its nine possible entry points -- where the entries
//     a window meets ARE all type A (three windows of four at 1080p: profiles/r04_lis_type_runs.txt) the sequencer
//     looks the window up instead of hopping through it, runs of such windows in a loop written in assembly.
//     The other wavefronts are workers:
//     they rebuild entry starts from the mask, lane = stream position, and produce next generation / retained list /
//     LIP / LSP appends; the running list lengths pass from worker to worker through a small LDS chain.
//   * refinement: bit t belongs to LSP entry t -- all wavefronts, no sequencing.
// Decoded values live next to the LSP (lsp_val, one private value per list entry) and are scattered into the
// coefficient array at the end.  On trees with duplicated nodes (odd ll_h / ll_w) two or three list
// entries own the same cell; the reference's writes to it are sequential (encoder_decoder.rs:364-372, 396-404, 443),
// so for such a cell the entries are brought together at the end (resolve_dups) and their operations -- all of them
// recoverable from the private values -- are replayed in stream order: exact for any byte string, not only for
// encoder-produced streams.
#include "common.h"

#ifndef DEC_NW
#define DEC_NW 12           // wavefronts per workgroup
#endif
// LIS pass roles: wavefront 0 = sequencer, wavefront 1 = helper (prepares the sequencer's per-window inputs ahead of
// it), wavefronts 2.. = workers.  Measured per 256 images (1080p, 0.5 bpp): 5 workers 8.6 ms, 6 workers 7.5 ms,
// 8: 7.16 ms, 10: 7.08 ms, 14: 7.14 ms -- the all-wave passes (LIP window scan, refinement, final scatter) gain from
// the extra waves too; keeping the sequencer's SIMD free of workers (waves 4, 8 parked) made no difference.
#define DEC_NWK (DEC_NW - 2)
#define DEC_WK(wave) ((wave) - 2)
#define DEC_IS_WORKER(wave) ((wave) >= 2)
#define DEC_PREP 64   // windows the helper may run ahead of the sequencer (ring in LDS)
#define DEC_PREP_B 32 // ... which it prepares and announces in batches of this many (lane = (window, half))
#ifndef DEC_RING
#define DEC_RING 64   // windows in flight between sequencer and workers; a power of two (32 measured 4 % slower)
#endif
static_assert((DEC_RING & (DEC_RING - 1)) == 0 && DEC_RING <= DEC_NW * 64, "ring: power of two, reset by one pass of the block");
#ifndef DEC_WSLEEP
#define DEC_WSLEEP 1  // s_sleep argument (x64 cycles) of a worker waiting for its next window
#endif
#define SEQ_OPEN 0xFFFFFFFFu
#define SPIN_LIMIT (1u << 24)  // bound on every LDS spin (about a second): a protocol bug must not hang the GPU

#ifdef DEC_PROF  // diagnostic build only: where the sequencer's time goes (s_memtime ticks >> 10 into err[16..])
#define PF_ADD(k) do { uint64_t _n = __builtin_amdgcn_s_memtime(); pf[k] += _n - pt; pt = _n; } while (0)
#define PF_CNT(k, v) pf[k] += (v)
#else
#define PF_ADD(k)
#define PF_CNT(k, v)
#endif

// Two builds of this file are linked (Makefile): 12 wavefronts per workgroup -- the shortest walk of one stream -- and
// 8 (-DDEC_NW=8 -DDEC_VARIANT=w8): slower alone (8.3 instead of 8.0 ms per 256 images), but a lighter neighbour for the
// HBM-bound kernels the pipelined schedule runs beside it (spiht_ctx_set_decoder_waves).  Everything below lives in a
// namespace of the build so that the kernels of the two differ in name.
#ifdef DEC_VARIANT
#define DEC_CAT2(a, b) a##b
#define DEC_CAT(a, b) DEC_CAT2(a, b)
#define DEC_NS DEC_CAT(dec_, DEC_VARIANT)
#define DEC_LAUNCH DEC_CAT(spiht_launch_decode_, DEC_VARIANT)
#else
#define DEC_NS dec_main
#define DEC_LAUNCH spiht_launch_decode
#endif
namespace DEC_NS {

struct Item {  // one 64-bit stream window of one pass
    uint32_t kind;        // 0 = LIP window, 1 = LIS window
    uint32_t Wb;          // stream position of window bit 0
    uint32_t pos0, pos1;  // the pass owns window bits [pos0, pos1)
    uint32_t e_start;     // LIP: index in the LIP of the first token; LIS: index in the queue of the first entry
    uint32_t cin;         // LIP: bit pos0 is the pending sign bit of the previous window's last token
    uint32_t m_rem;       // LIP: tokens still to read at window start
    uint32_t first;       // LIS: first window of the generation -> bases below instead of the chain
    uint32_t b_lsp, b_lip, b_nxt, b_ret;  // list lengths before this window (LIP pass: exact, from the sequencer)
    uint64_t lo, hi;      // window bits and the next window's bits (zero at and past nbits)
    uint64_t fm;          // LIS: start positions of fired type-A entries with offspring
};

// LDS form of a LIS window: only what the walk itself produces (one 16-byte write by one lane) + the `ready` word,
// which is written LAST and holds sequence number + 1.  The windows of a phase are consecutive, so everything else
// follows from the sequence number and the phase's PhaseInfo; the worker fetches the window's bits itself.
struct __attribute__((aligned(16))) Slot {
    uint32_t fm0, fm1;    // start positions of fired type-A entries with offspring
    uint32_t e_start;     // index in the queue of the window's first entry
    uint32_t w3;          // pos0 | pos1 << 8
    uint32_t ready;
    uint32_t pad[3];
};

struct PhaseInfo {  // written by the sequencer before the first window of a LIS phase (generation)
    uint32_t seq0;        // sequence number of the phase's first window
    uint32_t widx0;       // its 64-bit window index in the stream
    uint32_t b_lsp, b_lip, b_ret;  // list lengths at the start of the phase
};

struct Chain {  // running list lengths after item k; seq == k+1 once item k published them
    uint32_t seq;
    uint32_t lsp, lip, nxt, ret;
};

struct LipItem {  // one LIP-pass window, produced by the parallel window scan
    uint64_t lo;
    uint32_t Wb, e_start, m_rem, b_lsp, b_lip;
    uint32_t flags;  // pos0 | cin << 8 | hibit0 << 9 | valid << 10
};

// one window walked under the all-type-A hypothesis, per entry point 0..8 (see DecShared::tab)
struct TabRow {
    uint64_t fm[9];    // start positions of the fired entries ... as a mask over the entries that start in the window
    uint16_t ce[9];    // entries that start in the window | (bits the last one hangs over into the next window) << 7
    uint16_t pad[3];
};
static_assert(sizeof(TabRow) == 96, "the run loop of the sequencer steps through the rows in assembly");

struct DecShared {
    // The LIP pass and the LIS pass never run at the same time (barriers between them), so their scratch shares
    // memory (the LIP scratch, 32 B per window of a round, is the larger of the two: 24.6 KB at 12 wavefronts).  The
    // footprint matters only when this kernel runs next to the DWT kernels on another stream (bench.py --pipeline 1):
    // 4 of their workgroups (34 KB each) + 17 KB fit a CU's 160 KB, 4 + 25 KB do not (tools/corun_spin.py: the DWT is
    // 23 % slower next to idle 25 KB workgroups, not at all next to 0 KB ones); build with -DDEC_NW=8 for that mode.
    union {
        LipItem lipq[DEC_NW * 64];
        struct {
            Slot ring[DEC_RING];
            Chain chain[DEC_RING];
            PhaseInfo ph;
            // helper -> sequencer: per window, the length a fired type-A entry would have at each of its 64 positions, as
            // three bit planes of (length - 5) and, fourth word, the window's bits themselves; the sequencer turns a row
            // into one length per lane when it has to hop through the window (most windows it looks up, see tab)
            uint64_t pln[DEC_PREP][4];
            // helper -> workers: the bits of each window and of the one after it (zero at and past the end of the
            // stream), so that a worker's window costs it no stream load.  A worker can lag DEC_RING windows behind the
            // sequencer, the helper run DEC_PREP ahead of it.
            uint64_t wbits[DEC_PREP + DEC_RING][2];
            // helper -> sequencer: the walk of each window under the hypothesis "every list entry this window meets is a
            // type-A entry with offspring" -- then which bits are entries depends on the bits alone: an entry is '0' or
            // '1' + 4..8 offspring bits, whatever its index in the queue.  Per entry point o = 0..8 (the window's first
            // o bits belong to the previous window's last entry): TabRow::fm = start positions of the fired entries, ce =
            // entries that start in the window | (bits the last one hangs over into the next window) << 7.
            TabRow tab[DEC_PREP];
        };
    };
    uint64_t wpart[DEC_NW];     // per-wave partials of the block scans
    uint32_t wfun[DEC_NW];      // per-wave carry functions
    uint32_t lp_end[4];         // LIP pass end: [0] kind (0 none, 1 ends, 2 trunc), [1] P after the pass
    uint32_t pprog;             // windows of this phase the helper has prepared
    uint32_t tprog;             // ... and walked under the all-type-A hypothesis (DecShared::tab): announced a little later
    uint32_t sprog;             // windows of this phase the sequencer is done with (announced every DEC_RING/2 windows)
    uint32_t head;              // items produced so far
    uint32_t phase_end[2];      // sequence number at which the phase of that parity ends, SEQ_OPEN while open
    uint32_t wdone[DEC_NWK];    // items completed per worker
    uint32_t bad;
    uint32_t ndup;              // LSP entries of duplicated cells met by the final scatter (resolve_dups walks only those)
    // results of a phase, written by the sequencer before the closing barrier
    uint32_t r_P, r_done, r_lsp, r_lip;
};

__device__ __forceinline__ uint32_t lds_load(uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// popcount of the bits of `m` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// encoder_decoder.rs:14-29
__device__ __forceinline__ int32_t set_bit_i32(int32_t x, uint32_t n, uint32_t bit) {
    uint32_t m = 1u << n;
    if (x >= 0) return bit ? (int32_t)((uint32_t)x | m) : (int32_t)((uint32_t)x & ~m);
    uint32_t a = (uint32_t)(-x);
    a = bit ? (a | m) : (a & ~m);
    return -(int32_t)a;
}

__device__ __forceinline__ void decomp(const Geom &g, uint32_t idx, uint32_t &k, uint32_t &i, uint32_t &j) {
    k = fdiv(idx, g.div_hw);
    uint32_t r = idx - k * g.hw;
    i = fdiv(r, g.div_w);
    j = r - i * (uint32_t)g.w;
}

__device__ __forceinline__ uint32_t child_base(const Geom &g, uint32_t k, uint32_t i, uint32_t j, uint32_t &r, uint32_t &cc) {
    if (i < (uint32_t)g.ll_h && j < (uint32_t)g.ll_w) {
        r = (i & 1u) * (uint32_t)g.ll_h + (i & ~1u);
        cc = (j & 1u) * (uint32_t)g.ll_w + (j & ~1u);
    } else {
        r = 2 * i;
        cc = 2 * j;
    }
    return k * g.hw + r * (uint32_t)g.w + cc;
}

__device__ __forceinline__ uint32_t make_a_entry(uint32_t idx, uint32_t ci, uint32_t cj, uint32_t H, uint32_t W) {
    return idx | ENT_A | ((2 * ci + 1 < H && 2 * cj + 1 < W) ? 0u : ENT_LEAF);
}

// Token-start mask of a LIP-pass window (tokens '0' | '1 s').  The pass owns bits >= pos of the window.
// cin: bit `pos` is the pending sign bit of the previous window's last token.  cout: the token starting at
// bit 63 is '1' and its sign bit is the next window's bit 0.  Runs of ones pair up from their first bit, so
// the sign positions are the odd offsets inside a run plus the zero that follows an odd-length run; the
// run-parity is found with the add-carry trick.
__device__ __forceinline__ uint64_t lip_starts(uint64_t W, uint32_t pos, uint32_t cin, uint32_t &cout) {
    const uint64_t E = 0x5555555555555555ull, O = 0xAAAAAAAAAAAAAAAAull;
    const uint64_t own = ~0ull << pos;
    uint64_t Wc = W & own;
    if (cin) Wc &= ~(1ull << pos);
    uint64_t RS = Wc & ~(Wc << 1);               // first bit of every run of ones
    uint64_t ce = Wc + (RS & E);                 // carry ripples through runs starting on even bits
    uint64_t co = Wc + (RS & O);
    uint64_t Me = (ce ^ Wc) & Wc;                // bits of runs that start on an even position
    uint64_t Mo = Wc & ~Me;
    uint64_t G = (Me & O) | (Mo & E);            // odd offsets inside a run: sign bits
    G |= ((ce & ~Wc) & O) | ((co & ~Wc) & E);    // the zero right after an odd-length run: sign bit
    if (cin) G |= 1ull << pos;
    uint64_t S = ~G & own;
    cout = (uint32_t)((S >> 63) & (W >> 63) & 1ull);
    return S;
}

// position of the k-th (0-based) set bit of x; needs popcount(x) > k
__device__ __forceinline__ uint32_t select64(uint64_t x, uint32_t k) {
    uint32_t r = 0;
    uint32_t c = (uint32_t)__popc((uint32_t)x);
    if (k >= c) { k -= c; x >>= 32; r = 32; }
    uint32_t y = (uint32_t)x;
    c = (uint32_t)__popc(y & 0xFFFFu);
    if (k >= c) { k -= c; y >>= 16; r += 16; }
    c = (uint32_t)__popc(y & 0xFFu);
    if (k >= c) { k -= c; y >>= 8; r += 8; }
    c = (uint32_t)__popc(y & 0xFu);
    if (k >= c) { k -= c; y >>= 4; r += 4; }
    c = (uint32_t)__popc(y & 0x3u);
    if (k >= c) { k -= c; y >>= 2; r += 2; }
    if (k >= (y & 1u)) r += 1;
    return r;
}

// What a LIP window contains, as a function of the item fields only (no cross-lane operation): the parallel
// window scan (one window per lane) and the per-window work (one window per wavefront) both call it.
struct LipWin {
    uint64_t S_in;     // token starts that belong to the pass and are complete
    uint64_t sig;      // those of them that are significant ('1 s')
    uint32_t ntok;     // tokens of the pass that start in this window (including a truncated last one)
    uint32_t trunc;    // the last token has its sign bit past the end of the stream: decoding stops
    uint32_t trunc_pos;  // ... and starts at this window bit
    uint32_t ends;     // the pass ends inside this window
    uint32_t pos1;     // ... at this bit
    uint32_t cout;
};
__device__ __forceinline__ LipWin lip_window(uint64_t lo, uint32_t Wb, uint32_t pos, uint32_t cin, uint32_t m_rem,
                                             uint32_t nbits) {
    LipWin r;
    const uint32_t vb = (nbits - Wb) < 64u ? (nbits - Wb) : 64u;
    uint64_t S = lip_starts(lo, pos, cin, r.cout);
    if (vb < 64) S &= (1ull << vb) - 1ull;
    const uint32_t cnt = (uint32_t)__popcll(S);
    r.ends = cnt > m_rem;
    r.pos1 = 64;
    if (r.ends) {
        // position of token start number m_rem (the first one that is NOT part of the pass); no cross-lane
        // operation here: this function also runs with one window per lane
        r.pos1 = select64(S, m_rem);
        S &= (1ull << r.pos1) - 1ull;
    }
    r.ntok = r.ends ? m_rem : cnt;
    uint64_t sg = S & lo;
    // a '1' whose sign bit is at or past nbits (only the very last token of the stream can be)
    r.trunc = 0;
    r.trunc_pos = 0;
    if (sg) {
        const uint32_t top = 63u - (uint32_t)__builtin_clzll(sg);
        if (Wb + top + 1 >= nbits) { r.trunc = 1; r.trunc_pos = top; sg &= ~(1ull << top); S &= ~(1ull << top); }
    }
    r.S_in = S;
    r.sig = sg;
    return r;
}

struct BitSrc {
    const uint32_t *gw;   // stream words of this image
    uint32_t nwords;      // words readable in the slot
    uint32_t nbits;       // valid bits (8*nbytes)
};

// 32-bit stream word `wi`, zero at and past nbits
__device__ __forceinline__ uint32_t stream_word(const BitSrc &bs, uint32_t wi) {
    uint32_t v = 0;
    const uint64_t b0 = (uint64_t)wi * 32;
    if (wi < bs.nwords && b0 < bs.nbits) {
        v = bs.gw[wi];
        const uint32_t rem = bs.nbits - (uint32_t)b0;
        if (rem < 32) v &= (1u << rem) - 1u;
    }
    return v;
}

// 64-bit stream word `w64` (two 32-bit words), zero at and past nbits
__device__ __forceinline__ uint64_t stream_word64(const BitSrc &bs, uint32_t w64) {
    return (uint64_t)stream_word(bs, 2 * w64) | ((uint64_t)stream_word(bs, 2 * w64 + 1) << 32);
}

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int o) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_up((int)lo, o);
    hi = (uint32_t)__shfl_up((int)hi, o);
    return ((uint64_t)hi << 32) | lo;
}

// exclusive prefix sum of a packed 64-bit value over the whole workgroup (two barriers)
__device__ __forceinline__ uint64_t block_exscan(DecShared &sh, uint64_t v, uint64_t &total, uint32_t wave, uint32_t lane) {
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = shfl_up_u64(inc, o);
        if (lane >= (uint32_t)o) inc += t;
    }
    if (lane == 63) sh.wpart[wave] = inc;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < DEC_NW; w++) {
        const uint64_t pw = sh.wpart[w];
        if (w < wave) pre += pw;
        tot += pw;
    }
    __syncthreads();
    total = tot;
    return pre + inc - v;
}

// ---- sequencer side of the ring ----
// LDS instructions of one wavefront execute in issue order, so the item words followed by the slot's `ready`
// word need no wait in between; a worker that sees ready == k+1 sees item k.  Ring space is checked once per
// half ring: before item k (k a multiple of RING/2) every item below k - RING/2 must have been consumed.
__device__ __forceinline__ void seq_ring_check(DecShared &sh, uint32_t seq, uint32_t &fc, uint32_t lane, uint32_t kw, uint64_t *pfw) {
    constexpr uint32_t HALF = DEC_RING / 2;
    fc = HALF;
    // the helper's ring entries of the windows before this one are free (announced here, every HALF windows,
    // rather than with a test of its own per window)
    if (lane == 0) __hip_atomic_store(&sh.sprog, kw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (seq >= DEC_RING) {
        const uint32_t lim = seq - HALF;  // items [0, lim) must be done
        // worker w takes items w, w + NWK, ...: lane w checks that worker's counter, one LDS read for all of them
        const uint32_t w = lane < DEC_NWK ? lane : 0u;
        const uint32_t need = lim > w ? (lim - w + DEC_NWK - 1) / DEC_NWK : 0;
        uint32_t spins = 0;
#ifdef DEC_PROF
        const uint64_t tws = __builtin_amdgcn_s_memtime();
#endif
        while (__ballot(lds_load(&sh.wdone[w]) < need) != 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { sh.bad = 2; break; }  // never expected: keeps a bug from hanging the GPU
        }
#ifdef DEC_PROF
        if (pfw) *pfw += __builtin_amdgcn_s_memtime() - tws;
#endif
    }
}

__device__ __forceinline__ void seq_publish(DecShared &sh, uint32_t seq, uint32_t &fc, uint64_t fm, uint32_t e_start,
                                            uint32_t pos0, uint32_t pos1, uint32_t lane, uint32_t kw, uint64_t *pfw = nullptr) {
    // `seq` is wave-uniform (the caller keeps it in an SGPR); fc counts down to the next multiple of HALF
    if (fc == 0) seq_ring_check(sh, seq, fc, lane, kw, pfw);
    asm volatile("s_add_i32 %0, %0, -1" : "+s"(fc) : : "scc");  // (asm: keeps the counter in an SGPR)
    if (lane == 0) {
        Slot *slot = &sh.ring[seq % DEC_RING];
        *reinterpret_cast<uint4 *>(slot) = make_uint4((uint32_t)fm, (uint32_t)(fm >> 32), e_start, pos0 | (pos1 << 8));
        asm volatile("" ::: "memory");  // compiler order only: the hardware keeps one wavefront's LDS writes in order
        // (a relaxed workgroup-scope atomic stays a ds_write; a volatile store through the generic pointer became a
        // flat_store + s_waitcnt vmcnt(0) on the sequencer's critical path)
        __hip_atomic_store(&slot->ready, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// worker side: the item of sequence number k (its slot has been seen ready)
__device__ __forceinline__ Item slot_unpack(DecShared &sh, const BitSrc &bs, uint32_t k) {
    const uint4 a = *reinterpret_cast<const uint4 *>(&sh.ring[k % DEC_RING]);
    const uint32_t seq0 = sh.ph.seq0;
    const uint32_t widx = sh.ph.widx0 + (k - seq0);
    Item it;
    it.kind = 1; it.cin = 0; it.m_rem = 0; it.b_nxt = 0;
    it.fm = (uint64_t)a.x | ((uint64_t)a.y << 32);
    it.e_start = a.z;
    it.pos0 = a.w & 0xFFu; it.pos1 = (a.w >> 8) & 0xFFu;
    it.first = k == seq0;
    it.b_lsp = sh.ph.b_lsp; it.b_lip = sh.ph.b_lip; it.b_ret = sh.ph.b_ret;
    it.Wb = widx << 6;
    const uint64_t *wb = sh.wbits[(k - seq0) % (DEC_PREP + DEC_RING)];  // written by the helper before the sequencer used it
    it.lo = wb[0];
    it.hi = wb[1];
    return it;
}

// ---- decode_with_metadata trace (k_decode<true> only): which list entry every stream position belongs to ----
struct Trace {
    uint32_t *ent;
    uint8_t *act;
    uint32_t nbits;
};
// record for stream position `pos`; position nbits is the operation left waiting when the stream ended
// (encoder_decoder.rs:664-684 assigns the row BEFORE pop_bit returns, :649-651)
__device__ __forceinline__ void tr_put(const Trace &tr, uint32_t pos, uint32_t act, int n, uint32_t e) {
    if (pos <= tr.nbits) {
        tr.ent[pos] = e;
        tr.act[pos] = (uint8_t)(act | ((uint32_t)n << 3));
    }
}

// ---- occupancy of the inverse transform's level-1 tiles (common.h: L1Flags) ----
// The cell idx of the image's coefficient array gets a value: if it is a cell of a level-1 detail band, the words of the
// inverse-transform tiles that stage it are set (the tile it lies in and, within F/2 - 1 band rows / columns of that
// tile's start, the tile before: a tile stages its halo behind itself).  Plain stores of 1: racing writers agree.
__device__ __forceinline__ void l1_mark(const Geom &g, const L1Flags &f, uint32_t plane0, uint32_t idx) {
    uint32_t k, i, j;
    decomp(g, idx, k, i, j);
    if ((int)i < f.off_h && (int)j < f.off_w) return;  // a coarser level or the root block
    const uint32_t bi = (int)i >= f.off_h ? i - (uint32_t)f.off_h : i, bj = (int)j >= f.off_w ? j - (uint32_t)f.off_w : j;
    if (bi >= (uint32_t)f.band_h || bj >= (uint32_t)f.band_w) return;  // a padding cell: the transform never reads it
    constexpr uint32_t TR = IW_TH / 2, TC = IW_TW / 2;
    const uint32_t ty = bi / TR, tx = bj / TC;
    const bool up = bi - ty * TR < (uint32_t)f.hf1 && ty > 0, lf = bj - tx * TC < (uint32_t)f.hf1 && tx > 0;
    const uint32_t gx = (uint32_t)f.gx, gy = (uint32_t)f.gy;
    uint32_t *p = f.p + (size_t)(plane0 + k) * gy * gx;
    if (ty < gy && tx < gx) p[ty * gx + tx] = 1u;
    if (up && ty - 1u < gy && tx < gx) p[(ty - 1u) * gx + tx] = 1u;
    if (lf && ty < gy && tx - 1u < gx) p[ty * gx + tx - 1u] = 1u;
    if (up && lf && ty - 1u < gy && tx - 1u < gx) p[(ty - 1u) * gx + tx - 1u] = 1u;
}

// ---- duplicated cells ----
// get_offspring (encoder_decoder.rs:43-75) maps root (i,j) to the 2x2 block at rows (i&1)*ll_h + (i&~1) .. +1, columns
// likewise.  With an odd ll_h the blocks of the even roots reach row ll_h, where the blocks of the odd roots start:
// a first-generation node in row ll_h (and, the same way, in column ll_w when ll_w is odd) whose other coordinate
// lies in the odd roots' range is the offspring of two roots (of three at the corner), and so is every node of its
// sub-tree.  Each instance travels through the lists on its own.
__device__ __forceinline__ bool dup_cell(const Geom &g, uint32_t idx) {
    uint32_t k, i, j;
    decomp(g, idx, k, i, j);
    const uint32_t lh = (uint32_t)g.ll_h, lw = (uint32_t)g.ll_w;
    // cheap rejection: a node of such a sub-tree has the bits of ll_h (ll_w) at the top of its row (column) index
    const int si = (int)__clz((int)lh) - (int)__clz((int)i), sj = (int)__clz((int)lw) - (int)__clz((int)j);
    const bool ri = (lh & 1u) && i >= lh && (i >> si) == lh, rj = (lw & 1u) && j >= lw && (j >> sj) == lw;
    if (!ri && !rj) return false;
    // first-generation ancestor: halve while the index-doubling parent is outside the root block
    while (!((i >> 1) < lh && (j >> 1) < lw)) { i >>= 1; j >>= 1; }
    const bool io = i >= lh && (((i - lh) & ~1u) + 1u) < lh;  // row of an odd root's block
    const bool jo = j >= lw && (((j - lw) & ~1u) + 1u) < lw;
    return ((lh & 1u) && i == lh && jo) || ((lw & 1u) && j == lw && io);
}

// Final value of a cell owned by `cnt` LSP entries (ascending LSP index = stream order of their first writes), from
// the entries' private values: entry X was written at plane nX = msb|vX| with the sign of vX, and bit m < nX of |vX|
// is the refinement bit X read at plane m.  The reference applies all of it to ONE cell (encoder_decoder.rs:364-372,
// 396-404: rec = +-1.5*2^n; :443: rec = set_bit(rec, n, bit)), plane by plane: the writes of the LIP and LIS passes
// (LSP order), then the refinement pass over the entries appended in earlier planes (LSP order).
__device__ __forceinline__ int32_t replay_dup(const uint32_t (&ts)[3], const int32_t (&vs)[3], int cnt, uint32_t ref_plane,
                                              uint32_t ref_count) {
    int nx[3] = {-1, -1, -1};
    uint32_t mag[3] = {0, 0, 0};
    int top = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (q < cnt) {
            mag[q] = (uint32_t)(vs[q] < 0 ? -vs[q] : vs[q]);
            nx[q] = 31 - (int)__clz((int)mag[q]);
            top = nx[q] > top ? nx[q] : top;
        }
    }
    int32_t cur = 0;
    for (int m = top; m >= 0; --m) {
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q < cnt && nx[q] == m) {
                const int32_t base = m == 0 ? 1 : (int32_t)(3u << (m - 1));
                cur = vs[q] < 0 ? -base : base;
            }
        if ((uint32_t)m >= ref_plane) {
#pragma unroll
            for (int q = 0; q < 3; q++)
                if (q < cnt && nx[q] > m && ((uint32_t)m > ref_plane || ts[q] < ref_count)) cur = set_bit_i32(cur, (uint32_t)m, (mag[q] >> m) & 1u);
        }
    }
    return cur;
}

__device__ __forceinline__ uint32_t ld_l2(const uint32_t *p) {  // bypasses the L1: the word is the target of L2 atomics
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Brings the LSP entries of every duplicated cell together and writes the cell.  No hash table: the entries meet in
// the (still zero) cell itself -- atomicMax of LSP index + 1 elects the newest entry as owner -- and the others post
// their index into the owner's two mail words (mailA / mailB: the LIP buffers, free by now; indexed by LSP index).
// All waves of the workgroup; barriers between the rounds.
template <uint32_t IDXM>
__device__ __forceinline__ void resolve_dups(const Geom &g, int32_t *out, const uint32_t *lsp_idx, const int32_t *lsp_val,
                                             uint32_t lsp_len, uint32_t *mailA, uint32_t *mailB, uint32_t ref_plane,
                                             uint32_t ref_count, const uint32_t *dlist, uint32_t ndup) {
    // dlist[0, ndup): the LSP positions whose cell is a duplicated one, in no particular order, as the final scatter met
    // them (a small part of the list: five walks over ALL of it took longer than the scatter itself).  dlist == nullptr
    // (the list did not fit its buffer: tiny root blocks at full rate): every LSP position is tested instead.
    constexpr uint32_t NT = DEC_NW * 64, OWNER = 0xFFFFFFFFu;
    uint32_t *cells = reinterpret_cast<uint32_t *>(out);
    auto each = [&](auto f) {
        if (dlist) {
            for (uint32_t q = threadIdx.x; q < ndup; q += NT) {
                const uint32_t t = dlist[q];
                f(t, lsp_idx[t] & IDXM);
            }
        } else {
            for (uint32_t t = threadIdx.x; t < lsp_len; t += NT) {
                const uint32_t idx = lsp_idx[t] & IDXM;
                if (dup_cell(g, idx)) f(t, idx);
            }
        }
    };
    each([&](uint32_t t, uint32_t idx) {  // round 0: elect
        mailA[t] = 0;
        mailB[t] = 0;
        atomicMax(&cells[idx], t + 1u);
    });
    __syncthreads();
    each([&](uint32_t t, uint32_t idx) {  // round 1: the newest of the others
        const uint32_t t1 = ld_l2(&cells[idx]) - 1u;
        if (t1 != t && t1 < lsp_len) atomicMax(&mailA[t1], t + 1u);  // (t1 >= lsp_len: the caller's array was not zero --
                                                                      //  its output is wrong then, but nothing may fault)
    });
    __syncthreads();
    each([&](uint32_t t, uint32_t idx) {  // round 2: a third entry (corner cells)
        const uint32_t t1 = ld_l2(&cells[idx]) - 1u;
        if (t1 != t && t1 < lsp_len && ld_l2(&mailA[t1]) != t + 1u) atomicMax(&mailB[t1], t + 1u);
    });
    __syncthreads();
    each([&](uint32_t t, uint32_t idx) {  // round 3: the owner replays
        if (ld_l2(&cells[idx]) - 1u != t) return;
        uint32_t m2 = ld_l2(&mailA[t]), m3 = ld_l2(&mailB[t]);
        if (m2 > lsp_len) m2 = 0;  // (as above: only a dirty array can leave such a word)
        if (m3 > lsp_len) m3 = 0;
        // (no run-time index into the small arrays: the compiler would move them to LDS -- 24 bytes per thread of the block)
        const uint32_t t3 = m3 ? m3 - 1u : 0u, t2 = m2 ? m2 - 1u : 0u;
        const int32_t v3 = m3 ? lsp_val[t3] : 0, v2 = m2 ? lsp_val[t2] : 0, v1 = lsp_val[t];
        const int cnt = 1 + (m2 ? 1 : 0) + (m3 ? 1 : 0);
        // ascending LSP index: [m3,] [m2,] t
        const uint32_t ts[3] = {m3 ? t3 : (m2 ? t2 : t), m3 ? (m2 ? t2 : t) : (m2 ? t : 0u), (m3 && m2) ? t : 0u};
        const int32_t vs[3] = {m3 ? v3 : (m2 ? v2 : v1), m3 ? (m2 ? v2 : v1) : (m2 ? v1 : 0), (m3 && m2) ? v1 : 0};
        mailA[t] = (uint32_t)replay_dup(ts, vs, cnt, ref_plane, ref_count);
        mailB[t] = OWNER;
    });
    __syncthreads();  // every entry has read the election result: the cells can take their values
    each([&](uint32_t t, uint32_t idx) {  // round 4 (an owner reads back its own two stores)
        if (mailB[t] == OWNER) out[idx] = (int32_t)mailA[t];
    });
}

// ---- per-lane work of one LIP window (worker) ----
template <bool META>
__device__ __forceinline__ void work_lip(DecShared &sh, const DecArgs &a, const Item &it, const uint32_t *lip,
                                         uint32_t *lipn, uint32_t *lsp_idx, int32_t *lsp_val, uint32_t nbits,
                                         int n, int32_t base_val, uint32_t lane, const Trace &tr) {
    // the window's LIP entries first (it holds at most 64 tokens; it.m_rem of the list are left from e_start on)
    const uint32_t pre = lip[it.e_start + (lane < it.m_rem ? lane : it.m_rem - 1u)];
    const LipWin lw = lip_window(it.lo, it.Wb, it.pos0, it.cin, it.m_rem, nbits);
    const bool isS = (lw.S_in >> lane) & 1ull;
    const uint32_t rank = mbcnt(lw.S_in);
    const bool sig = (lw.sig >> lane) & 1ull;
    const uint32_t epick = (uint32_t)__shfl((int)pre, (int)rank);
    const uint32_t e = isS ? epick : 0u;
    if (META) {
        if (isS) {
            tr_put(tr, it.Wb + lane, 0, n, e);              // action 0 (:707)
            if (sig) tr_put(tr, it.Wb + lane + 1, 1, n, e);  // action 1 (:712)
        }
        if (lw.trunc && lane == 0) {  // '1' in the last bit of the stream: its sign row is the waiting one
            const uint32_t et = lip[it.e_start + (uint32_t)__popcll(lw.S_in)];
            tr_put(tr, it.Wb + lw.trunc_pos, 0, n, et);
            tr_put(tr, nbits, 1, n, et);
        }
    }
    if (isS && sig) {
        const uint32_t sgn = lane < 63 ? ((uint32_t)(it.lo >> (lane + 1)) & 1u) : ((uint32_t)it.hi & 1u);
        const uint32_t t = it.b_lsp + mbcnt(lw.sig);
        const int32_t v = sgn ? base_val : -base_val;
        if (t < a.caps.lsp) {
            lsp_idx[t] = e;
            lsp_val[t] = v;
        }
    } else if (isS) {
        lipn[it.b_lip + mbcnt(lw.S_in & ~lw.sig)] = e;
    }
}

// ---- per-lane work of one LIS window (worker); lane = stream position inside the window ----
// `pre`: queue entry it.e_start + lane, loaded by the caller before anything else of the window was looked at (a window
// of 64 bits holds at most 64 entries): the entry a lane needs is then a cross-lane read away instead of a second trip
// to memory behind the stream bits.
template <bool META>
__device__ __forceinline__ void work_lis(DecShared &sh, const DecArgs &a, const Geom &g, const Item &it, uint32_t seqno,
                                         uint32_t pre, uint32_t *nxt, uint32_t *ret, uint32_t *lip,
                                         uint32_t *lsp_idx, int32_t *lsp_val, uint32_t nbits, int n,
                                         int32_t base_val, uint32_t lane, const Trace &tr, uint64_t *wpf = nullptr) {
    const uint32_t W = (uint32_t)g.w, H = (uint32_t)g.h;
    constexpr uint32_t IDXM = META ? ENT_IDX_META : ENT_IDX;
    // entry starts: owned bits that are not child bits of a fired type-A entry starting in this window
    const uint64_t own = (it.pos1 >= 64 ? ~0ull : ((1ull << it.pos1) - 1ull)) & (~0ull << it.pos0);
    const uint64_t below = it.fm & ((1ull << lane) - 1ull);
    bool payload = false;
    if (below) {
        const uint32_t f = 63u - (uint32_t)__builtin_clzll(below);
        const uint64_t bb = f ? ((it.lo >> f) | (it.hi << (64 - f))) : it.lo;
        uint32_t pl = (uint32_t)(bb >> 1) & 0xFFu, ns = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t s = pl & 1u;
            pl >>= 1 + s;
            ns += s;
        }
        payload = lane < f + 5 + ns;
    }
    const bool isE = ((own >> lane) & 1ull) && !payload;
    const uint64_t ES = __ballot(isE);
    const uint32_t mypos = it.Wb + lane;
    const uint32_t epick = (uint32_t)__shfl((int)pre, (int)mbcnt(ES));
    const uint32_t e = isE ? epick : 0u;
    const uint32_t idx = e & IDXM;
    const bool isA = (e & ENT_A) != 0;
    const bool leaf = (e & ENT_LEAF) != 0;
    uint32_t nQ = 0, nR = 0, nLIP = 0, nLSP = 0;
    uint32_t sigm = 0, signm = 0, lipm = 0, cb = 0, cr = 0, ccol = 0;
    uint32_t cf = 0;  // META: filter of the offspring, in entry position (get_offspring_filter, :133-150)
    bool fired = false;
    const bool far = it.Wb + 64u + 16u <= nbits;  // (wave-uniform) the window's tokens end before the stream does
    if (META && isE) tr_put(tr, mypos, isA ? 2u : 5u, n, e);  // action 2 (:735) / action 5 (:787)
    if (isE) {
        const uint32_t avail = nbits - mypos;  // >= 1: the sequencer never places an entry at or past nbits
        const uint64_t bb = lane ? ((it.lo >> lane) | (it.hi << (64 - lane))) : it.lo;
        const uint32_t bits = (uint32_t)bb & 0xFFFFu;
        fired = bits & 1u;
        if (!fired) {
            nR = 1;
        } else if (leaf) {
            // a set with no offspring cannot be significant in a real stream; the reference drops the entry
            // (no offspring to read, has_descendents_past_offspring is false too)
            fired = false;
        } else {
            uint32_t k, ii, jj;
            decomp(g, idx, k, ii, jj);
            cb = child_base(g, k, ii, jj, cr, ccol);
            if (META) {
                const uint32_t pf = (e >> ENT_FILT_SHIFT) & 3u;
                const uint32_t fl = pf ? pf : (((ii & 1u) && (jj & 1u)) ? 3u : ((!(ii & 1u) && (jj & 1u)) ? 2u : 1u));
                cf = fl << ENT_FILT_SHIFT;
            }
            if (!isA) {
                nQ = 4;
            } else if (!META && far) {
                // every token of this window lies wholly inside the stream: no look at the end of it per offspring
                uint32_t o = 1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t sg = (bits >> o) & 1u;
                    sigm |= sg << q;
                    signm |= (sg & (bits >> (o + 1))) << q;
                    o += 1u + sg;
                }
                lipm = ~sigm & 15u;
                nLSP = (uint32_t)__popc(sigm);
                nLIP = 4u - nLSP;
                nQ = (4 * ii + 3 < H && 4 * jj + 3 < W) ? 1u : 0u;  // :411-414
            } else {
                bool stop = false;
                uint32_t o = 1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!stop) {
                        if (META) {
                            const uint32_t ch = (cb + (q >> 1) * W + (q & 1)) | cf;
                            tr_put(tr, mypos + o, 3, n, ch);                                            // action 3 (:745)
                            if (o < avail && ((bits >> o) & 1u)) tr_put(tr, mypos + o + 1, 4, n, ch);  // action 4 (:749)
                        }
                        if (o >= avail) { stop = true; }
                        else if ((bits >> o) & 1u) {
                            if (o + 1 >= avail) { stop = true; }
                            else {
                                sigm |= 1u << q;
                                signm |= ((bits >> (o + 1)) & 1u) << q;
                                o += 2;
                            }
                        } else {
                            lipm |= 1u << q;
                            o += 1;
                        }
                    }
                }
                nLSP = (uint32_t)__popc(sigm);
                nLIP = (uint32_t)__popc(lipm);
                if (!stop) nQ = (4 * ii + 3 < H && 4 * jj + 3 < W) ? 1u : 0u;  // :411-414
            }
        }
    }
    // exclusive prefix sums of the small per-lane counts, bit-sliced through ballots
    const uint64_t mR = __ballot(nR != 0);
    const uint64_t mQ1 = __ballot(nQ == 1), mQ4 = __ballot(nQ == 4);
    const uint64_t mS0 = __ballot(nLSP & 1u), mS1 = __ballot(nLSP & 2u), mS2 = __ballot(nLSP & 4u);
    const uint64_t mL0 = __ballot(nLIP & 1u), mL1 = __ballot(nLIP & 2u), mL2 = __ballot(nLIP & 4u);
    const uint32_t tR = (uint32_t)__popcll(mR);
    const uint32_t tQ = (uint32_t)__popcll(mQ1) + 4u * (uint32_t)__popcll(mQ4);
    const uint32_t tLSP = (uint32_t)__popcll(mS0) + 2u * (uint32_t)__popcll(mS1) + 4u * (uint32_t)__popcll(mS2);
    const uint32_t tLIP = (uint32_t)__popcll(mL0) + 2u * (uint32_t)__popcll(mL1) + 4u * (uint32_t)__popcll(mL2);
    // running list lengths: from the item for the first window of a generation, else from the previous window
    uint32_t b_lsp, b_lip, b_nxt, b_ret;
    if (it.first) {
        b_lsp = it.b_lsp; b_lip = it.b_lip; b_nxt = it.b_nxt; b_ret = it.b_ret;
    } else {
        Chain *pc = &sh.chain[(seqno - 1) % DEC_RING];
        uint32_t spins = 0;
#ifdef DEC_PROF
        const uint64_t tc0 = __builtin_amdgcn_s_memtime();
#endif
        while (lds_load(&pc->seq) != seqno) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { sh.bad = 2; break; }
        }
#ifdef DEC_PROF
        if (wpf) wpf[2] += __builtin_amdgcn_s_memtime() - tc0;
#endif
        b_lsp = pc->lsp; b_lip = pc->lip; b_nxt = pc->nxt; b_ret = pc->ret;
    }
    {
        Chain *mc = &sh.chain[seqno % DEC_RING];
        if (lane == 0) { mc->lsp = b_lsp + tLSP; mc->lip = b_lip + tLIP; mc->nxt = b_nxt + tQ; mc->ret = b_ret + tR; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_store(&mc->seq, seqno + 1);
    }
    if (b_nxt + tQ > a.caps.lis || b_ret + tR > a.caps.lis || b_lip + tLIP > a.caps.lip || b_lsp + tLSP > a.caps.lsp) {
        if (lane == 0) sh.bad = 1;
        return;
    }
    if (isE) {
        if (nR) {
            ret[b_ret + mbcnt(mR)] = e;
        } else if (!fired) {
            // fired leaf: dropped
        } else if (!isA) {
            const uint32_t oq = b_nxt + mbcnt(mQ1) + 4u * mbcnt(mQ4);
            nxt[oq] = make_a_entry(cb, cr, ccol, H, W) | cf;
            nxt[oq + 1] = make_a_entry(cb + 1, cr, ccol + 1, H, W) | cf;
            nxt[oq + 2] = make_a_entry(cb + W, cr + 1, ccol, H, W) | cf;
            nxt[oq + 3] = make_a_entry(cb + W + 1, cr + 1, ccol + 1, H, W) | cf;
        } else {
            uint32_t ol = b_lip + mbcnt(mL0) + 2u * mbcnt(mL1) + 4u * mbcnt(mL2);
            uint32_t os = b_lsp + mbcnt(mS0) + 2u * mbcnt(mS1) + 4u * mbcnt(mS2);
            if (!META && far) {
                // every offspring goes to one of the two lists: one store of its index to the address picked, the value behind it
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t ci = cb + (q >> 1) * W + (q & 1);
                    const bool sg = (sigm >> q) & 1u;
                    uint32_t *dst = sg ? lsp_idx + os : lip + ol;
                    *dst = ci;
                    if (sg) lsp_val[os] = ((signm >> q) & 1u) ? base_val : -base_val;
                    os += sg ? 1u : 0u;
                    ol += sg ? 0u : 1u;
                }
            } else
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t ci = (cb + (q >> 1) * W + (q & 1)) | cf;
                if (sigm & (1u << q)) {
                    lsp_idx[os] = ci;
                    lsp_val[os] = ((signm >> q) & 1u) ? base_val : -base_val;
                    os++;
                } else if (lipm & (1u << q)) {
                    lip[ol++] = ci;
                }
            }
            if (nQ) nxt[b_nxt + mbcnt(mQ1) + 4u * mbcnt(mQ4)] = e & ~(ENT_A | ENT_LEAF);  // type B
        }
    }
}

// worker loop of one phase: consume items until the sequencer closes the phase
template <bool META>
__device__ __forceinline__ void worker_phase(DecShared &sh, const DecArgs &a, const Geom &g, uint32_t &myk, uint32_t par,
                                             const uint32_t *lip_rd, uint32_t *lip_wr, uint32_t *lip_app,
                                             const uint32_t *cur, uint32_t cur_len, uint32_t *nxt, uint32_t *ret,
                                             uint32_t *lsp_idx, int32_t *lsp_val, uint32_t nbits, int n,
                                             int32_t base_val, uint32_t wk, uint32_t lane, const Trace &tr,
                                             const BitSrc &bs, uint64_t *wpf = nullptr) {
    // The window's queue entries are its one trip to memory, and beside an HBM-bound kernel that trip is what a window
    // costs (round 4: the sequencer waited for ring space a fifth of its time there): the entries of this worker's NEXT
    // window are requested, if the sequencer has published it already, before the current one is worked on.
    bool have_next = false;
    uint32_t pre_next = 0;
    for (;;) {
        bool got = false;
        uint32_t spins = 0;
        uint32_t *rdy = &sh.ring[myk % DEC_RING].ready;
#ifdef DEC_PROF
        const uint64_t tw0 = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
            // relaxed workgroup-scope load = plain ds_read (a volatile read through the generic pointer is a flat load)
            if (__hip_atomic_load(rdy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == myk + 1) {
                asm volatile("" ::: "memory");  // the item is read after its flag (hardware: one wave's LDS reads are in order)
                got = true;
                break;
            }
            const uint32_t pe = lds_load(&sh.phase_end[par]);
            if (pe != SEQ_OPEN && pe <= myk) break;
            __builtin_amdgcn_s_sleep(DEC_WSLEEP);
            if (++spins > SPIN_LIMIT) { sh.bad = 2; break; }
        }
        if (!got) break;
#ifdef DEC_PROF
        const uint64_t tw1 = __builtin_amdgcn_s_memtime();
        if (wpf) { wpf[0] += tw1 - tw0; wpf[4] += 1; }
#endif
        // the window's queue entries first (the only trip to memory of the window), then its bits from LDS
        uint32_t pre;
        if (have_next) pre = pre_next;
        else {
            const uint32_t e0 = sh.ring[myk % DEC_RING].e_start + lane;
            pre = cur[e0 < cur_len ? e0 : cur_len - 1u];
        }
        {
            const uint32_t kn = myk + DEC_NWK;
            Slot *sn = &sh.ring[kn % DEC_RING];
            have_next = __hip_atomic_load(&sn->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kn + 1;
            asm volatile("" ::: "memory");
            if (have_next) {
                const uint32_t e1 = sn->e_start + lane;
                pre_next = cur[e1 < cur_len ? e1 : cur_len - 1u];
            }
        }
        const Item it = slot_unpack(sh, bs, myk);
#ifdef DEC_PROF
        {   // what the compiler's own wait in front of the first use of `pre` costs: everything this wavefront has in flight
            const uint64_t tv0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wpf) wpf[1] += __builtin_amdgcn_s_memtime() - tv0;
        }
        work_lis<META>(sh, a, g, it, myk, pre, nxt, ret, lip_app, lsp_idx, lsp_val, nbits, n, base_val, lane, tr, wpf);
        if (wpf) wpf[3] += __builtin_amdgcn_s_memtime() - tw1;
#else
        work_lis<META>(sh, a, g, it, myk, pre, nxt, ret, lip_app, lsp_idx, lsp_val, nbits, n, base_val, lane, tr);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_store(&sh.wdone[wk], myk / DEC_NWK + 1);
        myk += DEC_NWK;
    }
}

// helper loop of one LIS phase (wavefront 1): window j of the phase is stream window widx0 + j.  Batches of 32 windows, lane =
// (window, half).  Token lengths first: the length of a fired type-A entry at position q is 5 + the significant ones among
// the four offspring that follow ('0' or '1' + sign bit each) -- worked out for all 64 positions of the lane's window at once
// with bitwise operations on 64-bit words (the k-th offspring's bit is one of the window shifted by k .. 2k-1, chosen by the
// two-bit count of significant ones before it), which leaves (length - 5) as three bit planes: some 80 instructions for 32
// windows where one lane per position took 40 per window (round 4: the helper, not the sequencer, set the decoder's pace in
// long stretches of looked-up windows).  Then the window walked under the all-type-A hypothesis (DecShared::tab): from
// entry point 0 in full -- a dozen steps at most, a fired entry takes five bits or more -- and four of the other eight entry
// points (half 0: 1..4, half 1: 5..8) only until they fall in step with that walk, which they do after a token or two (the
// code synchronises itself).  The windows are announced before their table (DecShared::pprog, tprog): the sequencer never waits
// for a table, it hops through a window whose table is not there yet (a phase's first few).
__device__ __forceinline__ void helper_phase(DecShared &sh, const BitSrc &bs, uint32_t par, uint32_t widx0, uint32_t lane) {
    static_assert(DEC_PREP_B == 32 && DEC_PREP % DEC_PREP_B == 0 && DEC_PREP >= 2 * DEC_PREP_B, "helper batches");
    const uint32_t uw = lane & 31u, hf = lane >> 5;
    // the words of the batch to come are fetched while the batch before is worked on
    uint64_t nlo = stream_word64(bs, widx0 + uw), nhi = stream_word64(bs, widx0 + uw + 1u);
    for (uint32_t j = 0;; j += DEC_PREP_B) {
        uint32_t spins = 0;
        for (;;) {  // room for another batch, or the end of the phase
            if (lds_load(&sh.phase_end[par]) != SEQ_OPEN) return;
            if (j + DEC_PREP_B <= lds_load(&sh.sprog) + DEC_PREP) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { sh.bad = 2; return; }
        }
        const uint64_t lo = nlo, hi = nhi;
        nlo = stream_word64(bs, widx0 + j + DEC_PREP_B + uw);
        nhi = stream_word64(bs, widx0 + j + DEC_PREP_B + uw + 1u);
        // X(k): bit q = stream bit q + k of the window
#define HX(k) ((lo >> (k)) | (hi << (64 - (k))))
        const uint64_t s1 = HX(1);
        const uint64_t s2 = (s1 & HX(3)) | (~s1 & HX(2));
        const uint64_t a0 = s1 ^ s2, a1 = s1 & s2;                                       // s1 + s2
        const uint64_t s3 = (a1 & HX(5)) | (~a1 & ((a0 & HX(4)) | (~a0 & HX(3))));
        const uint64_t b0 = a0 ^ s3, b1 = a1 | (a0 & s3);                                // s1 + s2 + s3
        const uint64_t s4 = (b1 & ((b0 & HX(7)) | (~b0 & HX(6)))) | (~b1 & ((b0 & HX(5)) | (~b0 & HX(4))));
        const uint64_t cy = b0 & s4;
        const uint64_t e0 = b0 ^ s4, e1 = b1 ^ cy, e2 = b1 & cy;                         // the four of them: 0 .. 4
#undef HX
        const uint32_t tslot = (j + uw) % DEC_PREP;
        if (hf == 0) {
            uint64_t *row = sh.pln[tslot];
            row[0] = e0; row[1] = e1; row[2] = e2; row[3] = lo;
            uint64_t *wb = sh.wbits[(j + uw) % (DEC_PREP + DEC_RING)];
            wb[0] = lo; wb[1] = hi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_store(&sh.pprog, j + DEC_PREP_B);  // the sequencer can hop through these windows from here on
        auto tok_len = [&](uint32_t q) -> uint32_t {  // length of the fired entry that starts at bit q
            return 5u + ((uint32_t)(e0 >> q) & 1u) + 2u * ((uint32_t)(e1 >> q) & 1u) + 4u * ((uint32_t)(e2 >> q) & 1u);
        };
        // entry point 0: S0 = every bit at which an entry starts, fm0 = the fired ones among them
        uint64_t S0 = 0, fm0 = 0;
        uint32_t p = 0;
        while (p < 64u) {
            const uint64_t rem = lo >> p;
            if (rem == 0) { S0 |= ~0ull << p; p = 64u; break; }  // zeros to the end: one-bit entries
            const uint32_t q = p + (uint32_t)__builtin_ctzll(rem);
            S0 |= ((2ull << (q - p)) - 1ull) << p;  // the zeros p .. q-1 and the fired entry at q
            fm0 |= 1ull << q;
            p = q + tok_len(q);
        }
        const uint32_t exit0 = p - 64u;
        if (hf == 0) {
            sh.tab[tslot].fm[0] = fm0;
            sh.tab[tslot].ce[0] = (uint16_t)((uint32_t)__popcll(S0) | (exit0 << 7));
        }
        for (uint32_t o = 1u + 4u * hf; o < 5u + 4u * hf; o++) {
            uint32_t pp = o, cnt = 0, ex = 0;
            uint64_t fmk = 0;
            for (;;) {
                if (pp >= 64u) { ex = pp - 64u; break; }  // never met the walk from 0
                // one-bit entries up to the next bit that is set or lies on the walk from 0
                const uint32_t z = (uint32_t)__builtin_ctzll(((lo | S0) >> pp) | (1ull << (63u - pp)));
                cnt += z;
                pp += z;
                if ((S0 >> pp) & 1ull) {  // in step from here on
                    cnt += (uint32_t)__popcll(S0 >> pp);
                    fmk |= fm0 & (~0ull << pp);
                    ex = exit0;
                    break;
                }
                if (!((lo >> pp) & 1ull)) { cnt += 1; ex = 0; pp = 64u; break; }  // (bit 63, a zero off the walk: the last entry)
                fmk |= 1ull << pp;
                cnt += 1;
                pp += tok_len(pp);
            }
            sh.tab[tslot].fm[o] = fmk;
            sh.tab[tslot].ce[o] = (uint16_t)(cnt | (ex << 7));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_store(&sh.tprog, j + DEC_PREP_B);  // ... and look them up from here on
    }
}

// the sequencer closes a phase: results first, then the slot of the NEXT phase is opened, then this one ends
__device__ __forceinline__ void seq_close(DecShared &sh, uint32_t par, uint32_t seq, uint32_t P, uint32_t dn, uint32_t lsp,
                                          uint32_t lipl, uint32_t lane) {
    if (lane == 0) {
        sh.r_P = P; sh.r_done = dn; sh.r_lsp = lsp; sh.r_lip = lipl;
        sh.head = seq;
        sh.phase_end[par ^ 1u] = SEQ_OPEN;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_store(&sh.phase_end[par], seq);
}

template <bool META>
#ifndef DEC_WAVES_PER_EU
#define DEC_WAVES_PER_EU 5  // caps the kernel at 96 VGPRs (35 spilled, none on the sequencer's path: same speed alone,
#endif                      // 6.95 ms) so that three DWT wavefronts per SIMD fit beside a resident decoder workgroup when
                            // the two run on different streams (bench.py --pipeline 1: 19.6 instead of 22.5 ms per step)
#ifdef DEC_NUM_VGPR  // A/B builds: an exact register budget between the two occupancy steps
__global__ __launch_bounds__(DEC_NW * 64) __attribute__((amdgpu_waves_per_eu(4, 5), amdgpu_num_vgpr(DEC_NUM_VGPR)))
#else
__global__ __launch_bounds__(DEC_NW * 64) __attribute__((amdgpu_waves_per_eu(DEC_WAVES_PER_EU, DEC_WAVES_PER_EU)))
#endif
void k_decode(DecArgs a) {
    __shared__ DecShared sh;
    constexpr uint32_t IDXM = META ? ENT_IDX_META : ENT_IDX;
    const Geom g = a.g;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (uniform: the roles below are scalar branches, what derives from it stays in SGPRs)
    const uint32_t slot = blockIdx.x;
    const uint32_t W = (uint32_t)g.w;

    uint32_t *lipA = a.lip0 + (size_t)slot * a.caps.lip;
    uint32_t *lipB = a.lip1 + (size_t)slot * a.caps.lip;
    uint32_t *lsp_idx = a.lsp_idx + (size_t)slot * a.caps.lsp;
    int32_t *lsp_val = a.lsp_val + (size_t)slot * a.caps.lsp;
    uint32_t *q0 = a.lis0 + (size_t)slot * a.caps.lis;
    uint32_t *q1 = a.lis1 + (size_t)slot * a.caps.lis;
    uint32_t *q2 = a.lis2 + (size_t)slot * a.caps.lis;

    for (int b = (int)blockIdx.x; b < a.B; b += (int)gridDim.x) {
        int32_t *__restrict__ out = a.out + (size_t)b * g.n;
        BitSrc bs;
        bs.gw = reinterpret_cast<const uint32_t *>(a.data + (size_t)b * a.slot_stride);
        bs.nwords = (uint32_t)(a.slot_stride >> 2);
        uint64_t nby = a.nbytes[b];
        bool bad = false;
        if (nby * 8 >= 0xFFFFFF00ull) { bad = true; nby = 0; }
        if (nby > a.slot_stride) { bad = true; nby = 0; }
        bs.nbits = (uint32_t)(nby * 8);
        const uint32_t nbits = bs.nbits;
        Trace tr;
        tr.ent = META ? a.tr_ent + (size_t)b * a.tr_stride : nullptr;
        tr.act = META ? a.tr_act + (size_t)b * a.tr_stride : nullptr;
        tr.nbits = nbits;
        int n = (int)a.max_n[b];
        if (n > 30) { bad = true; n = 0; }

        // every wave keeps its own copy of the (uniform) decoder state; the sequencer's results travel through LDS
        uint32_t *lip = lipA, *lipn = lipB;
        uint32_t *lis = q0, *qa = q1, *qb = q2;
        uint32_t lip_len = 0, lsp_len = 0, lis_len = 0;
        uint32_t P = 0;
        // refinement passes that ran: every plane above ref_plane completely, plane ref_plane for the first ref_count
        // LSP entries, no plane below (resolve_dups replays duplicated cells from this)
        uint32_t ref_plane = 32, ref_count = 0;
        uint32_t seq = 0;                    // items produced so far (kept in step by every wave at phase ends)
        uint32_t myk = DEC_IS_WORKER(wave) ? DEC_WK(wave) : 0;  // worker: sequence number of its next item
        uint32_t phase = 0;
#ifdef DEC_PROF
        uint64_t pf[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t wpf[5] = {0, 0, 0, 0, 0};  // worker 0: waiting for a window, for memory, for the chain; in windows in all; windows
        uint64_t pt = __builtin_amdgcn_s_memtime();
        const uint64_t pf_rt0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, whatever the shader clock does
#endif

        __syncthreads();  // previous image fully finished with the shared state
        if (threadIdx.x == 0) {
            sh.head = 0; sh.phase_end[0] = SEQ_OPEN; sh.phase_end[1] = SEQ_OPEN; sh.bad = 0; sh.ndup = 0;
            sh.pprog = 0; sh.sprog = 0; sh.tprog = 0;
            for (int w = 0; w < DEC_NWK; w++) sh.wdone[w] = 0;
            for (int r = 0; r < DEC_RING; r++) { sh.chain[r].seq = 0; sh.ring[r].ready = 0; }
        }

        // ---- initial LIP / LIS (encoder_decoder.rs:327-348), wave 0 ----
        const uint32_t nroot = (uint32_t)(g.ll_h * g.ll_w * g.c);
        if (wave == 0) {
            uint32_t ll = 0;
            for (uint32_t base = 0; base < nroot; base += 64) {
                uint32_t t = base + lane;
                bool act = t < nroot;
                uint32_t k = 0, i = 0, j = 0;
                if (act) {
                    uint32_t ij = t / (uint32_t)g.c;
                    k = t - ij * (uint32_t)g.c;
                    i = ij / (uint32_t)g.ll_w;
                    j = ij - i * (uint32_t)g.ll_w;
                }
                uint32_t idx = k * g.hw + i * W + j;
                bool inlis = act && (((i | j) & 1u) != 0);
                uint64_t m = __ballot(inlis);
                if (act && t < a.caps.lip) lip[t] = idx;
                uint32_t o = ll + mbcnt(m);
                if (inlis && o < a.caps.lis) lis[o] = idx | ENT_A;
                ll += (uint32_t)__popcll(m);
            }
            if (lane == 0) sh.r_lip = ll;
        }
        __syncthreads();
        lis_len = sh.r_lip;
        lip_len = nroot;
        if (lip_len > a.caps.lip || lis_len > a.caps.lis) bad = true;
        __syncthreads();

        // the sequencer is the serial chain of the whole image: when other kernels (or other images' workers) share
        // its SIMD it must not queue for issue slots behind them
#ifndef DEC_SEQ_NOPRIO
        if (wave == 0) __builtin_amdgcn_s_setprio(3);
#endif
        bool done = bad;
        for (; !done; --n) {
            const uint32_t lsp_len0 = lsp_len;
            const int32_t base_val = (n == 0) ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));  // :364-370

            // ================= LIP pass (encoder_decoder.rs:355-377) =================
            // No sequencer here: the only serial dependence between 64-bit windows is one carry bit (does the window
            // start on a pending sign bit?), so DEC_NW*64 windows are scanned at once, one per lane: both carry-in
            // hypotheses per window, a block scan of the carry functions, then block scans of the token / LSP / LIP
            // counts give every window its exact list offsets; afterwards each wavefront works through its 64 windows.
            {
                PF_ADD(7);
                uint32_t m_rem = lip_len, tok_base = 0, lipn_len = 0, cin = 0, lsp_l = lsp_len;
                bool dn = false;
                while (m_rem > 0 && !dn) {
                    if (P >= nbits) {
                        if (META && threadIdx.x == 0) tr_put(tr, nbits, 0, n, lip[tok_base]);  // waiting for a LIP bit
                        dn = true;
                        break;
                    }
                    const uint32_t widx0 = P >> 6, gw = wave * 64 + lane;
                    const uint32_t widx = widx0 + gw, Wb = widx << 6;
                    const uint32_t pos = gw == 0 ? (P & 63u) : 0u;
                    const uint64_t lo = (uint64_t)stream_word(bs, 2 * widx) | ((uint64_t)stream_word(bs, 2 * widx + 1) << 32);
                    const uint32_t hb = stream_word(bs, 2 * widx + 2) & 1u;
                    // carry function of this window: cout for cin = 0 and for cin = 1 (window 0's cin is known)
                    uint32_t c0, c1;
                    (void)lip_starts(lo, pos, 0, c0);
                    (void)lip_starts(lo, pos, 1, c1);
                    if (gw == 0) { if (cin) c0 = c1; else c1 = c0; }
                    uint32_t F = c0 | (c1 << 1);  // bit x = cout when cin = x
                    // inclusive scan of function composition along the lanes (H = G after F: H(x) = G(F(x)))
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t Fp = (uint32_t)__shfl_up((int)F, o);
                        if (lane >= (uint32_t)o) F = ((F >> (Fp & 1u)) & 1u) | (((F >> ((Fp >> 1) & 1u)) & 1u) << 1);
                    }
                    if (lane == 63) sh.wfun[wave] = F;
                    __syncthreads();
                    uint32_t wcin = cin;  // carry into this wave's first window
                    for (uint32_t w = 0; w < wave; w++) wcin = (sh.wfun[w] >> wcin) & 1u;
                    uint32_t Fprev = (uint32_t)__shfl_up((int)F, 1);
                    const uint32_t my_cin = lane == 0 ? wcin : ((Fprev >> wcin) & 1u);
                    const uint32_t my_cout = (F >> wcin) & 1u;
                    // tokens that start in this window (stream end respected), then their prefix over the block
                    uint32_t cdummy;
                    uint64_t S = lip_starts(lo, pos, my_cin, cdummy);
                    const uint32_t vb = Wb >= nbits ? 0u : ((nbits - Wb) < 64u ? (nbits - Wb) : 64u);
                    if (vb < 64) S &= vb ? ((1ull << vb) - 1ull) : 0ull;
                    const uint32_t cnt = (uint32_t)__popcll(S);
                    uint64_t tot1;
                    const uint64_t ex1 = block_exscan(sh, (uint64_t)cnt, tot1, wave, lane);
                    const uint32_t before = (uint32_t)ex1;
                    const uint32_t m_here = m_rem > before ? m_rem - before : 0u;
                    const bool in_pass = m_here > 0 && vb > 0;
                    LipWin lw;
                    lw.S_in = 0; lw.sig = 0; lw.ntok = 0; lw.trunc = 0; lw.trunc_pos = 0; lw.ends = 0; lw.pos1 = 64; lw.cout = 0;
                    if (in_pass) lw = lip_window(lo, Wb, pos, my_cin, m_here, nbits);
                    const uint32_t nsig = (uint32_t)__popcll(lw.sig);
                    const uint32_t nnon = (uint32_t)__popcll(lw.S_in & ~lw.sig);
                    uint64_t tot2;
                    const uint64_t ex2 = block_exscan(sh, (uint64_t)nsig | ((uint64_t)nnon << 20) | ((uint64_t)lw.ntok << 40), tot2,
                                                      wave, lane);
                    {
                        LipItem &q = sh.lipq[gw];
                        q.lo = lo; q.Wb = Wb; q.e_start = tok_base + before; q.m_rem = m_here;
                        q.b_lsp = lsp_l + ((uint32_t)ex2 & 0xFFFFFu);
                        q.b_lip = lipn_len + ((uint32_t)(ex2 >> 20) & 0xFFFFFu);
                        q.flags = pos | (my_cin << 8) | (hb << 9) | ((in_pass ? 1u : 0u) << 10);
                    }
                    if (threadIdx.x == 0) sh.lp_end[0] = 0;
                    __syncthreads();
                    if (in_pass && (lw.ends || lw.trunc)) {  // at most one window
                        sh.lp_end[0] = lw.trunc ? 2u : 1u;
                        sh.lp_end[1] = Wb + lw.pos1;
                    } else if (in_pass && lw.ntok > 0 && before + lw.ntok == m_rem) {
                        // the pass completes exactly with this window's last token (whose sign bit may be the next
                        // window's bit 0: carried in lp_end[3])
                        sh.lp_end[0] = 3u;
                        sh.lp_end[1] = Wb + 64;
                        sh.lp_end[3] = my_cout;
                    }
                    if (gw == DEC_NW * 64 - 1) sh.lp_end[2] = my_cout;
                    const uint32_t t_sig = (uint32_t)tot2 & 0xFFFFFu, t_non = (uint32_t)(tot2 >> 20) & 0xFFFFFu;
                    const uint32_t t_tok = (uint32_t)(tot2 >> 40);
                    if (lsp_l + t_sig > a.caps.lsp && threadIdx.x == 0) sh.bad = 1;
                    __syncthreads();
                    // ---- per-window work ----
                    if (!sh.bad) {
                        for (uint32_t qn = wave; qn < DEC_NW * 64; qn += DEC_NW) {  // round robin: short passes use all waves
                            const LipItem q = sh.lipq[qn];
                            if (!((q.flags >> 10) & 1u)) break;  // windows of a pass are contiguous
                            Item it;
                            it.kind = 0; it.Wb = q.Wb; it.pos0 = q.flags & 0xFFu; it.pos1 = 64; it.e_start = q.e_start;
                            it.cin = (q.flags >> 8) & 1u; it.m_rem = q.m_rem; it.first = 1; it.b_lsp = q.b_lsp; it.b_lip = q.b_lip;
                            it.b_nxt = 0; it.b_ret = 0; it.lo = q.lo; it.hi = (q.flags >> 9) & 1u; it.fm = 0;
                            work_lip<META>(sh, a, it, lip, lipn, lsp_idx, lsp_val, nbits, n, base_val, lane, tr);
                        }
                    }
                    // ---- advance (every wave computes the same) ----
                    lsp_l += t_sig;
                    lipn_len += t_non;
                    const uint32_t endk = sh.lp_end[0];
                    if (endk == 2) { dn = true; cin = 0; }
                    else if (endk == 1) { P = sh.lp_end[1]; m_rem = 0; cin = 0; }
                    else if (endk == 3) { P = sh.lp_end[1]; m_rem = 0; cin = sh.lp_end[3]; }
                    else {
                        m_rem -= t_tok;
                        tok_base += t_tok;
                        P = (widx0 + DEC_NW * 64) << 6;
                        cin = sh.lp_end[2];
                    }
                    if (sh.bad) dn = true;
                    __syncthreads();
                }
                if (cin && !dn) P += 1;
                lsp_len = lsp_l;
                lip_len = lipn_len;
                if (dn) done = true;
                { uint32_t *t = lip; lip = lipn; lipn = t; }
                // the LIP scratch overlays the LIS ring: its flags have to start from "nothing published" again (the
                // barrier that ends every round above has already put all waves past their last read of lipq)
                if (threadIdx.x < DEC_RING) { sh.ring[threadIdx.x].ready = 0; sh.chain[threadIdx.x].seq = 0; }
                __syncthreads();
                PF_ADD(0);
            }
            if (done) break;

            // ================= LIS pass (encoder_decoder.rs:379-436) =================
            uint32_t *cur = lis, *nxt = qa, *ret = qb;
            uint32_t cur_len = lis_len, ret_len = 0;
            while (cur_len > 0 && !done) {
                const uint32_t par = phase & 1u;
                const uint32_t seq0 = seq;  // every wave enters the phase with seq == head
                if (wave == 0) {
                    uint32_t i = 0, dn = 0;
                    // Every value the walk depends on is wave-uniform; v_readfirstlane tells hipcc so (values that came
                    // from LDS or global memory are otherwise kept in VGPRs and every branch on them goes through EXEC).
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
                    uint32_t sP = RFL(P), sSeq = RFL(seq);
                    const uint32_t widx0s = sP >> 6;  // first window of the phase (the helper starts there too)
                    uint32_t pknown = 0;              // windows known to be prepared
                    uint32_t tknown = 0;              // ... and to have their all-type-A table (DecShared::tab)
                    uint32_t sFc = (0u - sSeq) & (DEC_RING / 2 - 1);  // publishes until the next ring-space check
                    const uint32_t cur_len_v = cur_len;
                    const uint32_t sCur = RFL(cur_len_v), sNb = RFL(nbits);
                    const uint32_t sLsp = RFL(lsp_len), sLip = RFL(lip_len), sRet = RFL(ret_len);
                    // Type mask of queue entries [Ebase, Ebase+64): bit = type A with offspring.  The masks of 64 chunks
                    // (4096 entries) are built at once, 16 coalesced loads in flight at a time, and kept one per lane
                    // in TAv; moving on to the next chunk is then two v_readlane instead of a load the walk would
                    // have to wait for (a rotating register prefetch does not work: hipcc waits for the newest load
                    // before it moves the older registers along).
                    const uint32_t last_e = sCur - 1;
                    uint64_t TAv = 0;        // lane l: mask of chunk blk0 + l
                    uint32_t blk0 = 0;       // first chunk of the block held in TAv
                    auto build_block = [&](uint32_t c0) {
                        const uint32_t nch_all = (sCur + 63u) >> 6;
                        const uint32_t nch = (nch_all - c0) < 64u ? (nch_all - c0) : 64u;
                        TAv = 0;
                        for (uint32_t t0 = 0; t0 < nch; t0 += 16) {
                            uint32_t e[16];
#pragma unroll
                            for (int u = 0; u < 16; u++) {
                                const uint32_t ix = ((c0 + t0 + (uint32_t)u) << 6) + lane;
                                e[u] = cur[ix < last_e ? ix : last_e];
                            }
#pragma unroll
                            for (int u = 0; u < 16; u++) {
                                const uint64_t m = __ballot((e[u] & ENT_A) && !(e[u] & ENT_LEAF));
                                TAv = (lane == t0 + (uint32_t)u) ? m : TAv;
                            }
                        }
                        // entries past the end of the queue (clamped loads above): only the queue's last chunk has any
                        const uint32_t tail = sCur & 63u;
                        if (tail && nch_all - 1u - c0 < 64u && lane == nch_all - 1u - c0) TAv &= (1ull << tail) - 1ull;
                        if (lane >= nch) TAv = 0;
                        blk0 = c0;
                    };
#ifdef DEC_PROF
                    { const uint64_t tb = __builtin_amdgcn_s_memtime(); build_block(0); pf[1] += __builtin_amdgcn_s_memtime() - tb; pf[10] += 1; }
#else
                    build_block(0);
#endif
#ifdef DEC_PAD  // set by the Makefile (tools/hop_align.py): s_nop padding, run once per phase, that puts the hop loop
                // at the offset in a 32-byte fetch block that was measured best (4 % between the best and the worst offset)
#define DEC_STR2(x) #x
#define DEC_STR(x) DEC_STR2(x)
                    asm volatile(".p2align 6\n\t.fill " DEC_STR(DEC_PAD) ", 4, 0xBF800000");
#endif
                    if (lane == 0) {  // before the phase's first slot (same wave: LDS writes stay in order)
                        sh.ph.seq0 = sSeq; sh.ph.widx0 = widx0s; sh.ph.b_lsp = sLsp; sh.ph.b_lip = sLip; sh.ph.b_ret = sRet;
                    }
                    const bool ran = i < sCur && sP < sNb;
                    if (ran) for (;;) {
                        const uint32_t widx = sP >> 6, Wb = widx << 6, pos0 = sP & 63u;
                        uint32_t pos = pos0;
#ifdef DEC_PROF
                        const uint64_t tw = __builtin_amdgcn_s_memtime();
#endif
                        // the helper wavefront has prepared this window: its bits and, per position, the length a
                        // fired type-A entry would have there
                        const uint32_t kw = widx - widx0s;
                        if (kw >= pknown) {
                            uint32_t spins = 0;
                            while ((pknown = RFL(lds_load(&sh.pprog))) <= kw) {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > SPIN_LIMIT) { sh.bad = 2; break; }
                            }
                        }
                        const uint32_t pslot = kw % DEC_PREP;
                        const uint32_t vb = (sNb - Wb) < 64u ? (sNb - Wb) : 64u;
                        const uint32_t i0 = i;
                        uint64_t fm = 0;
                        bool by_table = false;
#ifdef DEC_PROF
                        const uint64_t th = __builtin_amdgcn_s_memtime();
                        pf[12] += th - tw;
#endif
                        // Entries [i, i+64) can be all this window meets (an entry takes at least one bit): their type
                        // mask is a funnel shift of the masks of chunk i/64 and the next one, both kept in TAv.
                        uint32_t rel = 0;
                        {
                            const uint32_t ch = i >> 6, r0 = i & 63u;
#ifdef DEC_PROF
                            if (ch + 1 >= blk0 + 64) { const uint64_t tb = __builtin_amdgcn_s_memtime(); build_block(ch); pf[1] += __builtin_amdgcn_s_memtime() - tb; pf[10] += 1; }
#else
                            if (ch + 1 >= blk0 + 64) build_block(ch);
#endif
                            const uint64_t TAc = readlane64(TAv, ch - blk0), TAn = readlane64(TAv, ch + 1 - blk0);
                            uint64_t Tr = r0 ? ((TAc >> r0) | (TAn << (64u - r0))) : TAc;
                            asm volatile("" ::: "memory");  // keep the LDS reads issued above, their first use below
                            // The window is whole and starts at most eight bits in (a token is nine bits at most: true of every
                            // window but a phase's first): the helper has walked it for that entry point already under the
                            // hypothesis "every entry is a type-A entry with offspring" (DecShared::tab).  That walk is THIS
                            // window's walk if the entries it met -- `cnt` of them, 19 on average at 1080p, not the 64 a window
                            // can reach at most -- are all of that type: one look-up instead of a hop per fired entry, for 72 %
                            // of the windows with 84 % of the fired entries at 1080p (profiles/r04_lis_type_runs.txt).
                            const bool tab_try = (Tr & 1ull) != 0 && pos0 <= 8u && vb == 64u;
                            if (tab_try && kw >= tknown) tknown = RFL(lds_load(&sh.tprog));
                            PF_CNT(18, tab_try ? 1 : 0);
                            PF_CNT(19, tab_try && kw >= tknown ? 1 : 0);
                            PF_CNT(21, (Tr & 1ull) ? 0 : 1);
                            PF_CNT(22, (Tr & 1ull) != 0 && !(pos0 <= 8u && vb == 64u) ? 1 : 0);
                            if (tab_try && kw < tknown) {
                                const uint32_t tl = lane < 9u ? lane : 0u;
                                const uint64_t tfm = sh.tab[pslot].fm[tl];
                                const uint32_t tce = sh.tab[pslot].ce[tl];
                                const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)tce, (int)pos0);
                                const uint32_t cnt = ce & 0x7Fu;
                                const uint32_t na = Tr == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~Tr);  // type-A entries ahead
                                if (cnt <= na) {  // the table's walk met type-A entries only: it is this window's walk
                                    fm = readlane64(tfm, pos0);
                                    rel = cnt;
                                    pos = 64u + (ce >> 7);
                                    i += rel;
                                    by_table = true;
                                    PF_CNT(5, 1);
                                } else PF_CNT(20, 1);
                            }
                            if (!by_table) {
                            // the window's bits and, one per lane, the length a fired type-A entry would have at that position
                            // (DecShared::pln: three bit planes of length - 5)
                            const uint32_t *prow = reinterpret_cast<const uint32_t *>(sh.pln[pslot]) + (lane >> 5);  // the lane's half
                            const uint32_t pe0 = prow[0], pe1 = prow[2], pe2 = prow[4], sl = lane & 31u;
                            const uint64_t plo = sh.pln[pslot][3];
                            const uint64_t lo = ((uint64_t)RFL((uint32_t)(plo >> 32)) << 32) | RFL((uint32_t)plo);
                            const uint32_t LAv = 5u + ((pe0 >> sl) & 1u) + 2u * ((pe1 >> sl) & 1u) + 4u * ((pe2 >> sl) & 1u);
                            uint64_t Lr = lo >> pos, c64;
                            uint32_t f, dd, len;
                            // The walk, hand-scheduled: this serial chain bounds the whole decoder and hipcc's version of
                            // it is more than twice as long (uniform conditions routed through VCC/EXEC).  Relative
                            // coordinates keep the dependent chain short: L = bits from the current position on, T = type
                            // mask from the current entry on, so a candidate is `L & T` (s_and sets SCC), its offset
                            // d = ff1, and one hop is  T >>= d+1,  L >>= d+len  (L in two shifts: a single shift count must
                            // stay below 64; T in one: d+1 = 64 shifts by 0 instead, but then L, which had its only candidate
                            // at bit 63, is 0 after its own shifts, the walk ends and T is not looked at again).  Chain per hop: and -> ff1 -> add -> readlane -> lshr -> and, about 100
                            // cycles (tools/ubench/hop.hip); four hops per trip save three of four taken branches.
                            // Four SALU instructions separate the s_add that makes the lane select from v_readlane.
#define HOP_BODY                                                \
    "s_ff1_i32_b64 %[d], %[c]\n\t"                              \
    "s_add_i32 %[f], %[pos], %[d]\n\t"                          \
    "s_lshr_b64 %[L], %[L], %[d]\n\t"                           \
    "s_add_i32 %[d], %[d], 1\n\t"                               \
    "s_bitset1_b64 %[fm], %[f]\n\t"                             \
    "s_lshr_b64 %[T], %[T], %[d]\n\t"                           \
    "v_readlane_b32 %[len], %[LAv], %[f]\n\t"                   \
    "s_add_i32 %[rel], %[rel], %[d]\n\t"                        \
    "s_lshr_b64 %[L], %[L], %[len]\n\t"                         \
    "s_add_i32 %[pos], %[f], %[len]\n\t"                        \
    "s_and_b64 %[c], %[L], %[T]\n\t"
                            asm volatile(
                                "s_and_b64 %[c], %[L], %[T]\n\t"
                                "s_cbranch_scc0 s_hop_done%=\n"
                                "s_hop_loop%=:\n\t"
                                HOP_BODY
                                "s_cbranch_scc0 s_hop_done%=\n\t"
                                HOP_BODY
                                "s_cbranch_scc0 s_hop_done%=\n\t"
                                HOP_BODY
                                "s_cbranch_scc0 s_hop_done%=\n\t"
                                HOP_BODY
                                "s_cbranch_scc1 s_hop_loop%=\n"
                                "s_hop_done%=:\n\t"
                                : [rel] "+s"(rel), [pos] "+s"(pos), [fm] "+s"(fm), [L] "+s"(Lr), [T] "+s"(Tr), [c] "=&s"(c64),
                                  [f] "=&s"(f), [d] "=&s"(dd), [len] "=&s"(len)
                                : [LAv] "v"(LAv)
                                : "scc");
#undef HOP_BODY
                            // no fired type-A entry left among the window's bits: the rest take one bit each
                            {
                                const int32_t zb = (int32_t)vb - (int32_t)pos, ze = (int32_t)(sCur - i - rel);  // ze >= 0
                                const int32_t zm = zb < ze ? zb : ze;
                                const uint32_t z = (uint32_t)(zm > 0 ? zm : 0);
                                rel += z;
                                pos += z;
                            }
                            i += rel;
                            }
                        }
#ifdef DEC_PROF
                        pf[13] += __builtin_amdgcn_s_memtime() - th;
#endif
#ifdef DEC_PROF
                        const uint64_t tq = __builtin_amdgcn_s_memtime();
#endif
#ifdef DEC_PROF
                        seq_publish(sh, sSeq, sFc, fm, i0, pos0, pos < 64u ? pos : 64u, lane, kw, &pf[16]);
#else
                        seq_publish(sh, sSeq, sFc, fm, i0, pos0, pos < 64u ? pos : 64u, lane, kw);
#endif
                        asm volatile("s_add_i32 %0, %0, 1" : "+s"(sSeq) : : "scc");  // (asm: keeps it in an SGPR)
#ifdef DEC_PROF
                        pf[11] += __builtin_amdgcn_s_memtime() - tq;
#endif
                        sP = Wb + pos;
                        // A RUN of such windows.  What a window costs the walk is by now less its hops than what surrounds them --
                        // the type mask of the 64 entries ahead (four cross-lane reads, a funnel shift), the window's row of
                        // token lengths, the helper's progress -- and inside a stretch of the queue that is all type A none of
                        // that is needed: the entries ahead of type A are counted once (`ones`, from the chunk masks in TAv),
                        // and while the next window's own entries (`cnt`, out of its table row) are no more than are left of
                        // them, a window is its table row and its slot.
                        if (by_table) {
                            uint32_t ones = 0;
                            {
                                const uint32_t c2 = i >> 6, r2 = i & 63u;
                                if (c2 >= blk0 && c2 < blk0 + 64u) {
                                    const uint64_t c0 = readlane64(TAv, c2 - blk0) >> r2;
                                    const bool all0 = c0 == (~0ull >> r2);
                                    ones = all0 ? 64u - r2 : (uint32_t)__builtin_ctzll(~c0);
                                    const uint32_t L1 = c2 - blk0 + 1u;
                                    if (all0 && L1 < 64u) {
                                        const uint64_t full = __ballot(TAv == ~0ull) >> L1;  // the chunks behind, all type A?
                                        const uint32_t nf = (uint32_t)__builtin_ctzll(~full);   // (<= 64 - L1: the shift brought zeros in)
                                        ones += 64u * nf;
                                        if (L1 + nf < 64u) ones += (uint32_t)__builtin_ctzll(~readlane64(TAv, L1 + nf));  // (not all ones)
                                    }
                                }
                            }
                            uint32_t rp0 = pos - 64u, rkw = kw + 1u, rWb = Wb + 64u;
#ifdef DEC_PROF
                            const uint64_t trl = __builtin_amdgcn_s_memtime();
#endif
                            // The loop itself is written out (the compiler's version of it is 55 instructions with nine
                            // branches, and the sequencer retires an instruction per ten cycles whatever it is): per window
                            // the row of the table is read a window ahead -- which row does not depend on the walk, only which
                            // of its nine entries is taken --, the entry's count is held against the type-A entries left, the
                            // slot is written by lane 0 (three LDS writes, the flag last: one wavefront's LDS writes stay in
                            // order) and the walk's four numbers move on: 36 instructions, one branch, eight vector registers.  It runs for as many
                            // windows as need no look at anything else: up to the next ring-space check, the tables known
                            // to be there, the whole windows left of the stream.
                            static_assert(DEC_PREP == 64 && DEC_RING == 64 && sizeof(Slot) == 32, "constants of the loop below");
                            const uint32_t tl = lane < 9u ? lane : 0u;
                            const uint32_t a_tab = (uint32_t)(uintptr_t)&sh.tab[0], a_ring = (uint32_t)(uintptr_t)&sh.ring[0];
                            const uint32_t vbfm = a_tab + tl * 8u, vbce = a_tab + 72u + tl * 2u;
                            for (;;) {
                                if (rWb + 64u > sNb) break;
                                if (rkw >= tknown) {
                                    tknown = RFL(lds_load(&sh.tprog));
                                    if (rkw >= tknown) { PF_CNT(15, 1); break; }  // (the helper is not there yet: the walk above hops)
                                }
#ifdef DEC_PROF
                                if (sFc == 0) seq_ring_check(sh, sSeq, sFc, lane, rkw, &pf[16]);
#else
                                if (sFc == 0) seq_ring_check(sh, sSeq, sFc, lane, rkw, nullptr);
#endif
                                uint32_t n = sFc;
                                n = n < tknown - rkw ? n : tknown - rkw;
                                n = n < ((sNb - rWb) >> 6) ? n : ((sNb - rWb) >> 6);
                                const uint32_t n0 = n;
                                uint32_t row = RFL((rkw % DEC_PREP) * (uint32_t)sizeof(TabRow));
                                n = RFL(n); rp0 = RFL(rp0); i = RFL(i); ones = RFL(ones);  // (wave-uniform all: tells the compiler so)
                                uint32_t va1, va2, vlo, vhi, vce, vs, d0, d1, ce, f0, f1, cnt;
                                asm volatile(
                                    "v_add_u32 %[va1], %[row], %[vbfm]\n\t"
                                    "v_add_u32 %[va2], %[row], %[vbce]\n\t"
                                    "ds_read_b32 %[vlo], %[va1]\n\t"
                                    "ds_read_b32 %[vhi], %[va1] offset:4\n\t"
                                    "ds_read_u16 %[vce], %[va2]\n\t"
                                    "s_waitcnt lgkmcnt(0)\n"
                                    "s_runl%=:\n\t"
                                    "v_readlane_b32 %[ce], %[vce], %[rp0]\n\t"
                                    "v_readlane_b32 %[f0], %[vlo], %[rp0]\n\t"
                                    "v_readlane_b32 %[f1], %[vhi], %[rp0]\n\t"
                                    "s_add_u32 %[row], %[row], 96\n\t"
                                    "s_cmp_eq_u32 %[row], 6144\n\t"
                                    "s_cselect_b32 %[row], 0, %[row]\n\t"
                                    "v_add_u32 %[va1], %[row], %[vbfm]\n\t"
                                    "v_add_u32 %[va2], %[row], %[vbce]\n\t"
                                    "s_and_b32 %[cnt], %[ce], 0x7f\n\t"
                                    "ds_read_b32 %[vlo], %[va1]\n\t"
                                    "ds_read_b32 %[vhi], %[va1] offset:4\n\t"
                                    "ds_read_u16 %[vce], %[va2]\n\t"
                                    "s_cmp_gt_u32 %[cnt], %[ones]\n\t"
                                    "s_cbranch_scc1 s_rund%=\n\t"
                                    "s_mov_b64 exec, 1\n\t"
                                    "v_mov_b32 %[d0], %[f0]\n\t"
                                    "v_mov_b32 %[d1], %[f1]\n\t"
                                    "s_and_b32 %[f0], %[seq], 63\n\t"
                                    "s_lshl_b32 %[f0], %[f0], 5\n\t"
                                    "v_add_u32 %[vs], %[f0], %[vring]\n\t"
                                    "s_or_b32 %[f1], %[rp0], 0x4000\n\t"
                                    "ds_write2_b32 %[vs], %[d0], %[d1] offset1:1\n\t"
                                    "v_mov_b32 %[d0], %[i]\n\t"
                                    "v_mov_b32 %[d1], %[f1]\n\t"
                                    "s_add_u32 %[seq], %[seq], 1\n\t"
                                    "ds_write2_b32 %[vs], %[d0], %[d1] offset0:2 offset1:3\n\t"
                                    "v_mov_b32 %[d0], %[seq]\n\t"
                                    "s_lshr_b32 %[rp0], %[ce], 7\n\t"
                                    "ds_write_b32 %[vs], %[d0] offset:16\n\t"
                                    "s_mov_b64 exec, -1\n\t"
                                    "s_add_u32 %[i], %[i], %[cnt]\n\t"
                                    "s_sub_u32 %[ones], %[ones], %[cnt]\n\t"
                                    "s_sub_u32 %[n], %[n], 1\n\t"
                                    "s_cmp_lg_u32 %[n], 0\n\t"
                                    "s_waitcnt lgkmcnt(3)\n\t"
                                    "s_cbranch_scc1 s_runl%=\n"
                                    "s_rund%=:\n\t"
                                    "s_waitcnt lgkmcnt(0)\n\t"
                                    : [rp0] "+s"(rp0), [i] "+s"(i), [ones] "+s"(ones), [seq] "+s"(sSeq), [n] "+s"(n), [row] "+s"(row),
                                      [va1] "=&v"(va1), [va2] "=&v"(va2), [vlo] "=&v"(vlo), [vhi] "=&v"(vhi), [vce] "=&v"(vce),
                                      [vs] "=&v"(vs), [d0] "=&v"(d0), [d1] "=&v"(d1),
                                      [ce] "=&s"(ce), [f0] "=&s"(f0), [f1] "=&s"(f1), [cnt] "=&s"(cnt)
                                    : [vbfm] "v"(vbfm), [vbce] "v"(vbce), [vring] "v"(a_ring)
                                    : "scc", "memory");
                                const uint32_t took = n0 - n;
                                sFc -= took;
                                rkw += took;
                                rWb += 64u * took;
                                PF_CNT(5, took);
                                PF_CNT(17, took);
                                if (n != 0) break;  // the next window reaches past the type-A stretch: the walk above takes it
                            }
#ifdef DEC_PROF
                            pf[14] += __builtin_amdgcn_s_memtime() - trl;
#endif
                            sP = rWb + rp0;
                        }
                        // exit test, once per window: queue exhausted (i == sCur) or stream exhausted (sP >= sNb); one sign
                        // test (all quantities are below 2^31) instead of two compare / select pairs
                        if ((int32_t)((sCur - 1u - i) | (sNb - 1u - sP)) < 0) break;
                    }
                    // the last entry's child bits ran past the end of the stream (a phase that starts at or past the end
                    // leaves its first entry waiting, like one that reaches the end exactly)
                    const bool over = ran && sP > sNb;
                    if (over) dn = 1;
                    else if (i < sCur) {  // stream exhausted before the queue: cur[i] is the entry left waiting for a bit
                        if (META && lane == 0) tr_put(tr, sNb, (cur[i] & ENT_A) ? 2u : 5u, n, cur[i]);
                        dn = 1;
                    }
                    P = sP;
                    seq = sSeq;
#undef RFL
                    seq_close(sh, par, seq, P, dn, 0, 0, lane);
                } else if (wave == 1) {
                    helper_phase(sh, bs, par, P >> 6, lane);
                } else if (DEC_IS_WORKER(wave)) {
#ifdef DEC_PROF
                    worker_phase<META>(sh, a, g, myk, par, nullptr, nullptr, lip, cur, cur_len, nxt, ret, lsp_idx, lsp_val, nbits,
                                       n, base_val, DEC_WK(wave), lane, tr, bs, wpf);
#else
                    worker_phase<META>(sh, a, g, myk, par, nullptr, nullptr, lip, cur, cur_len, nxt, ret, lsp_idx, lsp_val, nbits,
                                       n, base_val, DEC_WK(wave), lane, tr, bs);
#endif
                }
                PF_ADD(2);
                __syncthreads();
                if (threadIdx.x == 0) { sh.pprog = 0; sh.sprog = 0; sh.tprog = 0; }  // between the two barriers that end the phase
                phase++;
                PF_CNT(8, 1);
                const uint32_t seq_end = sh.head;
                PF_CNT(6, seq_end - seq0);
                uint32_t nxt_len = 0;
                if (seq_end > seq0) {  // running lengths after the generation's last window
                    const Chain *lc = &sh.chain[(seq_end - 1) % DEC_RING];
                    lsp_len = lc->lsp; lip_len = lc->lip; nxt_len = lc->nxt; ret_len = lc->ret;
                }
                seq = seq_end;
                P = sh.r_P;
                if (sh.r_done || sh.bad) done = true;
                __syncthreads();
                PF_ADD(3);
                { uint32_t *t = cur; cur = nxt; nxt = t; }
                cur_len = nxt_len;
            }
            if (done) break;
            lis = ret; lis_len = ret_len;
            qa = cur; qb = nxt;

            // ================= refinement (encoder_decoder.rs:438-444): all wavefronts =================
            {
                const uint32_t left = nbits > P ? nbits - P : 0u;
                const uint32_t count = lsp_len0 < left ? lsp_len0 : left;
                for (uint32_t t0 = wave * 256u; t0 < count; t0 += DEC_NW * 256u) {
                    int32_t v[4];
                    uint32_t bit[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t t = t0 + (uint32_t)u * 64 + lane;
                        v[u] = t < count ? lsp_val[t] : 0;
                        const uint32_t pos = P + t;
                        bit[u] = t < count ? ((bs.gw[pos >> 5] >> (pos & 31)) & 1u) : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t t = t0 + (uint32_t)u * 64 + lane;
                        const bool in = t < count;
                        if (in) lsp_val[t] = set_bit_i32(v[u], (uint32_t)n, bit[u]);
                        if (META && in) tr_put(tr, P + t, 6, n, lsp_idx[t]);  // action 6 (:822)
                    }
                }
                if (META && count < lsp_len0 && threadIdx.x == 0) tr_put(tr, nbits, 6, n, lsp_idx[count]);  // waiting
                P += count;
                ref_plane = (uint32_t)n;
                ref_count = count;
                if (count < lsp_len0) done = true;
                __syncthreads();
                PF_ADD(4);
            }
            if (n == 0) break;
        }

        // ---------------- scatter decoded values ----------------
        __syncthreads();
        PF_ADD(7);
        // With an odd ll_h or ll_w some cells are owned by two or three list entries (dup_cell); those are left out here
        // and resolved below.
        const bool dups = ((g.ll_h | g.ll_w) & 1) != 0;
        // On the capacity-error path (sh.bad: reported below, the output is void) the length kept counting past the
        // list's end: what follows must stay inside this slot's LSP -- and inside its LIP buffers, which resolve_dups
        // uses as mail words indexed by LSP position (caps.lsp <= caps.lip whenever no error is pending).
        lsp_len = min(lsp_len, min(a.caps.lsp, a.caps.lip));
        // Four entries per thread per batch, and the loads of the next batch are issued BEFORE the stores of this one:
        // loads and stores retire through one in-order counter (vmcnt), so a load issued after a scattered store
        // cannot be waited for without waiting for that store to be acknowledged (a plain loop pays load latency +
        // store latency per batch; the stores may alias the lists as far as the compiler knows, so it keeps this order).
        // Two register sets, alternating, so that no register move has to wait for the newest loads.
        {
            constexpr uint32_t STR = DEC_NW * 64, U = 4;
            const uint32_t t_end = lsp_len;
            int32_t va[U], vb[U];
            uint32_t ia[U], ib[U];
            auto ld = [&](uint32_t t0, int32_t(&v)[U], uint32_t(&ix)[U]) {
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {  // unconditional loads (clamped index): no branches between them
                    const uint32_t t = t0 + u * STR, tc = t < t_end ? t : t_end - 1u;
                    const int32_t lv = lsp_val[tc];
                    ix[u] = lsp_idx[tc];
                    v[u] = t < t_end ? lv : 0;  // a decoded value is never 0
                }
            };
            const bool mark = !META && a.fl.p != nullptr;
            const uint32_t plane0 = (uint32_t)b * (uint32_t)g.c;
            // LSP positions of duplicated cells go to a list of their own (the LIS buffers are free by now)
            auto st = [&](uint32_t t0, const int32_t(&v)[U], const uint32_t(&ix)[U]) {
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    if (v[u]) {
                        if (dups && dup_cell(g, ix[u] & IDXM)) {
                            const uint32_t q = atomicAdd(&sh.ndup, 1u);
                            if (q < a.caps.lis) q0[q] = t0 + u * STR;
                        } else {
                            out[ix[u] & IDXM] = v[u];
                        }
                        // (every entry with a value, those of duplicated cells included: a set word only means "read")
                        if (mark) l1_mark(g, a.fl, plane0, ix[u] & IDXM);
                    }
                }
            };
            if (t_end > 0) {
                uint32_t t0 = threadIdx.x;
                ld(t0, va, ia);
                for (; t0 < t_end; t0 += 2 * STR * U) {
                    ld(t0 + STR * U, vb, ib);
                    st(t0, va, ia);
                    ld(t0 + 2 * STR * U, va, ia);
                    st(t0 + STR * U, vb, ib);
                }
            }
        }
        if (dups) {
            __syncthreads();  // the list and its length are complete
            const uint32_t nd = sh.ndup;
            resolve_dups<IDXM>(g, out, lsp_idx, lsp_val, lsp_len, lipA, lipB, ref_plane, ref_count, nd <= a.caps.lis ? q0 : nullptr, nd);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (a.lsp_count) a.lsp_count[slot] = lsp_len;
            uint32_t ecode = 0;
            if (bad) ecode |= 1u | 0x100u;
            if (sh.bad == 1) ecode |= 1u | 0x200u;            // list capacity
            if (sh.bad == 2) ecode |= 1u | 0x400u;            // spin limit
            if (ecode) atomicOr(a.err, ecode);
#ifdef DEC_PROF
            PF_ADD(9);
            if (b == 0) for (int q = 0; q < 24; q++) a.err[16 + q] = (uint32_t)(q == 5 || q == 6 || q == 8 || q == 10 || q == 15 || q >= 17 ? pf[q] : (pf[q] >> 10));
            }
        if (b == 0 && threadIdx.x == 128) for (int q = 0; q < 5; q++) a.err[40 + q] = (uint32_t)(q == 4 ? wpf[q] : (wpf[q] >> 10));
        if (threadIdx.x == 0) {
            if (b < 448) {  // where and when this workgroup ran (tools/prof_decode.py: spread over the CUs, stragglers)
                uint32_t *r = a.err + 64 + 4 * b;
                r[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
                r[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
                r[2] = (uint32_t)pf_rt0;
                r[3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            }
#endif
        }
        __syncthreads();
    }
}

}  // namespace DEC_NS

#ifndef DEC_VARIANT
// Puts back the zeros: clears exactly the cells the decoder wrote (its LSP lists are still in the slot scratch), so a
// coefficient array that is only ever used as decoder output and inverse-transform input never needs a full zero-fill
// again.  One image per slot (B <= nslots).
#ifndef UNSC_BLOCKS
#define UNSC_BLOCKS 8
#endif
__global__ __launch_bounds__(256) void k_unscatter(DecArgs a) {
    const uint32_t slot = blockIdx.x / UNSC_BLOCKS, part = blockIdx.x % UNSC_BLOCKS;
    const uint32_t cnt = a.lsp_count[slot];
    const uint32_t *idx = a.lsp_idx + (size_t)slot * a.caps.lsp;
    int32_t *out = a.out + (size_t)slot * a.g.n;
    // eight loads in flight per thread, then the eight stores (a plain loop waits for memory once per entry: the
    // store may alias the list as far as the compiler knows)
    for (uint32_t t0 = part * 256u + threadIdx.x; t0 < cnt; t0 += UNSC_BLOCKS * 256u * 8u) {
        uint32_t ix[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t t = t0 + u * UNSC_BLOCKS * 256u;
            ix[u] = t < cnt ? idx[t] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (ix[u] != 0xFFFFFFFFu) out[ix[u] & ENT_IDX] = 0;
    }
}

extern "C" int spiht_launch_unscatter(const DecArgs *a, hipStream_t st) {
    if (a->B < 1) return 0;
    hipLaunchKernelGGL(k_unscatter, dim3(a->B * UNSC_BLOCKS), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}

#endif  // !DEC_VARIANT

extern "C" int DEC_LAUNCH(const DecArgs *a, hipStream_t st) {
    int grid = a->nslots < a->B ? a->nslots : a->B;
    if (grid < 1) return 0;
    if (a->tr_ent)
        hipLaunchKernelGGL(DEC_NS::k_decode<true>, dim3(grid), dim3(DEC_NW * 64), 0, st, *a);
    else
        hipLaunchKernelGGL(DEC_NS::k_decode<false>, dim3(grid), dim3(DEC_NW * 64), 0, st, *a);
    return (int)hipGetLastError();
}
