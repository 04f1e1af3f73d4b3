// SPIHT list decoder (gfx950): one wavefront owns one image at a time.
//
// Reproduces /root/reference/src/encoder_decoder.rs:307-454 on all 8*nbytes bits of the stream
// (src/lib.rs:38 hands the pad bits to the decoder as data).  Unlike the encoder, the position of a
// list entry's bits depends on every bit decoded before it, so the three passes are attacked
// differently:
//   * LIP pass: tokens are '0' | '1 s'.  For a 64-bit window the token-start mask is computed in O(1)
//     with the carry trick used for escaped characters in SIMD JSON parsers (runs of ones pair up from
//     their first bit); lane l then owns stream position l, its token rank is a popcount, and LIP
//     reads / LSP+LIP writes are coalesced.
//   * LIS pass: per generation, windows of 64 entries.  Unfired entries and fired B entries take one
//     bit; only a fired A entry (1 + 4..8 bits) shifts what follows.  Lane l precomputes the length a
//     fired A entry would have at stream position l; a scalar walk then hops from fired A to fired A
//     (find-first-set on `bits & type-mask`), assigning every entry its position.  Outputs (next
//     generation, retained list, LIP/LSP appends) are then produced in parallel with one packed
//     wave scan.
//   * refinement: bit t belongs to LSP entry t.
// Decoded magnitudes live next to the LSP (lsp_val) and are scattered into the coefficient array
// at the end; the few operations that consumed one of the last 8 bits of the stream (possible pad
// bits, Q9) are replayed serially in list order so duplicated tree nodes (Q4) end exactly as the
// reference's sequential writes leave them.
#include "common.h"

#define DEC_CH 2048  // 32-bit words of stream staged in LDS per wave
#define DEC_TAIL 16
#define POS_INVALID 0xFFFFFFFFu

struct TailOp {
    uint32_t idx;
    int32_t val;   // value for a write, bit for a refine
    uint32_t n;    // plane
    uint32_t kind; // 0 = write, 1 = refine
};

#define DEC_LIST 1024  // list entries staged in LDS per refill

struct DecShared {
    uint32_t w[DEC_CH + 8];
    uint32_t lst[DEC_LIST];
    uint32_t seg[64];
    TailOp tail[DEC_TAIL];
};

__device__ __forceinline__ uint64_t lt_mask(uint32_t lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int o) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_up((int)lo, o);
    hi = (uint32_t)__shfl_up((int)hi, o);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_exscan(uint64_t v, uint64_t &total, uint32_t lane) {
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = shfl_up_u64(inc, o);
        if (lane >= (uint32_t)o) inc += t;
    }
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)inc, 63);
    uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(inc >> 32), 63);
    total = ((uint64_t)hi << 32) | lo;
    return inc - v;
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

struct BitSrc {
    const uint32_t *gw;   // stream words of this image
    uint32_t nwords;      // words readable in the slot
    uint32_t nbits;       // valid bits (8*nbytes)
    uint32_t cb;          // first word staged in LDS
};

// stage words [wbase, wbase+DEC_CH+8) with everything at or past nbits forced to zero
__device__ __forceinline__ void refill(DecShared &sh, BitSrc &bs, uint32_t wbase, uint32_t lane) {
    __syncthreads();
    bs.cb = wbase;
    for (uint32_t t = lane; t < DEC_CH + 8; t += 64) {
        uint32_t wi = wbase + t;
        uint32_t v = 0;
        uint64_t b0 = (uint64_t)wi * 32;
        if (wi < bs.nwords && b0 < bs.nbits) {
            v = bs.gw[wi];
            uint32_t rem = bs.nbits - (uint32_t)b0;
            if (rem < 32) v &= (1u << rem) - 1u;
        }
        sh.w[t] = v;
    }
    __syncthreads();
}

// make sure bits [P, P+need) (+64 of slack for peek64) are staged; P never decreases
__device__ __forceinline__ void ensure(DecShared &sh, BitSrc &bs, uint32_t P, uint32_t need, uint32_t lane) {
    uint32_t lastw = (uint32_t)(((uint64_t)P + need + 63) >> 5) + 2;
    if (lastw >= bs.cb + DEC_CH + 8 || (P >> 5) < bs.cb) refill(sh, bs, P >> 5, lane);
}

__device__ __forceinline__ uint64_t peek64(const DecShared &sh, const BitSrc &bs, uint32_t pos) {
    uint32_t w = (pos >> 5) - bs.cb, s = pos & 31;
    uint64_t lo = (uint64_t)sh.w[w] | ((uint64_t)sh.w[w + 1] << 32);
    uint32_t hi = sh.w[w + 2];
    return s ? ((lo >> s) | ((uint64_t)hi << (64 - s))) : lo;
}

// Token-start mask of a LIP-pass window (tokens '0' | '1 s').  The pass owns bits >= pos of the window.
// cin: bit `pos` is the pending sign bit of the previous window's last token.  cout: the token starting at
// bit 63 is '1' and its sign bit is the next window's bit 0.  Runs of ones pair up from their first bit, so
// the sign positions are the odd offsets inside a run plus the zero that follows an odd-length run; the
// run-parity is found with the add-carry trick used for escaped characters in SIMD JSON parsers.
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

// popcount of the bits of `m` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

// 64 consecutive 64-bit stream words, one per lane, so that the serial passes fetch their windows with
// v_readlane instead of a memory access
struct RegChunk {
    uint64_t v;
    uint32_t base64;  // index of lane 0's 64-bit word
    uint32_t valid;
};

// sequential reader of a list in global memory through an LDS stage of DEC_LIST entries
struct ListRd {
    const uint32_t *src;
    uint32_t len, lo, hi;  // entries [lo,hi) are staged
};

__device__ __forceinline__ void regchunk_load(DecShared &sh, BitSrc &bs, RegChunk &rc, uint32_t base64, uint32_t lane) {
    // words [2*base64, 2*base64+128) must be staged
    uint32_t wfirst = 2 * base64;
    if (wfirst < bs.cb || wfirst + 130 > bs.cb + DEC_CH + 8) refill(sh, bs, wfirst, lane);
    uint32_t o = wfirst - bs.cb + 2 * lane;
    rc.v = (uint64_t)sh.w[o] | ((uint64_t)sh.w[o + 1] << 32);
    rc.base64 = base64;
    rc.valid = 1;
}

// (lo, hi) = 64-bit stream words widx, widx+1
__device__ __forceinline__ void window(DecShared &sh, BitSrc &bs, RegChunk &rc, uint32_t widx, uint32_t lane,
                                       uint64_t &lo, uint64_t &hi) {
    if (!rc.valid || widx < rc.base64 || widx + 1 >= rc.base64 + 64) regchunk_load(sh, bs, rc, widx, lane);
    const uint32_t k = widx - rc.base64;
    lo = readlane64(rc.v, k);
    hi = readlane64(rc.v, k + 1);
}

__device__ __forceinline__ void list_stage(DecShared &sh, ListRd &r, uint32_t from, uint32_t lane) {
    __syncthreads();
    r.lo = from;
    r.hi = (r.len - from) < (uint32_t)DEC_LIST ? r.len : from + DEC_LIST;
    uint32_t v[DEC_LIST / 64];
#pragma unroll
    for (int u = 0; u < DEC_LIST / 64; u++) {
        uint32_t t = from + (uint32_t)u * 64 + lane;
        v[u] = t < r.hi ? r.src[t] : 0u;
    }
#pragma unroll
    for (int u = 0; u < DEC_LIST / 64; u++) sh.lst[u * 64 + lane] = v[u];
    __syncthreads();
}
// make entries [a, min(a+64,len)) available in sh.lst at offset a - r.lo
__device__ __forceinline__ void list_need(DecShared &sh, ListRd &r, uint32_t a, uint32_t lane) {
    uint32_t b = (r.len - a) < 64u ? r.len : a + 64;
    if (a < r.lo || b > r.hi) list_stage(sh, r, a, lane);
}

__global__ __launch_bounds__(64) void k_decode(DecArgs a) {
    __shared__ DecShared sh;
    const Geom g = a.g;
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x;
    const uint32_t W = (uint32_t)g.w, H = (uint32_t)g.h;

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
        const uint32_t tail_start = nbits >= 8 ? nbits - 8 : 0;
        int n = (int)a.max_n[b];
        if (n > 30) { bad = true; n = 0; }
        refill(sh, bs, 0, lane);
        RegChunk rc;
        rc.v = 0; rc.base64 = 0; rc.valid = 0;

        uint32_t *lip = lipA, *lipn = lipB;
        uint32_t *lis = q0, *qa = q1, *qb = q2;
        uint32_t lip_len = 0, lsp_len = 0, lis_len = 0;
        uint32_t P = 0, cut = 0, ntail = 0;

        // ---- initial LIP / LIS (encoder_decoder.rs:327-348) ----
        const uint32_t nroot = (uint32_t)(g.ll_h * g.ll_w * g.c);
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
            uint32_t o = lis_len + mbcnt(m);
            if (inlis && o < a.caps.lis) lis[o] = idx | ENT_A;
            lis_len += (uint32_t)__popcll(m);
        }
        lip_len = nroot;
        if (lip_len > a.caps.lip || lis_len > a.caps.lis) bad = true;
        __syncthreads();

        bool done = bad;
        for (; !done; --n) {
            const uint32_t lsp_len0 = lsp_len;
            const int32_t base_val = (n == 0) ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));  // :364-370

            // ---------------- LIP pass (encoder_decoder.rs:355-377) ----------------
            {
                ListRd rd;
                rd.src = lip; rd.len = lip_len; rd.lo = 0; rd.hi = 0;
                uint32_t m_rem = lip_len, tok_base = 0, lipn_len = 0, cin = 0;
                while (m_rem > 0 && !done) {
                    if (P >= nbits) { done = true; break; }
                    const uint32_t widx = P >> 6, pos = P & 63u, Wb = widx << 6;
                    uint64_t Wd, Wn;
                    window(sh, bs, rc, widx, lane, Wd, Wn);
                    const uint32_t nxtbit = (uint32_t)Wn & 1u;
                    const uint32_t vb = (nbits - Wb) < 64u ? (nbits - Wb) : 64u;
                    uint32_t cout;
                    uint64_t S = lip_starts(Wd, pos, cin, cout);
                    if (vb < 64) S &= (1ull << vb) - 1ull;
                    const uint32_t cnt = (uint32_t)__popcll(S);
                    list_need(sh, rd, tok_base, lane);
                    const bool isS = (S >> lane) & 1ull;
                    const uint32_t rank = mbcnt(S);
                    const bool inpass = isS && rank < m_rem;
                    const uint32_t sig = (uint32_t)(Wd >> lane) & 1u;
                    const bool trunc = inpass && sig && (Wb + lane + 1 >= nbits);
                    const bool valid = inpass && !trunc;
                    const uint32_t sgn = lane < 63 ? ((uint32_t)(Wd >> (lane + 1)) & 1u) : nxtbit;
                    const uint32_t e = valid ? sh.lst[tok_base + rank - rd.lo] : 0u;
                    const uint64_t sigm = __ballot(valid && sig);
                    const uint64_t nsm = __ballot(valid && !sig);
                    const uint32_t nsig = (uint32_t)__popcll(sigm);
                    if (lsp_len + nsig > a.caps.lsp) { bad = true; done = true; break; }
                    const bool istail = valid && sig && (Wb + lane + 1 >= tail_start);
                    const uint64_t tm = __ballot(istail);
                    if (valid && sig) {
                        uint32_t t = lsp_len + mbcnt(sigm);
                        int32_t v = sgn ? base_val : -base_val;
                        lsp_idx[t] = e;
                        lsp_val[t] = istail ? 0 : v;
                        if (istail) {
                            uint32_t tp = ntail + mbcnt(tm);
                            if (tp < DEC_TAIL) { sh.tail[tp].idx = e; sh.tail[tp].val = v; sh.tail[tp].n = (uint32_t)n; sh.tail[tp].kind = 0; }
                        }
                    } else if (valid) {
                        lipn[lipn_len + mbcnt(nsm)] = e;
                    }
                    ntail += (uint32_t)__popcll(tm);
                    lsp_len += nsig;
                    lipn_len += (uint32_t)__popcll(nsm);
                    if (__ballot(trunc)) { done = true; break; }
                    if (cnt > m_rem) {
                        // the pass ends inside this window, at the start of token number m_rem
                        uint64_t pm = __ballot(isS && rank == m_rem);
                        P = Wb + (uint32_t)__builtin_ctzll(pm);
                        m_rem = 0;
                        cin = 0;
                    } else {
                        m_rem -= cnt;
                        tok_base += cnt;
                        P = Wb + 64;
                        cin = cout;
                    }
                }
                if (cin) P += 1;
                { uint32_t *t = lip; lip = lipn; lipn = t; }
                lip_len = lipn_len;
            }
            if (done) break;

            // ---------------- LIS pass (encoder_decoder.rs:379-436) ----------------
            uint32_t *cur = lis, *nxt = qa, *ret = qb;
            uint32_t cur_len = lis_len, ret_len = 0;
            while (cur_len > 0 && !done) {
                uint32_t nxt_len = 0;
                ListRd rd;
                rd.src = cur; rd.len = cur_len; rd.lo = 0; rd.hi = 0;
                for (uint32_t e0 = 0; e0 < cur_len && !done; e0 += 64) {
                    const uint32_t nE = (cur_len - e0) < 64u ? (cur_len - e0) : 64u;
                    const bool act = lane < nE;
                    list_need(sh, rd, e0, lane);
                    const uint32_t e = act ? sh.lst[e0 + lane - rd.lo] : 0u;
                    const uint32_t idx = e & ENT_IDX;
                    const bool isA = (e & ENT_A) != 0;
                    // only a type-A entry WITH offspring is followed by child bits when it fires
                    const bool leaf = (e & ENT_LEAF) != 0;
                    const uint64_t TA = __ballot(act && isA && !leaf);
                    // the window's bits span at most 64*9 positions; stage them now so the walk and the
                    // per-entry gathers below never trigger a refill in between
                    // (a register-chunk reload wants 130 words past its base: cover that too)
                    ensure(sh, bs, P, 64 * 9 + 130 * 32 + 256, lane);
                    // ---- position walk (uniform control flow): hop from fired type-A entry to fired type-A entry;
                    //      everything in between takes exactly one bit.  Each hop records a segment: entries
                    //      [i, ...) sit at stream position delta + entry index ----
                    uint64_t segmask = 0;
                    uint32_t nseg = 0, i = 0;
                    while (i < nE && P < nbits) {
                        const uint32_t widx = P >> 6, Wb = widx << 6;
                        uint32_t pos = P & 63u;
                        uint64_t lo, hi;
                        window(sh, bs, rc, widx, lane, lo, hi);
                        // length of a fired type-A entry starting at window position `lane`
                        uint32_t LAv;
                        {
                            uint64_t bb = lane ? ((lo >> lane) | (hi << (64 - lane))) : lo;
                            uint32_t pl = (uint32_t)(bb >> 1) & 0xFFu;
                            uint32_t ns = 0;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                uint32_t s = pl & 1u;
                                pl >>= 1 + s;
                                ns += s;
                            }
                            LAv = 5 + ns;
                        }
                        const uint32_t vb = (nbits - Wb) < 64u ? (nbits - Wb) : 64u;
                        while (i < nE && pos < vb) {
                            const uint64_t cand = lo & ((TA >> i) << pos);  // TA has no bits at or past nE
                            segmask |= 1ull << i;
                            if (lane == 0) sh.seg[nseg] = Wb + pos - i;
                            nseg++;
                            if (cand == 0) {
                                const uint32_t z = (vb - pos) < (nE - i) ? (vb - pos) : (nE - i);
                                i += z;
                                pos += z;
                            } else {
                                const uint32_t f = (uint32_t)__builtin_ctzll(cand);
                                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)LAv, (int)f);
                                i += f - pos + 1;
                                pos = f + len;
                            }
                        }
                        P = Wb + pos;
                    }
                    uint32_t mypos = POS_INVALID;
                    {
                        const uint32_t sidx = (uint32_t)__popcll(segmask & ((2ull << lane) - 1ull));
                        const uint32_t dl = sh.seg[sidx ? sidx - 1 : 0];
                        if (lane < i) mypos = dl + lane;
                    }
                    // ---- per-entry outputs ----
                    const bool have = act && mypos != POS_INVALID && mypos < nbits;
                    bool stop = act && !have;
                    uint32_t nQ = 0, nR = 0, nLIP = 0, nLSP = 0;
                    uint32_t sigm = 0, signm = 0, lipm = 0, tailm = 0, cb = 0, cr = 0, ccol = 0;
                    bool fired = false;
                    if (have) {
                        const uint32_t avail = nbits - mypos;
                        const uint32_t bits = (uint32_t)peek64(sh, bs, mypos) & 0xFFFFu;
                        fired = bits & 1u;
                        if (!fired) {
                            nR = 1;
                        } else if (leaf) {
                            // a set with no offspring cannot be significant in a real stream; the reference drops
                            // the entry (no offspring to read, has_descendents_past_offspring is false too)
                            fired = false;
                        } else {
                            uint32_t k, ii, jj;
                            decomp(g, idx, k, ii, jj);
                            cb = child_base(g, k, ii, jj, cr, ccol);
                            if (!isA) {
                                nQ = 4;
                            } else {
                                uint32_t o = 1;
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    if (!stop) {
                                        if (o >= avail) { stop = true; }
                                        else if ((bits >> o) & 1u) {
                                            if (o + 1 >= avail) { stop = true; }
                                            else {
                                                sigm |= 1u << q;
                                                signm |= ((bits >> (o + 1)) & 1u) << q;
                                                if (mypos + o + 1 >= tail_start) tailm |= 1u << q;
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
                    if (nxt_len + tQ > a.caps.lis || ret_len + tR > a.caps.lis || lip_len + tLIP > a.caps.lip ||
                        lsp_len + tLSP > a.caps.lsp) { bad = true; done = true; break; }
                    const uint64_t mT = __ballot(tailm != 0);
                    uint32_t exT = 0, tT = 0;
                    if (mT) {  // rare: an entry consumed one of the last 8 bits
                        uint64_t tot;
                        uint64_t ex = wave_exscan((uint64_t)__popc(tailm), tot, lane);
                        exT = (uint32_t)ex;
                        tT = (uint32_t)tot;
                    }
                    if (have) {
                        if (nR) {
                            ret[ret_len + mbcnt(mR)] = e;
                        } else if (!fired) {
                            // fired leaf: dropped
                        } else if (!isA) {
                            uint32_t oq = nxt_len + mbcnt(mQ1) + 4u * mbcnt(mQ4);
                            nxt[oq] = make_a_entry(cb, cr, ccol, H, W);
                            nxt[oq + 1] = make_a_entry(cb + 1, cr, ccol + 1, H, W);
                            nxt[oq + 2] = make_a_entry(cb + W, cr + 1, ccol, H, W);
                            nxt[oq + 3] = make_a_entry(cb + W + 1, cr + 1, ccol + 1, H, W);
                        } else {
                            uint32_t ol = lip_len + mbcnt(mL0) + 2u * mbcnt(mL1) + 4u * mbcnt(mL2);
                            uint32_t os = lsp_len + mbcnt(mS0) + 2u * mbcnt(mS1) + 4u * mbcnt(mS2);
                            uint32_t ot = ntail + exT;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                uint32_t ci = cb + (q >> 1) * W + (q & 1);
                                if (sigm & (1u << q)) {
                                    int32_t v = ((signm >> q) & 1u) ? base_val : -base_val;
                                    bool tl = (tailm >> q) & 1u;
                                    lsp_idx[os] = ci;
                                    lsp_val[os] = tl ? 0 : v;
                                    os++;
                                    if (tl) {
                                        if (ot < DEC_TAIL) { sh.tail[ot].idx = ci; sh.tail[ot].val = v; sh.tail[ot].n = (uint32_t)n; sh.tail[ot].kind = 0; }
                                        ot++;
                                    }
                                } else if (lipm & (1u << q)) {
                                    lip[ol++] = ci;
                                }
                            }
                            if (nQ) nxt[nxt_len + mbcnt(mQ1) + 4u * mbcnt(mQ4)] = idx;  // type B
                        }
                    }
                    nxt_len += tQ; ret_len += tR; lip_len += tLIP; lsp_len += tLSP; ntail += tT;
                    if (__ballot(stop)) done = true;
                }
                __syncthreads();  // entries of the next generation are read back through the LDS stage
                { uint32_t *t = cur; cur = nxt; nxt = t; }
                cur_len = nxt_len;
            }
            if (done) break;
            lis = ret; lis_len = ret_len;
            qa = cur; qb = nxt;

            // ---------------- refinement (encoder_decoder.rs:438-444) ----------------
            {
                const uint32_t left = nbits > P ? nbits - P : 0u;
                const uint32_t count = lsp_len0 < left ? lsp_len0 : left;
                for (uint32_t t0 = 0; t0 < count; t0 += 256) {
                    ensure(sh, bs, P + t0, 256 + 64, lane);
                    int32_t v[4];
                    uint32_t bit[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        uint32_t t = t0 + (uint32_t)u * 64 + lane;
                        v[u] = t < count ? lsp_val[t] : 0;
                        uint32_t pos = P + t;
                        bit[u] = t < count ? ((sh.w[(pos >> 5) - bs.cb] >> (pos & 31)) & 1u) : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        uint32_t t = t0 + (uint32_t)u * 64 + lane;
                        const bool in = t < count;
                        const bool tl = in && (P + t >= tail_start);
                        if (in && !tl) lsp_val[t] = set_bit_i32(v[u], (uint32_t)n, bit[u]);
                        const uint64_t tm = __ballot(tl);
                        if (tm) {
                            if (tl) {
                                uint32_t tp = ntail + mbcnt(tm);
                                if (tp < DEC_TAIL) { sh.tail[tp].idx = lsp_idx[t]; sh.tail[tp].val = (int32_t)bit[u]; sh.tail[tp].n = (uint32_t)n; sh.tail[tp].kind = 1; }
                            }
                            ntail += (uint32_t)__popcll(tm);
                        }
                    }
                }
                P += count;
                if (count < lsp_len0) { cut = count; done = true; }
            }
            if (n == 0) break;
        }

        // ---------------- scatter decoded values ----------------
        __syncthreads();
        for (uint32_t t = cut + lane; t < lsp_len; t += 64) {
            int32_t v = lsp_val[t];
            if (v) out[lsp_idx[t]] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        for (uint32_t t = lane; t < cut; t += 64) {
            int32_t v = lsp_val[t];
            if (v) out[lsp_idx[t]] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (lane == 0) {
            uint32_t nt = ntail < DEC_TAIL ? ntail : DEC_TAIL;
            for (uint32_t q = 0; q < nt; q++) {
                TailOp op = sh.tail[q];
                if (op.kind == 0) out[op.idx] = op.val;
                else out[op.idx] = set_bit_i32(out[op.idx], op.n, (uint32_t)op.val);
            }
            if (bad || ntail > DEC_TAIL) atomicOr(a.err, 1u);
        }
        __syncthreads();
    }
}

extern "C" int spiht_launch_decode(const DecArgs *a, hipStream_t st) {
    int grid = a->nslots < a->B ? a->nslots : a->B;
    if (grid < 1) return 0;
    hipLaunchKernelGGL(k_decode, dim3(grid), dim3(64), 0, st, *a);
    return (int)hipGetLastError();
}
