// SPIHT list encoder (gfx950): one workgroup owns one image at a time.
//
// Reproduces the bit order of the reference encoder (/root/reference/src/encoder_decoder.rs:155-303)
// exactly, including duplicated tree nodes and the FIFO "entries appended during a
// LIS pass are processed in the same pass" rule (Q6), but replaces
//   * the recursive significance search by one byte lookup in the D/L pyramid (pyramid.hip),
//   * per-entry deque pushes by block-wide prefix sums: every pass is cut into chunks of BLOCK
//     list entries; each thread works out how many bits / list appends its entry produces, one
//     packed 64-bit exclusive scan gives every thread its offsets, bits are ORed into an LDS
//     staging buffer and flushed as whole 32-bit words.
// The LIS pass is run generation by generation (generation g+1 = entries appended while
// processing generation g, in order), which is the FIFO order.
//
// Integer/bit work, latency- and LDS-bound; no roofline claim.
#include "common.h"
#include "encode_common.h"

#ifndef ENC_BLOCK
#define ENC_BLOCK 1024
#endif

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int o) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_up((int)lo, o);
    hi = (uint32_t)__shfl_up((int)hi, o);
    return ((uint64_t)hi << 32) | lo;
}

template <int BLOCK>
struct EncShared {
    static constexpr int WB = (9 * BLOCK + 31) / 32 + 2;
    uint32_t wbuf[2][WB];
    uint64_t part[2][BLOCK / 64];
};

// packed exclusive scan over the block; one __syncthreads; `par` alternates the partial buffer
template <int BLOCK>
__device__ __forceinline__ uint64_t block_exscan(uint64_t v, uint64_t &total, EncShared<BLOCK> &sh, int par) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = shfl_up_u64(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh.part[par][wave] = inc;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < BLOCK / 64; wv++) {
        uint64_t p = sh.part[par][wv];
        if (wv < wave) pre += p;
        tot += p;
    }
    total = tot;
    return pre + inc - v;
}

// Emit this thread's `nb` bits (LSB first) at block-relative bit offset `off`; `tot` = bits of the whole
// block this iteration.  Stream position `bitpos` is advanced; bits at or past max_bits are dropped.
// One __syncthreads.  `par` alternates the staging buffer (the partial last word is carried into the other).
template <int BLOCK>
__device__ __forceinline__ void emit_bits(uint32_t bits, uint32_t nb, uint32_t off, uint32_t tot, uint64_t &bitpos,
                                          uint64_t max_bits, uint32_t *__restrict__ outw, EncShared<BLOCK> &sh,
                                          int par) {
    const uint32_t shft = (uint32_t)(bitpos & 31);
    const uint64_t rem = max_bits - bitpos;  // bitpos < max_bits on entry
    const uint32_t totv = (uint64_t)tot < rem ? tot : (uint32_t)rem;
    if (nb && off < totv) {
        uint32_t nbv = (off + nb <= totv) ? nb : (totv - off);
        uint32_t v = bits & ((1u << nbv) - 1u);
        uint32_t p = shft + off;
        uint32_t wi = p >> 5, bo = p & 31;
        if (v) {
            atomicOr(&sh.wbuf[par][wi], v << bo);
            if (bo + nbv > 32) atomicOr(&sh.wbuf[par][wi + 1], v >> (32 - bo));
        }
    }
    __syncthreads();
    const uint32_t endb = shft + totv;
    const uint32_t nfull = endb >> 5;
    const uint64_t w0 = bitpos >> 5;
    for (uint32_t wv = threadIdx.x; wv <= nfull; wv += BLOCK) {
        uint32_t val = sh.wbuf[par][wv];
        sh.wbuf[par][wv] = 0;
        if (wv < nfull) outw[w0 + wv] = val;
        else if (val) atomicOr(&sh.wbuf[par ^ 1][0], val);
    }
    bitpos += totv;
}

template <int BLOCK>
#ifndef ENC_WAVES_PER_EU
#define ENC_WAVES_PER_EU 6  // caps the kernel at 80 VGPRs (5 spilled; same speed alone): beside a resident encoder
#endif                      // workgroup (16 wavefronts) three instead of two DWT wavefronts per SIMD fit -- 19.5 instead
                            // of 19.75 ms per pipelined step.  64 VGPRs: the encoder itself takes 4.7 instead of 2.7 ms
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(ENC_WAVES_PER_EU, ENC_WAVES_PER_EU)))
void k_encode(EncArgs a) {
    __shared__ EncShared<BLOCK> sh;
    const Geom g = a.g;
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = blockIdx.x;
    const uint32_t W = (uint32_t)g.w, H = (uint32_t)g.h;

    uint32_t *lipA = a.lip0 + (size_t)slot * a.caps.lip;
    uint32_t *lipB = a.lip1 + (size_t)slot * a.caps.lip;
    uint32_t *lsp = a.lsp + (size_t)slot * a.caps.lsp;
    uint32_t *q0 = a.lis0 + (size_t)slot * a.caps.lis;
    uint32_t *q1 = a.lis1 + (size_t)slot * a.caps.lis;
    uint32_t *q2 = a.lis2 + (size_t)slot * a.caps.lis;

    for (int b = (int)blockIdx.x; b < a.B; b += (int)gridDim.x) {
        // behind k_encode_wide (a.redo: one slot per image): only the images whose group of workgroups gave up for lack of
        // residency are coded here, from scratch -- what that kernel left in the slot goes first
        if (a.redo) {
            if (!(a.redo[b].bad & 2u)) continue;
            uint32_t *zw = reinterpret_cast<uint32_t *>(a.out + (size_t)b * a.slot_stride);
            for (uint64_t t = tid; t < a.slot_stride / 4; t += BLOCK) zw[t] = 0;
            __syncthreads();
        }
        const int32_t *__restrict__ X = a.x + (size_t)b * g.n;
        const uint8_t *__restrict__ DM = a.dmsb + (size_t)b * g.n;
        const uint8_t *__restrict__ LM = a.lmsb + (size_t)b * g.n;
        uint32_t *__restrict__ outw = reinterpret_cast<uint32_t *>(a.out + (size_t)b * a.slot_stride);
        const uint64_t capb = a.slot_stride * 8;  // never write past the slot
        const uint64_t max_bits = a.max_bits < capb ? a.max_bits : capb;

        for (uint32_t t = tid; t < (uint32_t)EncShared<BLOCK>::WB; t += BLOCK) { sh.wbuf[0][t] = 0; sh.wbuf[1][t] = 0; }
        __syncthreads();

        const uint32_t maxabs = a.maxabs[b];
        const int max_n = start_plane(maxabs, a.log2_thresh);
        bool bad = maxabs >= (1u << 30);

        uint32_t *lip = lipA, *lipn = lipB;
        uint32_t *lis = q0, *qa = q1, *qb = q2;
        uint32_t lip_len = 0, lsp_len = 0, lis_len = 0;
        uint64_t bitpos = 0;
        int par = 0;     // scan partial buffer parity
        int wpar = 0;    // bit staging buffer parity

        // ---- initial LIP / LIS (encoder_decoder.rs:169-190): i, j, then channel innermost ----
        const uint32_t nroot = (uint32_t)(g.ll_h * g.ll_w * g.c);
        for (uint32_t base = 0; base < nroot; base += BLOCK) {
            uint32_t t = base + tid;
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
            uint64_t tot;
            uint64_t ex = block_exscan<BLOCK>(inlis ? 1ull : 0ull, tot, sh, par);
            par ^= 1;
            if (act && t < a.caps.lip) lip[t] = idx;
            if (inlis && lis_len + (uint32_t)ex < a.caps.lis) lis[lis_len + (uint32_t)ex] = idx | ENT_A;
            lis_len += (uint32_t)tot;
        }
        lip_len = nroot;
        if (lip_len > a.caps.lip || lis_len > a.caps.lis) bad = true;

        bool done = bad || (max_bits == 0);
        for (int n = max_n; !done; --n) {
            const uint32_t T = 1u << n;
            const uint32_t lsp_len0 = lsp_len;

            // ---- LIP pass (encoder_decoder.rs:207-222) ----
            // two consecutive entries per thread: half as many scans / barriers per entry (the pass is bound by those,
            // not by its loads: 16 wavefronts per CU hide them)
            uint32_t lipn_len = 0;
            for (uint32_t base = 0; base < lip_len && !done; base += 2 * BLOCK) {
                const uint32_t r = base + 2 * tid;
                const bool act0 = r < lip_len, act1 = r + 1 < lip_len;
                uint32_t e0 = 0, e1 = 0;
                if (act1) {
                    const uint2 ee = *reinterpret_cast<const uint2 *>(lip + r);  // r is even, the list is 256-byte aligned
                    e0 = ee.x; e1 = ee.y;
                } else if (act0) {
                    e0 = lip[r];
                }
                const int32_t x0 = act0 ? X[e0] : 0, x1 = act1 ? X[e1] : 0;
                const bool sig0 = act0 && iabs_u(x0) >= T, sig1 = act1 && iabs_u(x1) >= T;
                const uint32_t b0 = sig0 ? (1u | ((x0 >= 0) ? 2u : 0u)) : 0u, b1 = sig1 ? (1u | ((x1 >= 0) ? 2u : 0u)) : 0u;
                const uint32_t nb0 = act0 ? (sig0 ? 2u : 1u) : 0u, nb1 = act1 ? (sig1 ? 2u : 1u) : 0u;
                const uint32_t ns = (sig0 ? 1u : 0u) + (sig1 ? 1u : 0u);
                const uint32_t nn = ((act0 && !sig0) ? 1u : 0u) + ((act1 && !sig1) ? 1u : 0u);
                uint64_t pk = (uint64_t)(nb0 + nb1) | ((uint64_t)ns << 16) | ((uint64_t)nn << 32);
                uint64_t tot;
                uint64_t ex = block_exscan<BLOCK>(pk, tot, sh, par);
                par ^= 1;
                uint32_t totLSP = (uint32_t)(tot >> 16) & 0xffffu;
                if (lsp_len + totLSP > a.caps.lsp) { bad = true; done = true; break; }
                uint32_t os = lsp_len + ((uint32_t)(ex >> 16) & 0xffffu), ol = lipn_len + (uint32_t)(ex >> 32);
                if (sig0) lsp[os++] = e0; else if (act0) lipn[ol++] = e0;
                if (sig1) lsp[os] = e1; else if (act1) lipn[ol] = e1;
                emit_bits<BLOCK>(b0 | (b1 << nb0), nb0 + nb1, (uint32_t)ex & 0xffffu, (uint32_t)tot & 0xffffu, bitpos, max_bits, outw,
                                 sh, wpar);
                wpar ^= 1;
                lsp_len += totLSP;
                lipn_len += (uint32_t)(tot >> 32);
                if (bitpos >= max_bits) done = true;
            }
            if (done) break;
            { uint32_t *t = lip; lip = lipn; lipn = t; }
            lip_len = lipn_len;

            // ---- LIS pass, generation by generation (encoder_decoder.rs:224-284) ----
            uint32_t *cur = lis, *nxt = qa, *ret = qb;
            uint32_t cur_len = lis_len, ret_len = 0;
            while (cur_len > 0 && !done) {
                uint32_t nxt_len = 0;
                for (uint32_t base = 0; base < cur_len && !done; base += BLOCK) {
                    uint32_t r = base + tid;
                    bool act = r < cur_len;
                    uint32_t e = act ? cur[r] : 0;
                    uint32_t idx = e & ENT_IDX;
                    bool isA = (e & ENT_A) != 0;
                    uint32_t code = (act && !(e & ENT_LEAF)) ? (isA ? DM[idx] : LM[idx]) : 0;
                    bool fired = act && (int)code > n;
                    uint32_t bits = 0, nb = act ? 1u : 0u, nQ = 0, nR = 0, nLIP = 0, nLSP = 0;
                    uint32_t cb = 0, sigm = 0, cr = 0, ccol = 0;
                    if (fired) {
                        uint32_t k, i, j;
                        decomp(g, idx, k, i, j);
                        cb = child_base(g, k, i, j, cr, ccol);
                        bits = 1u;
                        if (isA) {
                            int32_t xc[4];
                            xc[0] = X[cb]; xc[1] = X[cb + 1]; xc[2] = X[cb + W]; xc[3] = X[cb + W + 1];
                            uint32_t o = 1;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                bool s = iabs_u(xc[q]) >= T;
                                if (s) {
                                    bits |= (1u << o) | ((xc[q] >= 0 ? 1u : 0u) << (o + 1));
                                    o += 2;
                                    sigm |= 1u << q;
                                } else {
                                    o += 1;
                                }
                            }
                            nb = o;
                            nLSP = (uint32_t)__popc(sigm);
                            nLIP = 4 - nLSP;
                            // has_descendents_past_offspring (encoder_decoder.rs:7-12), raw coordinates
                            nQ = (4 * i + 3 < H && 4 * j + 3 < W) ? 1u : 0u;
                        } else {
                            nQ = 4;
                        }
                    } else if (act) {
                        nR = 1;
                    }
                    uint64_t pk = (uint64_t)nb | ((uint64_t)nQ << 14) | ((uint64_t)nR << 27) | ((uint64_t)nLIP << 38) |
                                  ((uint64_t)nLSP << 51);
                    uint64_t tot;
                    uint64_t ex = block_exscan<BLOCK>(pk, tot, sh, par);
                    par ^= 1;
                    uint32_t tQ = (uint32_t)(tot >> 14) & 0x1fffu, tR = (uint32_t)(tot >> 27) & 0x7ffu;
                    uint32_t tLIP = (uint32_t)(tot >> 38) & 0x1fffu, tLSP = (uint32_t)(tot >> 51) & 0x1fffu;
                    if (nxt_len + tQ > a.caps.lis || ret_len + tR > a.caps.lis || lip_len + tLIP > a.caps.lip ||
                        lsp_len + tLSP > a.caps.lsp) { bad = true; done = true; break; }
                    if (fired) {
                        uint32_t oq = nxt_len + ((uint32_t)(ex >> 14) & 0x1fffu);
                        if (isA) {
                            uint32_t ol = lip_len + ((uint32_t)(ex >> 38) & 0x1fffu);
                            uint32_t os = lsp_len + ((uint32_t)(ex >> 51) & 0x1fffu);
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                uint32_t ci = cb + (q >> 1) * W + (q & 1);
                                if (sigm & (1u << q)) lsp[os++] = ci; else lip[ol++] = ci;
                            }
                            if (nQ) nxt[oq] = idx;  // type B
                        } else {
                            nxt[oq] = make_a_entry(cb, cr, ccol, H, W);
                            nxt[oq + 1] = make_a_entry(cb + 1, cr, ccol + 1, H, W);
                            nxt[oq + 2] = make_a_entry(cb + W, cr + 1, ccol, H, W);
                            nxt[oq + 3] = make_a_entry(cb + W + 1, cr + 1, ccol + 1, H, W);
                        }
                    } else if (act) {
                        ret[ret_len + ((uint32_t)(ex >> 27) & 0x7ffu)] = e;
                    }
                    emit_bits<BLOCK>(bits, nb, (uint32_t)ex & 0x3fffu, (uint32_t)tot & 0x3fffu, bitpos, max_bits, outw, sh, wpar);
                    wpar ^= 1;
                    nxt_len += tQ; ret_len += tR; lip_len += tLIP; lsp_len += tLSP;
                    if (bitpos >= max_bits) done = true;
                }
                // the next generation's entries are read by other threads than wrote them
                __syncthreads();
                { uint32_t *t = cur; cur = nxt; nxt = t; }
                cur_len = nxt_len;
            }
            if (done) break;
            // retained entries become the LIS of the next plane; `cur`/`nxt` are the two free buffers
            lis = ret; lis_len = ret_len;
            qa = cur; qb = nxt;

            // ---- refinement (encoder_decoder.rs:286-292) ----
            for (uint32_t base = 0; base < lsp_len0 && !done; base += 2 * BLOCK) {  // two entries per thread, as above
                const uint32_t r = base + 2 * tid;
                const bool act0 = r < lsp_len0, act1 = r + 1 < lsp_len0;
                uint32_t e0 = 0, e1 = 0;
                if (act1) {
                    const uint2 ee = *reinterpret_cast<const uint2 *>(lsp + r);
                    e0 = ee.x; e1 = ee.y;
                } else if (act0) {
                    e0 = lsp[r];
                }
                const uint32_t bit0 = act0 ? ((iabs_u(X[e0]) >> n) & 1u) : 0u, bit1 = act1 ? ((iabs_u(X[e1]) >> n) & 1u) : 0u;
                const uint32_t cnt = (lsp_len0 - base) < 2u * BLOCK ? (lsp_len0 - base) : 2u * BLOCK;
                emit_bits<BLOCK>(bit0 | (bit1 << 1), (act0 ? 1u : 0u) + (act1 ? 1u : 0u), 2 * tid, cnt, bitpos, max_bits, outw, sh,
                                 wpar);
                wpar ^= 1;
                if (bitpos >= max_bits) done = true;
            }
            if (n == 0) break;
        }

        // flush the partial last word (carried in the staging buffer of the next iteration)
        __syncthreads();
        if (tid == 0) {
            if (bitpos & 31) outw[bitpos >> 5] = sh.wbuf[wpar][0];
            a.out_nbits[b] = bitpos;
            a.out_maxn[b] = (uint8_t)max_n;
            if (bad) atomicOr(a.err, maxabs >= (1u << 30) ? 2u : 1u);
            if (a.max_bits > capb && bitpos >= capb) atomicOr(a.err, 4u);
        }
        __syncthreads();
    }
}

extern "C" int spiht_launch_encode(const EncArgs *a, hipStream_t st) {
    int grid = a->nslots < a->B ? a->nslots : a->B;
    if (grid < 1) return 0;
    hipLaunchKernelGGL(k_encode<ENC_BLOCK>, dim3(grid), dim3(ENC_BLOCK), 0, st, *a);
    return (int)hipGetLastError();
}
