// SPIHT list encoder, one image on several CUs (gfx950): the latency of ONE encode call.
//
// k_encode (encode.hip) gives an image one workgroup, whose CU is then the bound -- about 7 cycles per list entry
// visited: 1.4 ms for a 1080p picture at 0.5 bpp, 20 ms for 4096 x 4096 at 1 bpp -- while the other 255 CUs idle.  That
// is the right shape for a batch (one image per CU) and the wrong one for a single call.  Here a group of G workgroups
// shares the passes of one image; the bits are those of /root/reference/src/encoder_decoder.rs:155-303, as k_encode's:
//   * a pass (the LIP pass, one LIS generation, the refinement pass) is cut into chunks of consecutive list entries; chunk
//     c goes to workgroup c mod G.  What a chunk needs from the chunks before it -- how many bits and list appends they
//     make -- comes from a single-pass scan with decoupled look-back: the chunk publishes its own counts (aggregate), sums
//     the aggregates of its predecessors back to the nearest one that has published its inclusive prefix, and publishes
//     its own inclusive prefix.  Descriptor words are 64-bit [pass number | count], written and read with agent-scope
//     atomics: self-validating, never cleared, coherent across XCDs;
//   * passes are separated by a barrier of the group (monotonic arrival counter + released epoch, __threadfence on both
//     sides: the lists one workgroup wrote are read by the others in the next pass);
//   * stream bits: a chunk stages its bits in LDS and writes whole words with plain stores, its first and last word --
//     shared with the neighbouring chunks -- with atomicOr (the slot is zero-filled before the launch);
//   * the planes in which the lists are still short are coded by workgroup 0 alone -- the same code, the scan over the
//     chunks kept in registers, no barriers -- and the others wait for its hand-over (WideCtl::go).
// Every workgroup of a group has to be resident at the same time.  The launcher keeps groups x G within what the device
// holds of this kernel (occupancy query) and chains such launches of one process -- but another process, or any kernel
// that holds the CUs, is outside that: so every wait is bounded by TIME (WIDE_TICKS, some tens of milliseconds), and a
// workgroup whose wait runs out marks the group "not co-resident" (WideCtl::bad & 2), on which every workgroup of the
// group leaves at its next look.  The launcher queues k_encode<redo> right behind this kernel: it codes exactly the images
// whose group gave up, one workgroup each, from scratch -- the call never fails for lack of residency (the reference's
// encode never fails on valid input, src/lib.rs:24-32), it only takes the single-workgroup time.
#include "common.h"
#include "encode_common.h"

#define WB_BLOCK 1024
#ifndef WIDE_U
#define WIDE_U 2      // LIS entries per thread and chunk (consecutive in the queue); 4: 82 registers spilled
#endif
#define WIDE_V 8      // LIP / LSP entries per thread and chunk; a multiple of 4
#define WIDE_TICKS 100000000ull  // s_memtime ticks (shader clocks: about 40 ms) a workgroup waits for another one of its group

__device__ __forceinline__ uint64_t wd_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wd_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t wu_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wu_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a wait of the group: false once it has lasted WIDE_TICKS or another workgroup has given up
__device__ __forceinline__ bool wide_wait_ok(const uint32_t *badp, uint64_t t0) {
    return !(wu_load(badp) & 2u) && __builtin_amdgcn_s_memtime() - t0 < WIDE_TICKS;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}

struct WideShared {
    static constexpr int WB = (9 * WIDE_U * WB_BLOCK + 31) / 32 + 3;  // (2 * WIDE_V * WB_BLOCK bits fit as well)
    static_assert(9 * WIDE_U >= 2 * WIDE_V, "staging buffer: the larger of the two chunk kinds");
    uint32_t wbuf[WB];
    uint64_t part[2][WB_BLOCK / 64];
    uint32_t E[4];        // the chunk's exclusive prefix, from wave 0 to the block
    uint32_t bc[12];      // broadcasts of thread 0
};

// packed exclusive scan over the block; one __syncthreads; `par` alternates the partial buffer
__device__ __forceinline__ uint64_t wide_exscan(uint64_t v, uint64_t &total, WideShared &sh, int par) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t lo = (uint32_t)inc, hi = (uint32_t)(inc >> 32);
        lo = (uint32_t)__shfl_up((int)lo, o);
        hi = (uint32_t)__shfl_up((int)hi, o);
        if (lane >= o) inc += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) sh.part[par][wave] = inc;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < WB_BLOCK / 64; wv++) {
        const uint64_t p = sh.part[par][wv];
        if (wv < wave) pre += p;
        tot += p;
    }
    total = tot;
    return pre + inc - v;
}

// wave 0 of the workgroup: exclusive prefix E of chunk c of pass `tag` over the chunks before it; publishes this chunk's
// aggregate A and its inclusive prefix.  false: the wait for a predecessor ran out (wide_wait_ok).
__device__ __forceinline__ bool wide_lookback(uint64_t *aggD, uint64_t *incD, uint32_t c, uint32_t tag, const uint32_t (&A)[4],
                                              uint32_t (&E)[4], uint32_t lane, const uint32_t *badp) {
    const uint64_t T = (uint64_t)tag << 32;
    const uint32_t mine = lane == 0 ? A[0] : lane == 1 ? A[1] : lane == 2 ? A[2] : lane == 3 ? A[3] : 0u;
    E[0] = E[1] = E[2] = E[3] = 0;
    if (c == 0) {
        if (lane < 4) wd_store(&incD[lane], T | mine);
        return true;
    }
    if (lane < 4) wd_store(&aggD[(size_t)c * 4 + lane], T | mine);
    int64_t j = (int64_t)c - 1;
    bool good = true;
    for (;;) {
        const int64_t p = j - (int64_t)lane;
        const bool valid = p >= 0;
        uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        bool ok = !valid, isinc = false;
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        for (;;) {
            if (!ok) {
                uint64_t w0 = wd_load(&incD[(size_t)p * 4]), w1 = wd_load(&incD[(size_t)p * 4 + 1]);
                uint64_t w2 = wd_load(&incD[(size_t)p * 4 + 2]), w3 = wd_load(&incD[(size_t)p * 4 + 3]);
                if ((w0 >> 32) == tag && (w1 >> 32) == tag && (w2 >> 32) == tag && (w3 >> 32) == tag) {
                    ok = true; isinc = true;
                } else {
                    w0 = wd_load(&aggD[(size_t)p * 4]); w1 = wd_load(&aggD[(size_t)p * 4 + 1]);
                    w2 = wd_load(&aggD[(size_t)p * 4 + 2]); w3 = wd_load(&aggD[(size_t)p * 4 + 3]);
                    ok = (w0 >> 32) == tag && (w1 >> 32) == tag && (w2 >> 32) == tag && (w3 >> 32) == tag;
                }
                if (ok) { v0 = (uint32_t)w0; v1 = (uint32_t)w1; v2 = (uint32_t)w2; v3 = (uint32_t)w3; }
            }
            const uint64_t okm = __ballot(ok), incm = __ballot(valid && isinc);
            // enough: an inclusive prefix at lane L with every nearer chunk known -- or everything known
            if (incm) {
                const uint32_t Lq = (uint32_t)__builtin_ctzll(incm);
                const uint64_t below = Lq ? (~0ull >> (64 - Lq)) : 0ull;
                if ((okm & below) == below) break;
            } else if (okm == ~0ull) {
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (!wide_wait_ok(badp, t0)) { good = false; break; }
        }
        if (!good) break;
        const uint64_t incm = __ballot(valid && isinc);
        const uint32_t Lq = incm ? (uint32_t)__builtin_ctzll(incm) : 64u;
        const bool use = valid && ok && lane <= Lq;
        E[0] += wave_sum_u32(use ? v0 : 0u);
        E[1] += wave_sum_u32(use ? v1 : 0u);
        E[2] += wave_sum_u32(use ? v2 : 0u);
        E[3] += wave_sum_u32(use ? v3 : 0u);
        if (incm) break;
        j -= 64;
        if (j < 0) break;  // (not reached: chunk 0 always has an inclusive prefix)
    }
    const uint32_t inc = (lane == 0 ? E[0] : lane == 1 ? E[1] : lane == 2 ? E[2] : lane == 3 ? E[3] : 0u) + mine;
    if (lane < 4) wd_store(&incD[(size_t)c * 4 + lane], T | inc);
    return good;
}

__global__ __launch_bounds__(WB_BLOCK) void k_encode_wide(EncArgs a, WideArgs w) {
    constexpr uint32_t BLOCK = WB_BLOCK;
    __shared__ WideShared sh;
    const Geom g = a.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t G = w.G, grp = blockIdx.x / G, wg = blockIdx.x % G;
    const uint32_t W = (uint32_t)g.w, H = (uint32_t)g.h;
    const int b = (int)grp;
    if (b >= a.B) return;
    WideCtl *ctl = w.ctl + grp;
    if (wu_load(&ctl->bad) & 2u) return;  // the group gave up before this workgroup got a CU: k_encode<redo> codes the image
    uint64_t *aggD = w.desc + (size_t)grp * 2 * w.maxchunks * 4, *incD = aggD + (size_t)w.maxchunks * 4;

    uint32_t *const lipbuf[2] = {a.lip0 + (size_t)grp * a.caps.lip, a.lip1 + (size_t)grp * a.caps.lip};
    uint32_t *const lsp = a.lsp + (size_t)grp * a.caps.lsp;
    uint32_t *const qbuf[3] = {a.lis0 + (size_t)grp * a.caps.lis, a.lis1 + (size_t)grp * a.caps.lis, a.lis2 + (size_t)grp * a.caps.lis};
    const int32_t *__restrict__ X = a.x + (size_t)b * g.n;
    const uint8_t *__restrict__ DM = a.dmsb + (size_t)b * g.n;
    const uint8_t *__restrict__ LM = a.lmsb + (size_t)b * g.n;
    uint32_t *outw = reinterpret_cast<uint32_t *>(a.out + (size_t)b * a.slot_stride);
    const uint64_t capb = a.slot_stride * 8;  // never write past the slot
    const uint64_t max_bits = a.max_bits < capb ? a.max_bits : capb;

    for (uint32_t t = tid; t < (uint32_t)WideShared::WB; t += BLOCK) sh.wbuf[t] = 0;
    __syncthreads();

    const uint32_t maxabs = a.maxabs[b];
    const int max_n = start_plane(maxabs, a.log2_thresh);
    bool bad = maxabs >= (1u << 30);

    // the coder's state: the same values in every workgroup that has joined
    uint32_t lipr = 0;            // lipbuf[lipr] is the LIP
    uint32_t lisr = 0;            // qbuf[lisr] is the LIS, the two behind it (mod 3) are free
    uint32_t lip_len = 0, lsp_len = 0, lis_len = 0;
    uint64_t bitpos = 0;
    int n = max_n;
    bool done = false, solo = true;
    uint32_t tag = 0, epoch = 0;
    int par = 0;
    uint32_t run[4] = {0, 0, 0, 0};  // solo: the scan over the chunks of a pass

    auto barrier = [&]() {
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            epoch++;
            const uint32_t arrived = atomicAdd(&ctl->bar_count, 1u) + 1u;
            if (arrived == epoch * G) {
                wu_store(&ctl->bar_gen, epoch);
            } else {
                const uint64_t t0 = __builtin_amdgcn_s_memtime();
                while (wu_load(&ctl->bar_gen) < epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (!wide_wait_ok(&ctl->bad, t0)) { atomicOr(&ctl->bad, 2u); break; }
                }
            }
            __threadfence();
        }
        __syncthreads();
    };
    // exclusive prefix E of chunk c over the chunks before it in this pass; A = the chunk's own counts
    auto chunk_prefix = [&](uint32_t c, const uint32_t (&A)[4], uint32_t (&E)[4]) {
        if (solo) {
#pragma unroll
            for (int k = 0; k < 4; k++) { E[k] = run[k]; run[k] += A[k]; }
            return;
        }
        if (tid < 64) {
            uint32_t Ew[4];
            if (!wide_lookback(aggD, incD, c, tag, A, Ew, lane, &ctl->bad)) atomicOr(&ctl->bad, 2u);
            if (lane == 0) { sh.E[0] = Ew[0]; sh.E[1] = Ew[1]; sh.E[2] = Ew[2]; sh.E[3] = Ew[3]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) E[k] = sh.E[k];
        __syncthreads();
    };
    // a token's bits into the LDS staging buffer (chunk bit 0 at LDS bit shft), cut at totv
    auto stage = [&](uint32_t bits, uint32_t nb, uint32_t off, uint32_t totv, uint32_t shft) {
        if (nb && off < totv) {
            const uint32_t nbv = (off + nb <= totv) ? nb : (totv - off);
            const uint32_t v = bits & ((1u << nbv) - 1u);  // nb <= 16
            const uint32_t p = shft + off, wi = p >> 5, bo = p & 31;
            if (v) {
                atomicOr(&sh.wbuf[wi], v << bo);
                if (bo + nbv > 32) atomicOr(&sh.wbuf[wi + 1], v >> (32 - bo));
            }
        }
    };
    // ... and out: whole words stored, the first / last word shared with the neighbouring chunks ORed
    auto flush = [&](uint64_t bit0, uint32_t totv) {
        __syncthreads();
        const uint32_t shft = (uint32_t)(bit0 & 31), endb = shft + totv, nw = (endb + 31) >> 5;
        const uint64_t w0 = bit0 >> 5;
        for (uint32_t wv = tid; wv < nw; wv += BLOCK) {
            const uint32_t val = sh.wbuf[wv];
            sh.wbuf[wv] = 0;
            const bool shared = (wv == 0 && shft != 0) || (wv == nw - 1 && (endb & 31) != 0);
            if (shared) { if (val) atomicOr(&outw[w0 + wv], val); }
            else outw[w0 + wv] = val;
        }
    };
    // end of a pass: its totals in every workgroup (wide: behind the group's barrier)
    auto pass_totals = [&](uint32_t nchunks, uint32_t (&Tt)[4]) {
        if (solo) {
#pragma unroll
            for (int k = 0; k < 4; k++) { Tt[k] = run[k]; run[k] = 0; }
            __syncthreads();  // the lists are read by other threads than wrote them
            return;
        }
        barrier();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) sh.bc[k] = nchunks ? wu_load(&ctl->tot[(tag & 1u) * 4 + k]) : 0u;  // (two sets: the next pass may finish early)
            // A capacity guard of THIS pass or an earlier one: every workgroup of the group reads the same answer behind
            // the barrier of pass `tag`.  (A workgroup that is already in pass tag + 1 may have tripped a guard there: that
            // one counts at the next barrier -- a workgroup leaving one barrier early would leave the others waiting.)
            const uint32_t at = wu_load(&ctl->bad_at);
            sh.bc[4] = (wu_load(&ctl->bad) & 2u) | ((at != 0u && ~at <= tag) ? 1u : 0u);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) Tt[k] = sh.bc[k];
        if (sh.bc[4]) bad = true;
        __syncthreads();
    };

    if (wg == 0) {
        // ---- initial LIP / LIS (encoder_decoder.rs:169-190): i, j, then channel innermost ----
        uint32_t *lip = lipbuf[0], *lis = qbuf[0];
        const uint32_t nroot = (uint32_t)(g.ll_h * g.ll_w * g.c);
        for (uint32_t base = 0; base < nroot; base += BLOCK) {
            const uint32_t t = base + tid;
            const bool act = t < nroot;
            uint32_t k = 0, i = 0, j = 0;
            if (act) {
                const uint32_t ij = t / (uint32_t)g.c;
                k = t - ij * (uint32_t)g.c;
                i = ij / (uint32_t)g.ll_w;
                j = ij - i * (uint32_t)g.ll_w;
            }
            const uint32_t idx = k * g.hw + i * W + j;
            const bool inlis = act && (((i | j) & 1u) != 0);
            uint64_t tot;
            const uint64_t ex = wide_exscan(inlis ? 1ull : 0ull, tot, sh, par);
            par ^= 1;
            if (act && t < a.caps.lip) lip[t] = idx;
            if (inlis && lis_len + (uint32_t)ex < a.caps.lis) lis[lis_len + (uint32_t)ex] = idx | ENT_A;
            lis_len += (uint32_t)tot;
        }
        lip_len = nroot;
        if (lip_len > a.caps.lip || lis_len > a.caps.lis) bad = true;
        done = bad || (max_bits == 0);
        __syncthreads();
    } else {
        // wait for workgroup 0's hand-over
        if (tid == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
            while (wu_load(&ctl->go) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (!wide_wait_ok(&ctl->bad, t0)) { atomicOr(&ctl->bad, 2u); break; }
            }
            __threadfence();
            for (int k = 0; k < 9; k++) sh.bc[k] = wu_load(&ctl->st[k]);
            sh.bc[9] = wu_load(&ctl->go);
        }
        __syncthreads();
        n = (int)sh.bc[0]; lip_len = sh.bc[1]; lsp_len = sh.bc[2]; lis_len = sh.bc[3];
        bitpos = (uint64_t)sh.bc[4] | ((uint64_t)sh.bc[5] << 32);
        lipr = sh.bc[6]; lisr = sh.bc[7];
        done = sh.bc[8] != 0 || sh.bc[9] == 0;
        solo = false;
        __syncthreads();
    }

    for (; !done; --n) {
        if (solo && G > 1 && lip_len + lsp_len + lis_len >= w.solo) {
            // hand-over: from this plane on the group works together (workgroup 0 gets here alone)
            __syncthreads();
            if (tid == 0) {
                wu_store(&ctl->st[0], (uint32_t)n); wu_store(&ctl->st[1], lip_len); wu_store(&ctl->st[2], lsp_len);
                wu_store(&ctl->st[3], lis_len); wu_store(&ctl->st[4], (uint32_t)bitpos); wu_store(&ctl->st[5], (uint32_t)(bitpos >> 32));
                wu_store(&ctl->st[6], lipr); wu_store(&ctl->st[7], lisr); wu_store(&ctl->st[8], 0u);
                __threadfence();  // the lists and the state before the flag
                wu_store(&ctl->go, 1u);
            }
            solo = false;
            tag = 0;  // pass numbers count from the hand-over on, in step with the workgroups that join here
            __syncthreads();
        }
        const uint32_t T = 1u << n;
        const uint32_t lsp_len0 = lsp_len;
        const uint32_t first = solo ? 0u : wg, stride = solo ? 1u : G;

        // ---- LIP pass (encoder_decoder.rs:207-222): one count to scan, the entries that turn significant ----
        {
            constexpr uint32_t V = WIDE_V, CHL = V * BLOCK;
            uint32_t *lip = lipbuf[lipr], *lipn = lipbuf[lipr ^ 1u];
            const uint32_t nch = (lip_len + CHL - 1) / CHL;
            tag++;
            for (uint32_t c = first; c < nch; c += stride) {
                const uint32_t base = c * CHL;
                const uint32_t cnt = (lip_len - base) < CHL ? (lip_len - base) : CHL;
                const uint32_t t0 = V * tid;
                const uint32_t before = t0 < cnt ? t0 : cnt;
                const uint32_t nact = t0 >= cnt ? 0u : ((cnt - t0) < V ? (cnt - t0) : V);
                uint32_t e[V];
                if (nact == V) {
#pragma unroll
                    for (uint32_t u = 0; u < V; u += 4) {  // (the lists are 256-byte aligned, V a multiple of 4)
                        const uint4 ee = *reinterpret_cast<const uint4 *>(lip + base + t0 + u);
                        e[u] = ee.x; e[u + 1] = ee.y; e[u + 2] = ee.z; e[u + 3] = ee.w;
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < V; u++) e[u] = u < nact ? lip[base + t0 + u] : 0u;
                }
                int32_t x[V];
#pragma unroll
                for (uint32_t u = 0; u < V; u++) x[u] = u < nact ? X[e[u]] : 0;
                uint32_t bits = 0, nb = 0, sgm = 0;
#pragma unroll
                for (uint32_t u = 0; u < V; u++) {
                    if (u < nact) {
                        if (iabs_u(x[u]) >= T) {
                            bits |= (1u | ((x[u] >= 0) ? 2u : 0u)) << nb;
                            nb += 2;
                            sgm |= 1u << u;
                        } else {
                            nb += 1;
                        }
                    }
                }
                const uint32_t ns = (uint32_t)__popc(sgm);
                uint64_t tot;
                const uint32_t pS = (uint32_t)wide_exscan((uint64_t)ns, tot, sh, par);
                par ^= 1;
                const uint32_t A[4] = {(uint32_t)tot, 0u, 0u, 0u};
                uint32_t E[4];
                chunk_prefix(c, A, E);
                const uint64_t bit0 = bitpos + (uint64_t)base + E[0];  // one bit per entry before, one more per significant one
                if (bit0 < max_bits) {  // (else the budget was spent before this chunk: nothing of it is kept)
                    if (lsp_len + E[0] + A[0] > a.caps.lsp || (base - E[0]) + (cnt - A[0]) > a.caps.lip) {
                        if (tid == 0) { atomicOr(&ctl->bad, 1u); atomicMax(&ctl->bad_at, ~tag); }
                        if (solo) bad = true;
                    } else {
                        uint32_t os = lsp_len + E[0] + pS, ol = (base - E[0]) + (before - pS);
#pragma unroll
                        for (uint32_t u = 0; u < V; u++) {
                            if (u < nact) {
                                if (sgm & (1u << u)) lsp[os++] = e[u]; else lipn[ol++] = e[u];
                            }
                        }
                        const uint64_t rem = max_bits - bit0;
                        const uint32_t tB = cnt + A[0], totv = (uint64_t)tB < rem ? tB : (uint32_t)rem;
                        stage(bits, nb, before + pS, totv, (uint32_t)(bit0 & 31));
                        flush(bit0, totv);
                    }
                }
                if (!solo && c == nch - 1 && tid == 0) {
                    uint32_t *tw = ctl->tot + (tag & 1u) * 4;
                    wu_store(&tw[0], E[0] + A[0]); wu_store(&tw[1], 0u); wu_store(&tw[2], 0u); wu_store(&tw[3], 0u);
                }
            }
            uint32_t Tt[4];
            pass_totals(nch, Tt);
            if (bad) { done = true; break; }
            bitpos += (uint64_t)lip_len + Tt[0];
            lsp_len += Tt[0];
            lip_len -= Tt[0];
            lipr ^= 1u;
            if (bitpos >= max_bits) { bitpos = max_bits; done = true; break; }
        }

        // ---- LIS pass, generation by generation (encoder_decoder.rs:224-284) ----
        // Four counts need the scan: significant offspring (S), fired type-A entries (FA), fired type-B entries (FB), fired
        // type-A entries that stay as type B (QA).  The rest follows: bits = entries + 4 FA + S, next generation = QA + 4 FB,
        // retained = entries - FA - FB, LIP appends = 4 FA - S.
        {
            constexpr uint32_t U = WIDE_U, CH = U * BLOCK;
            uint32_t curr = lisr, nxtr = (lisr + 1u) % 3u, retr = (lisr + 2u) % 3u;
            uint32_t cur_len = lis_len, ret_len = 0;
            uint32_t *lip = lipbuf[lipr];
            while (cur_len > 0 && !done) {
                uint32_t *cur = qbuf[curr], *nxt = qbuf[nxtr], *ret = qbuf[retr];
                const uint32_t nch = (cur_len + CH - 1) / CH;
                tag++;
                for (uint32_t c = first; c < nch; c += stride) {
                    const uint32_t base = c * CH;
                    const uint32_t cnt = (cur_len - base) < CH ? (cur_len - base) : CH;
                    const uint32_t t0 = U * tid;
                    const uint32_t before = t0 < cnt ? t0 : cnt;
                    const uint32_t nact = t0 >= cnt ? 0u : ((cnt - t0) < U ? (cnt - t0) : U);
                    uint32_t e[U];
                    if constexpr (U == 2) {
                        if (nact == 2) {  // (the queues are 256-byte aligned)
                            const uint2 ee = *reinterpret_cast<const uint2 *>(cur + base + t0);
                            e[0] = ee.x; e[1] = ee.y;
                        } else {
                            e[0] = nact ? cur[base + t0] : 0u; e[1] = 0u;
                        }
                    } else {
#pragma unroll
                        for (uint32_t u = 0; u < U; u++) e[u] = u < nact ? cur[base + t0 + u] : 0u;
                    }
                    uint32_t code[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t idx = e[u] & ENT_IDX;
                        code[u] = (u < nact && !(e[u] & ENT_LEAF)) ? ((e[u] & ENT_A) ? DM[idx] : LM[idx]) : 0u;
                    }
                    uint32_t cb[U], cr[U], ccol[U], qa1[U];
                    int32_t xc[U][4];
                    bool firedA[U], firedB[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const bool fired = u < nact && (int)code[u] > n;
                        firedA[u] = fired && (e[u] & ENT_A);
                        firedB[u] = fired && !(e[u] & ENT_A);
                        cb[u] = cr[u] = ccol[u] = 0; qa1[u] = 0;
                        xc[u][0] = xc[u][1] = xc[u][2] = xc[u][3] = 0;
                        if (fired) {
                            uint32_t k, i, j;
                            decomp(g, e[u] & ENT_IDX, k, i, j);
                            cb[u] = child_base(g, k, i, j, cr[u], ccol[u]);
                            if (firedA[u]) {
                                xc[u][0] = X[cb[u]]; xc[u][1] = X[cb[u] + 1]; xc[u][2] = X[cb[u] + W]; xc[u][3] = X[cb[u] + W + 1];
                                // has_descendents_past_offspring (encoder_decoder.rs:7-12), raw coordinates
                                qa1[u] = (4 * i + 3 < H && 4 * j + 3 < W) ? 1u : 0u;
                            }
                        }
                    }
                    uint32_t bits[U], nb[U], sigm[U];
                    uint32_t tS = 0, tFA = 0, tFB = 0, tQA = 0;
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        bits[u] = 0; nb[u] = u < nact ? 1u : 0u; sigm[u] = 0;
                        if (firedA[u]) {
                            uint32_t o = 1, bb = 1u;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (iabs_u(xc[u][q]) >= T) {
                                    bb |= (1u << o) | ((xc[u][q] >= 0 ? 1u : 0u) << (o + 1));
                                    o += 2;
                                    sigm[u] |= 1u << q;
                                } else {
                                    o += 1;
                                }
                            }
                            bits[u] = bb;
                            nb[u] = o;
                            tS += (uint32_t)__popc(sigm[u]);
                            tFA += 1;
                            tQA += qa1[u];
                        } else if (firedB[u]) {
                            bits[u] = 1u;
                            tFB += 1;
                        }
                    }
                    // S <= 4 CH, FA, FB, QA <= CH: 16 bits each
                    const uint64_t pk = (uint64_t)tS | ((uint64_t)tFA << 16) | ((uint64_t)tFB << 32) | ((uint64_t)tQA << 48);
                    uint64_t tot;
                    const uint64_t ex = wide_exscan(pk, tot, sh, par);
                    par ^= 1;
                    const uint32_t pS = (uint32_t)ex & 0xffffu, pFA = (uint32_t)(ex >> 16) & 0xffffu, pFB = (uint32_t)(ex >> 32) & 0xffffu,
                                   pQA = (uint32_t)(ex >> 48) & 0xffffu;
                    const uint32_t A[4] = {(uint32_t)tot & 0xffffu, (uint32_t)(tot >> 16) & 0xffffu, (uint32_t)(tot >> 32) & 0xffffu,
                                           (uint32_t)(tot >> 48) & 0xffffu};
                    uint32_t E[4];
                    chunk_prefix(c, A, E);
                    const uint64_t bit0 = bitpos + (uint64_t)base + 4ull * E[1] + E[0];
                    if (bit0 < max_bits) {
                        const uint32_t tQ = A[3] + 4 * A[2], tR = cnt - A[1] - A[2], tLIP = 4 * A[1] - A[0], tLSP = A[0], tB = cnt + 4 * A[1] + A[0];
                        const uint32_t bQ = E[3] + 4 * E[2], bR = ret_len + (base - E[1] - E[2]), bLIP = lip_len + 4 * E[1] - E[0], bLSP = lsp_len + E[0];
                        if (bQ + tQ > a.caps.lis || bR + tR > a.caps.lis || bLIP + tLIP > a.caps.lip || bLSP + tLSP > a.caps.lsp) {
                            if (tid == 0) { atomicOr(&ctl->bad, 1u); atomicMax(&ctl->bad_at, ~tag); }
                            if (solo) bad = true;
                        } else {
                            uint32_t oq = bQ + pQA + 4 * pFB, orr = bR + (before - pFA - pFB), ol = bLIP + 4 * pFA - pS, os = bLSP + pS,
                                     ob = before + 4 * pFA + pS;
                            const uint64_t rem = max_bits - bit0;
                            const uint32_t totv = (uint64_t)tB < rem ? tB : (uint32_t)rem;
                            const uint32_t shft = (uint32_t)(bit0 & 31);
#pragma unroll
                            for (uint32_t u = 0; u < U; u++) {
                                if (firedA[u]) {
#pragma unroll
                                    for (int q = 0; q < 4; q++) {
                                        const uint32_t ci = cb[u] + (q >> 1) * W + (q & 1);
                                        if (sigm[u] & (1u << q)) lsp[os++] = ci; else lip[ol++] = ci;
                                    }
                                    if (qa1[u]) nxt[oq++] = e[u] & ENT_IDX;  // type B
                                } else if (firedB[u]) {
                                    nxt[oq] = make_a_entry(cb[u], cr[u], ccol[u], H, W);
                                    nxt[oq + 1] = make_a_entry(cb[u] + 1, cr[u], ccol[u] + 1, H, W);
                                    nxt[oq + 2] = make_a_entry(cb[u] + W, cr[u] + 1, ccol[u], H, W);
                                    nxt[oq + 3] = make_a_entry(cb[u] + W + 1, cr[u] + 1, ccol[u] + 1, H, W);
                                    oq += 4;
                                } else if (u < nact) {
                                    ret[orr++] = e[u];
                                }
                                stage(bits[u], nb[u], ob, totv, shft);
                                ob += nb[u];
                            }
                            flush(bit0, totv);
                        }
                    }
                    if (!solo && c == nch - 1 && tid == 0) {
                        uint32_t *tw = ctl->tot + (tag & 1u) * 4;
                        wu_store(&tw[0], E[0] + A[0]); wu_store(&tw[1], E[1] + A[1]);
                        wu_store(&tw[2], E[2] + A[2]); wu_store(&tw[3], E[3] + A[3]);
                    }
                }
                uint32_t Tt[4];
                pass_totals(nch, Tt);
                if (bad) { done = true; break; }
                bitpos += (uint64_t)cur_len + 4ull * Tt[1] + Tt[0];
                ret_len += cur_len - Tt[1] - Tt[2];
                lip_len += 4 * Tt[1] - Tt[0];
                lsp_len += Tt[0];
                cur_len = Tt[3] + 4 * Tt[2];
                if (bitpos >= max_bits) { bitpos = max_bits; done = true; break; }
                { const uint32_t t = curr; curr = nxtr; nxtr = t; }
            }
            if (done) break;
            // retained entries become the LIS of the next plane; the other two buffers are free
            lisr = retr;
            lis_len = ret_len;
        }

        // ---- refinement (encoder_decoder.rs:286-292): bit t of the pass belongs to LSP entry t, nothing to scan ----
        {
            constexpr uint32_t V = WIDE_V, CHL = V * BLOCK;
            const uint32_t nch = (lsp_len0 + CHL - 1) / CHL;
            for (uint32_t c = first; c < nch; c += stride) {
                const uint32_t base = c * CHL;
                const uint32_t cnt = (lsp_len0 - base) < CHL ? (lsp_len0 - base) : CHL;
                const uint32_t t0 = V * tid;
                const uint32_t nact = t0 >= cnt ? 0u : ((cnt - t0) < V ? (cnt - t0) : V);
                const uint64_t bit0 = bitpos + base;
                if (bit0 >= max_bits) continue;  // (the same in every thread of the block)
                uint32_t e[V];
                if (nact == V) {
#pragma unroll
                    for (uint32_t u = 0; u < V; u += 4) {
                        const uint4 ee = *reinterpret_cast<const uint4 *>(lsp + base + t0 + u);
                        e[u] = ee.x; e[u + 1] = ee.y; e[u + 2] = ee.z; e[u + 3] = ee.w;
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < V; u++) e[u] = u < nact ? lsp[base + t0 + u] : 0u;
                }
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t u = 0; u < V; u++) bits |= (u < nact ? ((iabs_u(X[e[u]]) >> n) & 1u) : 0u) << u;
                const uint64_t rem = max_bits - bit0;
                const uint32_t totv = (uint64_t)cnt < rem ? cnt : (uint32_t)rem;
                stage(bits, nact, t0 < cnt ? t0 : cnt, totv, (uint32_t)(bit0 & 31));
                flush(bit0, totv);
                __syncthreads();  // the staging buffer is clean before the next chunk's bits go in
            }
            bitpos += lsp_len0;
            if (bitpos >= max_bits) { bitpos = max_bits; done = true; }
        }
        if (n == 0) break;
    }

    if (wg == 0) {
        __syncthreads();
        if (tid == 0) {
            if (solo) {  // never handed over: release the workgroups that wait for it
                wu_store(&ctl->st[8], 1u);
                __threadfence();
                wu_store(&ctl->go, 1u);
            }
            const uint32_t cb_ = wu_load(&ctl->bad);
            if (!(cb_ & 2u)) {  // (a group that gave up reports nothing: k_encode<redo> codes the image and reports for it)
                a.out_nbits[b] = bitpos;
                a.out_maxn[b] = (uint8_t)max_n;
                if (bad || (cb_ & 1u)) atomicOr(a.err, maxabs >= (1u << 30) ? 2u : 1u);
                if (a.max_bits > capb && bitpos >= capb) atomicOr(a.err, 4u);
            }
        }
    }
}

// workgroups of k_encode_wide a CU holds at once (the launcher keeps groups x G within that times the CUs); < 1: unknown
extern "C" int spiht_wide_groups_per_cu(void) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_wide, WB_BLOCK, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int spiht_launch_encode_wide(const EncArgs *a, const WideArgs *w, int groups, hipStream_t st) {
    if (groups < 1 || w->G < 1) return 0;
    hipLaunchKernelGGL(k_encode_wide, dim3(groups * w->G), dim3(WB_BLOCK), 0, st, *a, *w);
    return (int)hipGetLastError();
}
