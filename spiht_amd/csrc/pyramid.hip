// Significance pyramid for the SPIHT list coder (gfx950).
//
// Replaces the recursive significance search of the reference (is_set_sig / is_l_sig,
// /root/reference/src/encoder_decoder.rs:78-121), which walks a whole sub-tree for every
// insignificant LIS entry in every bit-plane.  Here the set maxima are computed once:
//
//   S(p) = max(|x_p|, D(p))          D(p) = max over offspring o of S(o)
//   L(p) = max over offspring o that themselves have offspring of D(o)
//
// with offspring as get_offspring (encoder_decoder.rs:43-75): index doubling for every node
// outside the ll_h x ll_w root block, the 2x2-block remap for root nodes.  Only the position of
// the most significant bit matters ("D(p) >= 2^n"), so one byte per node is stored:
// code = 0 for an all-zero/empty set, else 1 + floor(log2(max)).  "significant at plane n" is
// code > n.
//
// A node (i,j) outside the root block has a sub-tree of depth >= d iff i*2^d+1 < h and
// j*2^d+1 < w (its top-left descendant chain is the longest).  k_pyr_12 handles the nodes of depth 1 and 2 in one
// launch (15/16 of the array is read there: a thread owns one depth->=2 node, loads its 2x2 children and 4x4
// grandchildren up front, and a lane pair writes the codes of the children's row as one 4-byte store); launch
// `round` d >= 3 handles the nodes of depth exactly d, whose offspring were finished before.
// HBM-bound: 4 B read per coefficient, ~1/4 + 1/16 B written.
#include "common.h"
#include <string.h>

__device__ __forceinline__ uint32_t msb_code(uint32_t v) { return v ? 32u - (uint32_t)__clz((int)v) : 0u; }
__device__ __forceinline__ uint32_t iabs_u(int32_t x) { return (uint32_t)(x < 0 ? -x : x); }

// max |x| per image (encoder_decoder.rs:165).  grid: (blocks, B)
__global__ __launch_bounds__(256) void k_absmax(const int32_t *__restrict__ x, uint32_t n, uint32_t *__restrict__ maxabs) {
    const int32_t *xi = x + (size_t)blockIdx.y * n;
    uint32_t m = 0;
    uint32_t n4 = n >> 2;
    // 16-byte loads when the image base is 16-byte aligned (n multiple of 4 keeps every image aligned)
    if ((n & 3u) == 0 && (reinterpret_cast<uintptr_t>(xi) & 15u) == 0) {
        const int4 *x4 = reinterpret_cast<const int4 *>(xi);
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += gridDim.x * blockDim.x) {
            int4 v = x4[t];
            m = max(m, max(max(iabs_u(v.x), iabs_u(v.y)), max(iabs_u(v.z), iabs_u(v.w))));
        }
    } else {
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
            m = max(m, iabs_u(xi[t]));
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    __shared__ uint32_t s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s[0], s[1]), max(s[2], s[3]));
        if (m) atomicMax(&maxabs[blockIdx.y], m);
    }
}

// 4-byte-aligned vector types: gfx950 runs in unaligned-access mode, so these become one dwordx4 / dwordx2 /
// dword / short access even though rows of an odd-width array start on arbitrary 4-byte (1-byte) boundaries
struct __attribute__((packed, aligned(4))) i4u { int32_t v[4]; };
struct __attribute__((packed, aligned(4))) i2u { int32_t v[2]; };
struct __attribute__((packed, aligned(1))) b4u { uint8_t v[4]; };
struct __attribute__((packed, aligned(1))) b2u { uint8_t v[2]; };

// One index-doubling round: thread = two horizontally adjacent parents (i, 2t), (i, 2t+1): their 2x4 block of
// children is read with one 16-byte load per child row.  1-D grid, XCD-contiguous tile order.
// tiles: (ceil(npairs/64), ceil(gi/4), B*c), block (64,4).
__global__ __launch_bounds__(256) void k_pyr_round(PyrArgs a, uint32_t gx, uint32_t gy, uint32_t gz) {
    const Geom g = a.g;
    const int d = a.round;
    uint32_t bx, by, bz;
    {
        const uint32_t nt = gx * gy * gz, L = blockIdx.x;
        const uint32_t q = nt >> 3, r = nt & 7u, x = L & 7u, jj = L >> 3;
        const uint32_t T = x * q + (x < r ? x : r) + jj;
        bx = T % gx;
        const uint32_t t2 = T / gx;
        by = t2 % gy;
        bz = t2 / gy;
    }
    const uint32_t t = bx * 64 + threadIdx.x;   // parent pair
    const uint32_t i = by * 4 + threadIdx.y;
    const uint32_t h = (uint32_t)g.h, w = (uint32_t)g.w;
    const uint64_t s1 = 1ull << d, s2 = 2ull << d;
    if ((uint64_t)i * s1 + 1 >= h) return;
    const uint32_t j0 = 2 * t, j1 = 2 * t + 1;
    // depth exactly d, and not a root-block node (k_pyr_ll handles those)
    bool v0 = (uint64_t)j0 * s1 + 1 < w && !((uint64_t)i * s2 + 1 < h && (uint64_t)j0 * s2 + 1 < w) &&
              !(i < (uint32_t)g.ll_h && j0 < (uint32_t)g.ll_w);
    bool v1 = (uint64_t)j1 * s1 + 1 < w && !((uint64_t)i * s2 + 1 < h && (uint64_t)j1 * s2 + 1 < w) &&
              !(i < (uint32_t)g.ll_h && j1 < (uint32_t)g.ll_w);
    if (!v0 && !v1) return;
    const size_t base = (size_t)bz * g.hw;
    const int32_t *__restrict__ x = a.x + base;
    uint8_t *__restrict__ dm = a.dmsb + base;
    uint8_t *__restrict__ lm = a.lmsb + base;
    const uint32_t ci = 2 * i, cj = 4 * t;
    const uint32_t c0 = ci * w + cj;
    int32_t xv[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    uint8_t dv[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool wide = cj + 3 < w;  // both parents' children exist as columns
    if (wide) {
        i4u r0 = *reinterpret_cast<const i4u *>(x + c0), r1 = *reinterpret_cast<const i4u *>(x + c0 + w);
#pragma unroll
        for (int q = 0; q < 4; q++) { xv[0][q] = r0.v[q]; xv[1][q] = r1.v[q]; }
        if (d > 1) {
            b4u e0 = *reinterpret_cast<const b4u *>(dm + c0), e1 = *reinterpret_cast<const b4u *>(dm + c0 + w);
#pragma unroll
            for (int q = 0; q < 4; q++) { dv[0][q] = e0.v[q]; dv[1][q] = e1.v[q]; }
        }
    } else {
        i2u r0 = *reinterpret_cast<const i2u *>(x + c0), r1 = *reinterpret_cast<const i2u *>(x + c0 + w);
        xv[0][0] = r0.v[0]; xv[0][1] = r0.v[1]; xv[1][0] = r1.v[0]; xv[1][1] = r1.v[1];
        if (d > 1) {
            b2u e0 = *reinterpret_cast<const b2u *>(dm + c0), e1 = *reinterpret_cast<const b2u *>(dm + c0 + w);
            dv[0][0] = e0.v[0]; dv[0][1] = e0.v[1]; dv[1][0] = e1.v[0]; dv[1][1] = e1.v[1];
        }
    }
    uint32_t dcode[2] = {0, 0}, lcode[2] = {0, 0};
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rr = q >> 1, cc = 2 * pr + (q & 1);
            const uint32_t oi = ci + rr, oj = cj + cc;
            uint32_t s = msb_code(iabs_u(xv[rr][cc]));
            if (d > 1 && 2 * oi + 1 < h && 2 * oj + 1 < w) {
                const uint32_t dc = dv[rr][cc];
                s = max(s, dc);
                lcode[pr] = max(lcode[pr], dc);
            }
            dcode[pr] = max(dcode[pr], s);
        }
    }
    if (v0) { dm[i * w + j0] = (uint8_t)dcode[0]; lm[i * w + j0] = (uint8_t)lcode[0]; }
    if (v1) { dm[i * w + j1] = (uint8_t)dcode[1]; lm[i * w + j1] = (uint8_t)lcode[1]; }
}

// Depth 1 and depth 2 in one pass.  Thread = one node q = (qi, qj) with 4*qi+1 < h (every node with offspring whose
// offspring have offspring is such a q or lies above one): its 2x2 block of children o and 4x4 block of grandchildren
// are loaded first -- four 16-byte rows and two 8-byte rows, and consecutive lanes read consecutive memory, so one
// wave-instruction covers 1 KB contiguously (a two-nodes-per-thread form, which reads every line with two instructions,
// measured 5 % slower).  Then
//   * every child o of depth exactly 1 (offspring, no grand-offspring) gets D(o) = max code of its four offspring
//     (L(o) is empty and is never looked up: a type-B entry needs grand-offspring, encoder_decoder.rs:7-12, :258);
//   * q itself, when its depth is exactly 2, gets D(q) = max over its offspring of max(code, D) and L(q) = max D.
// Lane pairs exchange their byte results with one cross-lane read each and the even lane stores for both (one 4-byte
// store per offspring row, 2-byte stores for the pair of q).  Nodes of the root block are left to k_pyr_ll.
// 1-D grid, XCD-contiguous tile order; tiles: (ceil(cols/64), ceil(rows/4), B*c), block (64,4).
__global__ __launch_bounds__(256) void k_pyr_12(PyrArgs a, uint32_t gx, uint32_t gy, uint32_t gz) {
    const Geom g = a.g;
    uint32_t bx, by, bz;
    {
        const uint32_t nt = gx * gy * gz, L = blockIdx.x;
        const uint32_t q = nt >> 3, r = nt & 7u, x = L & 7u, jj = L >> 3;
        const uint32_t T = x * q + (x < r ? x : r) + jj;
        bx = T % gx;
        const uint32_t t2 = T / gx;
        by = t2 % gy;
        bz = t2 / gy;
    }
    const uint32_t qj = bx * 64 + threadIdx.x, qi = by * 4 + threadIdx.y;
    const uint32_t h = (uint32_t)g.h, w = (uint32_t)g.w, lh = (uint32_t)g.ll_h, lw = (uint32_t)g.ll_w;
    const size_t base = (size_t)bz * g.hw;
    const int32_t *__restrict__ x = a.x + base;
    uint8_t *__restrict__ dm = a.dmsb + base;
    uint8_t *__restrict__ lm = a.lmsb + base;
    const uint32_t ci = 2 * qi, cj = 2 * qj, gi = 4 * qi, gj = 4 * qj;
    const bool dom = 4 * qi + 1 < h && (uint64_t)4 * qj + 1 < w;  // q has grand-offspring rows and columns at all
    // which of the four offspring have depth exactly 1, and is q of depth exactly 2
    bool wr[2][2], any = false;
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t r = ci + rr, c = cj + q;
            const bool has = 2 * r + 1 < h && 2 * c + 1 < w;
            const bool deep = (uint64_t)4 * r + 1 < h && (uint64_t)4 * c + 1 < w;
            wr[rr][q] = dom && has && !deep && !(r < lh && c < lw);
            any = any || wr[rr][q];
        }
    const bool wq = dom && !((uint64_t)8 * qi + 1 < h && (uint64_t)8 * qj + 1 < w) && !(qi < lh && qj < lw);
    any = any || wq;
    int32_t xc[2][2] = {{0, 0}, {0, 0}}, xg[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (any) {
        if (gi + 3 < h && gj + 3 < w) {
            const i2u c0 = *reinterpret_cast<const i2u *>(x + (size_t)ci * w + cj), c1 = *reinterpret_cast<const i2u *>(x + (size_t)(ci + 1) * w + cj);
            i4u r[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) r[rr] = *reinterpret_cast<const i4u *>(x + (size_t)(gi + rr) * w + gj);
            xc[0][0] = c0.v[0]; xc[0][1] = c0.v[1]; xc[1][0] = c1.v[0]; xc[1][1] = c1.v[1];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int q = 0; q < 4; q++) xg[rr][q] = r[rr].v[q];
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int q = 0; q < 2; q++) xc[rr][q] = (ci + rr < h && cj + q < w) ? x[(size_t)(ci + rr) * w + cj + q] : 0;
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int q = 0; q < 4; q++) xg[rr][q] = (gi + rr < h && gj + q < w) ? x[(size_t)(gi + rr) * w + gj + q] : 0;
        }
    }
    uint32_t dch[2][2], dq = 0, lq = 0;
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t r = ci + rr, c = cj + q;
            const bool has = 2 * r + 1 < h && 2 * c + 1 < w;
            uint32_t d = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) d = max(d, msb_code(iabs_u(xg[2 * rr + (u >> 1)][2 * q + (u & 1)])));
            dch[rr][q] = has ? d : 0u;
            dq = max(dq, max(msb_code(iabs_u(xc[rr][q])), dch[rr][q]));
            lq = max(lq, dch[rr][q]);
        }
    // pack: this lane's bytes | partner's bytes (lanes 2k, 2k+1 own adjacent columns)
    const uint32_t mine0 = dch[0][0] | (dch[0][1] << 8), mine1 = dch[1][0] | (dch[1][1] << 8);
    const uint32_t mflags = (wr[0][0] ? 1u : 0u) | (wr[0][1] ? 2u : 0u) | (wr[1][0] ? 4u : 0u) | (wr[1][1] ? 8u : 0u) | (wq ? 16u : 0u);
    const uint32_t mineq = dq | (lq << 8) | (mflags << 16);
    const uint32_t oth0 = (uint32_t)__shfl_xor((int)mine0, 1), oth1 = (uint32_t)__shfl_xor((int)mine1, 1);
    const uint32_t othq = (uint32_t)__shfl_xor((int)mineq, 1);
    if ((threadIdx.x & 1u) == 0) {
        const uint32_t oflags = othq >> 16;
        const uint32_t row[2] = {mine0 | (oth0 << 16), mine1 | (oth1 << 16)};
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t f4 = ((mflags >> (2 * rr)) & 3u) | (((oflags >> (2 * rr)) & 3u) << 2);
            uint8_t *p = dm + (size_t)(ci + rr) * w + cj;
            if (f4 == 15u) {
                b4u pk;
                pk.v[0] = (uint8_t)row[rr]; pk.v[1] = (uint8_t)(row[rr] >> 8); pk.v[2] = (uint8_t)(row[rr] >> 16); pk.v[3] = (uint8_t)(row[rr] >> 24);
                *reinterpret_cast<b4u *>(p) = pk;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((f4 >> q) & 1u) p[q] = (uint8_t)(row[rr] >> (8 * q));
            }
        }
        const bool w0 = (mflags >> 4) & 1u, w1 = (oflags >> 4) & 1u;
        const size_t qo = (size_t)qi * w + qj;
        if (w0 && w1) {
            b2u pd, pl;
            pd.v[0] = (uint8_t)dq; pd.v[1] = (uint8_t)othq;
            pl.v[0] = (uint8_t)lq; pl.v[1] = (uint8_t)(othq >> 8);
            *reinterpret_cast<b2u *>(dm + qo) = pd;
            *reinterpret_cast<b2u *>(lm + qo) = pl;
        } else {
            if (w0) { dm[qo] = (uint8_t)dq; lm[qo] = (uint8_t)lq; }
            if (w1) { dm[qo + 1] = (uint8_t)othq; lm[qo + 1] = (uint8_t)(othq >> 8); }
        }
    }
}

// Root block (encoder_decoder.rs:44-63).  grid: (ceil(ll_w*ll_h/256), 1, B*c)
__global__ __launch_bounds__(256) void k_pyr_ll(PyrArgs a) {
    const Geom g = a.g;
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint32_t)(g.ll_h * g.ll_w)) return;
    uint32_t i = t / (uint32_t)g.ll_w, j = t - i * (uint32_t)g.ll_w;
    if (((i | j) & 1u) == 0) return;  // both even: no offspring
    const uint32_t h = (uint32_t)g.h, w = (uint32_t)g.w;
    const size_t base = (size_t)blockIdx.z * g.hw;
    const int32_t *x = a.x + base;
    uint8_t *dm = a.dmsb + base;
    uint8_t *lm = a.lmsb + base;
    uint32_t ri = (i & 1u) * (uint32_t)g.ll_h + (i & ~1u);
    uint32_t rj = (j & 1u) * (uint32_t)g.ll_w + (j & ~1u);
    uint32_t dcode = 0, lcode = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t oi = ri + (q >> 1), oj = rj + (q & 1);
        uint32_t o = oi * w + oj;
        uint32_t s = msb_code(iabs_u(x[o]));
        if (2 * oi + 1 < h && 2 * oj + 1 < w) {
            uint32_t dc = dm[o];
            s = max(s, dc);
            lcode = max(lcode, dc);
        }
        dcode = max(dcode, s);
    }
    dm[i * w + j] = (uint8_t)dcode;
    lm[i * w + j] = (uint8_t)lcode;
}

__global__ void k_nbits_to_nbytes(const uint64_t *nbits, int B, uint64_t *nbytes) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) nbytes[t] = (nbits[t] + 7) >> 3;
}
extern "C" int spiht_launch_nbits_to_nbytes(const uint64_t *d_nbits, int B, uint64_t *d_nbytes, hipStream_t st) {
    hipLaunchKernelGGL(k_nbits_to_nbytes, dim3((B + 255) / 256), dim3(256), 0, st, d_nbits, B, d_nbytes);
    return (int)hipGetLastError();
}

// ---- host launchers -------------------------------------------------------------------------

extern "C" int spiht_launch_absmax(const int32_t *d_x, int B, uint32_t n, uint32_t *d_maxabs, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_maxabs, 0, sizeof(uint32_t) * (size_t)B, st);
    if (e != hipSuccess) return (int)e;
    uint32_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_absmax, dim3(blocks, B), dim3(256), 0, st, d_x, n, d_maxabs);
    return (int)hipGetLastError();
}

// number of index-doubling rounds needed: max depth of a node outside the root block
extern "C" int spiht_pyr_rounds(const Geom *g) {
    int best = 0;
    // the deepest non-root nodes are (0, ll_w) and (ll_h, 0)
    for (int d = 1; d < 31; d++) {
        bool a = ((uint64_t)g->ll_w << d) + 1 < (uint64_t)g->w;   // node (0, ll_w): i = 0 always passes
        bool b = ((uint64_t)g->ll_h << d) + 1 < (uint64_t)g->h;   // node (ll_h, 0)
        if (a || b) best = d; else break;
    }
    return best;
}

extern "C" int spiht_launch_pyramid(const Geom *g, int B, const int32_t *d_x, uint8_t *d_dmsb, uint8_t *d_lmsb, hipStream_t st) {
    PyrArgs a;
    a.g = *g;
    a.B = B;
    a.x = d_x;
    a.dmsb = d_dmsb;
    a.lmsb = d_lmsb;
    a.maxabs = nullptr;
    int rounds = spiht_pyr_rounds(g);
    if (rounds >= 1) {  // depth 1 and 2 together
        const uint32_t rows = (uint32_t)(((uint64_t)g->h - 1 + 3) >> 2);   // qi with 4*qi+1 < h
        const uint32_t pairs = (uint32_t)(((uint64_t)g->w - 1 + 7) >> 3);  // t with 8*t+1 < w
        if (rows && pairs) {
            const uint32_t cols = 2 * pairs;  // an even number of columns: lane pairs stay together
            const uint32_t gx = (cols + 63) / 64, gy = (rows + 3) / 4, gz = (uint32_t)(B * g->c);
            hipLaunchKernelGGL(k_pyr_12, dim3(gx * gy * gz), dim3(64, 4), 0, st, a, gx, gy, gz);
        }
    }
    for (int d = 3; d <= rounds; d++) {
        a.round = d;
        uint32_t gi = (uint32_t)(((uint64_t)g->h - 1 + (1ull << d) - 1) >> d);
        uint32_t gj = (uint32_t)(((uint64_t)g->w - 1 + (1ull << d) - 1) >> d);
        if (gi == 0 || gj == 0) continue;
        const uint32_t npairs = (gj + 1) / 2;
        const uint32_t gx = (npairs + 63) / 64, gy = (gi + 3) / 4, gz = (uint32_t)(B * g->c);
        hipLaunchKernelGGL(k_pyr_round, dim3(gx * gy * gz), dim3(64, 4), 0, st, a, gx, gy, gz);
    }
    a.round = 0;
    dim3 grid((uint32_t)((g->ll_h * g->ll_w + 255) / 256), 1, (uint32_t)(B * g->c));
    hipLaunchKernelGGL(k_pyr_ll, grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

// One wavefront that waits until a counter in device memory has reached `target` (arithmetic modulo 2^32) -- or gives up
// after `ticks` shader clocks: what is queued behind it on its stream starts when the workgroups that raise the counter are
// on the CUs (spiht_ctx_wait_resident; csrc/pipeline.cpp launches the list decoder behind the persistent workgroups of the
// inverse transform's level 1 that way).
__global__ __launch_bounds__(64) void k_gate(const uint32_t *counter, uint32_t target, uint64_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while ((int32_t)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 &&
           __builtin_amdgcn_s_memtime() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
}
extern "C" int spiht_launch_gate(const uint32_t *counter, uint32_t target, uint64_t ticks, hipStream_t st) {
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, counter, target, ticks);
    return (int)hipGetLastError();
}

#ifdef SPIHT_DIAG
// diagnostic (tools/corun.py): workgroups that occupy the CUs for a given number of clock ticks without touching memory
__global__ __launch_bounds__(512) void k_spin(uint64_t ticks, uint32_t lds_words, uint32_t *sink) {
    extern __shared__ uint32_t dyn[];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        acc++;
    }
    if (lds_words) dyn[threadIdx.x % lds_words] = acc;
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
// ... and neighbours that are NOT idle, one kind of activity each (tools/corun_kinds.py): what of a list-coding workgroup's
// doings costs the transform kernels beside it?  mode 1: wavefront 0 runs a dependent scalar chain (the sequencer's kind of
// work), the others sleep; 2: every wavefront polls LDS between short sleeps (the workers'); 3: wavefront 0 runs a dependent
// vector chain; 4: wavefront 0 scalar chain + wavefront 1 vector / LDS work (sequencer + helper); 5: every wavefront issues
// scalar work (no sleeps at all).
__device__ __forceinline__ void spin_kind_body(uint64_t ticks, uint32_t lds_words, uint32_t *sink, int mode) {
    extern __shared__ uint32_t dyn[];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t acc = threadIdx.x, s = (uint32_t)ticks | 1u, sb = 5u;  // (s, sb: wave-uniform, kept in SGPRs by the asm below)
    const bool salu = (mode == 1 && wave == 0) || (mode == 4 && wave == 0) || mode == 5;
    const bool valu = (mode == 3 && wave == 0) || (mode == 4 && wave == 1);
    const bool poll = mode == 2;
    for (;;) {
        if (salu) {
            uint32_t s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s), b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sb);
#pragma unroll
            for (int k = 0; k < 64; k++) {  // dependent scalar chain
                asm volatile("s_lshr_b32 %0, %0, 1\n\ts_add_i32 %0, %0, %1\n\ts_ff1_i32_b32 %1, %0\n\ts_add_i32 %1, %1, 3"
                             : "+s"(s1), "+s"(b1) : : "scc");
            }
            s = s1; sb = b1;
        } else if (valu) {
#pragma unroll
            for (int k = 0; k < 64; k++) acc = acc * 1664525u + (lds_words ? dyn[(acc >> 7) % lds_words] : 1013904223u);
        } else if (poll) {
            if (lds_words) acc += __hip_atomic_load(&dyn[threadIdx.x % lds_words], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_sleep(1);
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
        if (__builtin_amdgcn_s_memtime() - t0 >= ticks) break;
    }
    if (lds_words) dyn[threadIdx.x % lds_words] = acc + s + sb;
    if (acc == 0xFFFFFFFFu && s == 77u) *sink = acc;
}
__global__ __launch_bounds__(512) void k_spin_kind(uint64_t ticks, uint32_t lds_words, uint32_t *sink, int mode) {
    spin_kind_body(ticks, lds_words, sink, mode);
}
// ... with the decoder's register footprint as well (96 VGPRs per thread)
__global__ __launch_bounds__(512) void k_spin_kind96(uint64_t ticks, uint32_t lds_words, uint32_t *sink, int mode) {
    asm volatile("v_mov_b32 v95, 0" ::: "v95");  // (what a kernel is given is the highest register it names)
    spin_kind_body(ticks, lds_words, sink, mode);
}
extern "C" int spiht_launch_spin_kind(int blocks, int threads, uint64_t ticks, uint32_t lds_bytes, uint32_t *sink, int mode, hipStream_t st) {
    if (mode >= 16) hipLaunchKernelGGL(k_spin_kind96, dim3(blocks), dim3(threads), lds_bytes, st, ticks, lds_bytes / 4, sink, mode - 16);
    else hipLaunchKernelGGL(k_spin_kind, dim3(blocks), dim3(threads), lds_bytes, st, ticks, lds_bytes / 4, sink, mode);
    return (int)hipGetLastError();
}
extern "C" int spiht_launch_spin(int blocks, int threads, uint64_t ticks, uint32_t lds_bytes, uint32_t *sink, hipStream_t st) {
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), lds_bytes, st, ticks, lds_bytes / 4, sink);
    return (int)hipGetLastError();
}
#endif  // SPIHT_DIAG
