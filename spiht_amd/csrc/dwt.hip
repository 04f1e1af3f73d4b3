// 2-D separable DWT / inverse DWT for the SPIHT image path (gfx950), float64.
//
// Replaces the PyWavelets calls of the reference wrapper
//   encode: pywt.wavedec2 -> coeffs_to_array -> channel_mults*arr -> quantize   (spiht_wrapper.py:163-172)
//   decode: rec/channel_mults -> dequantize -> array_to_coeffs -> waverec2       (spiht_wrapper.py:259-276)
// One launch per decomposition level, both axes fused:
//   forward: a workgroup owns a TH x TW tile of output positions of all four sub-bands.  The axis -2
//            filter runs straight from global memory: thread = input column, and the whole column segment
//            the tile needs is loaded into registers FIRST (2*TH+F-2 independent 8-byte loads per lane in
//            flight, consecutive lanes -> consecutive addresses), then filtered; the low/high intermediates
//            go to LDS split by column parity so that the axis -1 filter reads them with unit stride.  LL
//            is written as float64 (input of the next level); the three detail bands are written already
//            quantised (int32, truncation toward zero) into their place in the zero-padded Mallat array --
//            coeffs_to_array and the two quantise passes of the wrapper cost no extra HBM traffic -- and
//            max|coefficient| (encoder_decoder.rs:165) is folded into the same pass.
//   inverse: band tiles are dequantised on load ((rec / m_k) / q as the wrapper does) into LDS; thread =
//            output column: axis -1 synthesis of one band row at a time feeds a register window of the
//            last F/2 rows, from which the axis -2 synthesis emits two output rows -- no LDS round trip for
//            the intermediate, stores are 512 B per wave.
// Arithmetic follows the published pywt definitions in the same summation order as oracle/dwt_oracle.c
// (ascending tap / band index, separate multiply and add: this file is compiled with -ffp-contract=off),
// so GPU and oracle agree bit for bit and both agree with pywt to a few ulp.
// HBM-bound: per level, 8 B per input sample + 8 B (LL) + 3*4 B (details) per output position.
#include "common.h"
#include <stdlib.h>
#include <string.h>

#ifndef DW_TH
#define DW_TH 12      // output rows per tile.  Level 1 of 256 1080p images: 12 rows 3.8-3.95 ms, 16 rows 4.05-4.2, 14: 4.1,
#endif                // 10: 4.4, 20: 4.5, 24: 4.2 (25.9 KB of LDS per workgroup at 12 rows: six workgroups per CU)
#define DW32_TH 16    // ... of the single-precision kernel (12 rows: 3.9 instead of 2.85 ms)
#ifndef DW_TW
#define DW_TW 64      // output cols per tile
#endif
#define DW_BLOCK 256
#ifndef DWF_BLOCK
#define DWF_BLOCK 192  // threads of a forward-transform workgroup (k_dwt_level); 132 of them hold an input column each.  Three
                       // wavefronts instead of four: beside a list-decoder workgroup (192 of a SIMD's 512 registers) FIVE of these
                       // workgroups fit a CU instead of four -- level 1 inside the pipelined step 6.3 instead of 6.55 ms, the step
                       // 16.4 instead of 16.7; alone 4.2 instead of 4.05 ms (round 4)
#endif

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Remap the linear block id so
// that each XCD walks one contiguous range of tiles: neighbouring tiles (shared halo columns/rows on the read
// side, shared partial cache lines on the write side) then meet in the same L2.  Speed only, never correctness.
__device__ __forceinline__ void xcd_tile_at(uint32_t L, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t &bx, uint32_t &by,
                                            uint32_t &bz) {
    const uint32_t nt = gx * gy * gz;
    const uint32_t q = nt >> 3, r = nt & 7u, x = L & 7u, j = L >> 3;
    const uint32_t T = x * q + (x < r ? x : r) + j;
    bx = T % gx;
    const uint32_t t2 = T / gx;
    by = t2 % gy;
    bz = t2 / gy;
}

__device__ __forceinline__ void xcd_tile(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t &bx, uint32_t &by, uint32_t &bz) {
    xcd_tile_at(blockIdx.x, gx, gy, gz, bx, by, bz);
}

// max |coefficient| of an image, raised by every tile of the image: an atomic on one word, and read-modify-writes of one
// memory line are served one after the other (11.5 ns each, measured: with one atomic per wavefront a 16-image launch
// of level 1 took 1.6 ms for its 141 000 atomics, six times what its bytes take, and 256 images were as much
// atomic-bound as HBM-bound).  So: one atomic per workgroup (the wavefronts meet in an LDS word first), and none when
// the word already holds as much (a cached look; measured against no look and a look the atomics' coherence point answers,
// DESIGN.md 6; a stale answer only costs a spare atomic).  `s_m` must have been zeroed before an earlier barrier.
// The cached look relies on this: the word is only ever RAISED between two zero-fills (hipMemsetAsync in front of every
// transform, api.cpp), and a zero-fill is a kernel / copy of its own on the same stream -- the caches that could hold the
// word (the CU's vector L1, and the per-XCD L2 for lines another XCD wrote) are invalidated at the kernel boundary in
// front of this launch, so a look can return an OLD value of this launch's epoch (lower: a spare atomic) but never one
// from before the zero-fill (higher: a lost maximum).  tests/test_gpu_dwt.py::test_maxabs_small_batch_after_large_batch
// runs exactly the case that would show it: large magnitudes, then small ones, same context and buffer.
__device__ __forceinline__ void block_raise_max(uint32_t *p, uint32_t v, uint32_t *s_m) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(s_m, v);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m = *s_m;
        if (m > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(p, m);
    }
}

// The same for kernels whose threads are a flat index over planes (a wavefront may straddle two images): one atomic per
// wavefront when all its lanes belong to one image, per lane otherwise.  img < 0: the lane has nothing to report.  All lanes
// of the wavefront call it.  (One atomic per THREAD on one word per image made the two-pass forward level 17 ms per 1080p
// picture: six million atomics on the same address.)
__device__ __forceinline__ void wave_raise_max(uint32_t *maxabs, int img, uint32_t v) {
    const int any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(img, __builtin_ctzll(__ballot(img >= 0) | (1ull << 63))));
    if (__ballot(img >= 0) == 0) return;
    if (__all(img < 0 || img == any)) {
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
        if ((threadIdx.x & 63) == 0 && v > __hip_atomic_load(&maxabs[any], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&maxabs[any], v);
    } else if (img >= 0 && v) {
        atomicMax(&maxabs[img], v);
    }
}

__device__ __forceinline__ int ext_index(int i, int N, int mode) {
    if (i >= 0 && i < N) return i;
    switch (mode) {
    case 0: {  // reflect (whole-sample symmetric)
        if (N == 1) return 0;
        int P = 2 * (N - 1);
        int m = i % P; if (m < 0) m += P;
        return m < N ? m : P - m;
    }
    case 1: {  // symmetric (half-sample)
        int P = 2 * N;
        int m = i % P; if (m < 0) m += P;
        return m < N ? m : P - 1 - m;
    }
    case 2: {  // periodic
        int m = i % N; if (m < 0) m += N;
        return m;
    }
    case 4: return i < 0 ? 0 : N - 1;  // constant
    default: return -1;                 // zero
    }
}

__device__ __forceinline__ int32_t quant(double v, double m, double q, bool has_m) {
    if (has_m) v = m * v;
    v = v * q;
    return (int32_t)v;
}
__device__ __forceinline__ uint32_t iabs_u(int32_t x) { return (uint32_t)(x < 0 ? -x : x); }
__device__ __forceinline__ double dequant(int32_t r, double m, double q, bool has_m) {
    double v = (double)r;
    if (has_m) v = v / m;
    return v / q;
}

// grid: (ceil(out_w/TW), ceil(out_h/TH), planes).  LOM / HIM: bit j set = tap j of dec_lo / dec_hi is non-zero;
// a zero tap contributes exactly nothing (0*x added to the running sum), so skipping it changes no bit and
// removes a third (bior2.2) to a fifth of the float64 arithmetic.
// One tile of k_dwt_level.
template <int F, uint32_t LOM, uint32_t HIM, int PS, int NR>
__device__ __forceinline__ void dwt_tile(const DwtKArgs &a, double (&s_lo)[2][PS], double (&s_hi)[2][PS], uint32_t tbx, uint32_t tby,
                                         uint32_t tbz) {
    constexpr int NC = 2 * DW_TW + F - 2;  // input columns needed by the tile
    constexpr int HC = (NC + 1) / 2;       // columns per parity plane
    static_assert(NC <= DWF_BLOCK, "one thread per input column");
    static_assert(NR == 2 * DW_TH + F - 2, "input rows needed");
    constexpr int RS = HC + 1;
    const int plane = (int)tbz;
    const int oh0 = (int)tby * DW_TH, ow0 = (int)tbx * DW_TW;
    const double *__restrict__ in = a.in + (size_t)plane * a.in_h * a.in_w;
    const int tid = threadIdx.x;

    // input row needed for output row o, tap j: 2*o + 1 - j ; first needed row r0 = 2*oh0 + 1 - (F-1)
    const int r0 = 2 * oh0 + 2 - F, c0 = 2 * ow0 + 2 - F;
    // the rows' element offsets (row index * in_w; -1: a row of zeros), worked out once per tile by NR threads
    __shared__ long long s_off[NR];
    if (tid < NR) {
        const int gr = ext_index(r0 + tid, a.in_h, a.mode);
        s_off[tid] = gr < 0 ? -1ll : (long long)gr * a.in_w;
    }
    __shared__ uint32_t s_amax;
    if (tid == 0) s_amax = 0;
    __syncthreads();

    // ---- axis -2: thread <-> input column; all loads first, then the filter ----
    if (tid < NC) {
        const int gc = ext_index(c0 + tid, a.in_w, a.mode);
        double x[NR];
        // Nothing of the address arithmetic on the scalar unit: a list-coding workgroup on the same CU keeps that unit busy
        // (its sequencer wavefront issues a dependent scalar instruction whenever it can, and it is the older wavefront),
        // and 28 rows x a dozen scalar instructions per tile were what this kernel lost beside it (DESIGN.md 6, round 3:
        // one scalar-busy wavefront per CU and nothing else cost this kernel 68 %).  The row offsets come out of LDS
        // through an index the compiler cannot prove uniform, so that they stay in vector registers.
        int lz = 0;
        asm volatile("" : "+v"(lz));
        if (a.mode != 3) {  // not zero padding: every index is inside the picture
#pragma unroll
            for (int r = 0; r < NR; r++) x[r] = in[s_off[r + lz] + gc];
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const long long off = s_off[r + lz];
                const bool ok = gc >= 0 && off >= 0;
                const double v = in[ok ? off + gc : 0];
                x[r] = ok ? v : 0.0;
            }
        }
        const int par = tid & 1, hc = tid >> 1;
#pragma unroll
        for (int o = 0; o < DW_TH; o++) {
            // out[o] = sum_j f[j] * x~[2(oh0+o)+1-j];  x~[2(oh0+o)+1-j] = x[2o + F-1-j]
            double sl = 0.0, shh = 0.0;
#pragma unroll
            for (int j = 0; j < F; j++) {
                if ((LOM >> j) & 1u) sl += a.lo[j] * x[2 * o + F - 1 - j];
                if ((HIM >> j) & 1u) shh += a.hi[j] * x[2 * o + F - 1 - j];
            }
            s_lo[par][o * RS + hc] = sl;
            s_hi[par][o * RS + hc] = shh;
        }
    }
    __syncthreads();

    // ---- axis -1 from LDS; 4 sub-bands per output position ----
    const int k = plane % a.c;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    int32_t *__restrict__ co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
    double *__restrict__ llo = a.last ? nullptr : a.ll_out + (size_t)plane * a.out_h * a.out_w;
    uint32_t amax = 0;
    constexpr int NU = (DW_TH * DW_TW + DWF_BLOCK - 1) / DWF_BLOCK;  // output positions per thread
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int p = tid + u * DWF_BLOCK;
        const int o = p / DW_TW, wcol = p % DW_TW;
        const int oh = oh0 + o, ow = ow0 + wcol;
        if (o >= DW_TH || oh >= a.out_h || ow >= a.out_w) continue;
        // x~ index 2*ow+1-j  ->  tile column 2*wcol + F-1-j  ->  parity (F-1-j)&1, half-column wcol + (F-1-j)/2
        double aa = 0.0, ad = 0.0, da = 0.0, dd = 0.0;
#pragma unroll
        for (int j = 0; j < F; j++) {
            const double vl = s_lo[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
            const double vh = s_hi[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
            if ((LOM >> j) & 1u) { aa += a.lo[j] * vl; da += a.lo[j] * vh; }
            if ((HIM >> j) & 1u) { ad += a.hi[j] * vl; dd += a.hi[j] * vh; }
        }
        const int32_t qad = quant(ad, mk, a.q, has_m), qda = quant(da, mk, a.q, has_m), qdd = quant(dd, mk, a.q, has_m);
        // the few outputs of the bottom / right overhang whose sum depends on PyWavelets' tap order are computed again
        // by k_dwt_edge, which overwrites them and accounts for their magnitude
        const bool mine = oh < a.ov_h && ow < a.ov_w;
        if (a.last) {
            const int32_t qaa = quant(aa, mk, a.q, has_m);
            co[(size_t)oh * a.enc_w + ow] = qaa;
            if (mine) amax = max(amax, iabs_u(qaa));
        } else {
            llo[(size_t)oh * a.out_w + ow] = aa;
        }
        co[(size_t)oh * a.enc_w + a.off_w + ow] = qad;                 // 'ad' top-right
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = qda;               // 'da' bottom-left
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = qdd;     // 'dd' bottom-right
        if (mine) amax = max(amax, max(iabs_u(qad), max(iabs_u(qda), iabs_u(qdd))));
    }
    if (a.maxabs != nullptr) {
        block_raise_max(&a.maxabs[plane / a.c], amax, &s_amax);
    }
}

template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(DWF_BLOCK) void k_dwt_level(DwtKArgs a) {
    constexpr int NC = 2 * DW_TW + F - 2, NR = 2 * DW_TH + F - 2, HC = (NC + 1) / 2;
    // two column-parity planes; the padding makes the plane stride an odd multiple of 16 banks, so the even and odd
    // lanes of one ds_write_b64 lane group land on different banks
    constexpr int RS = HC + 1, PS = DW_TH * RS + (24 - (DW_TH * RS) % 16) % 16;  // PS % 16 == 8
    __shared__ double s_lo[2][PS];
    __shared__ double s_hi[2][PS];
    uint32_t tbx, tby, tbz;
    // the tile's coordinates worked out in vector registers (every lane the same values): the two divisions and everything
    // that follows from them -- plane, origin, the base pointers -- then cost the scalar unit nothing (see dwt_tile)
    uint32_t lin = blockIdx.x;
    asm volatile("" : "+v"(lin));
    xcd_tile_at(lin, (a.out_w + DW_TW - 1) / DW_TW, (a.out_h + DW_TH - 1) / DW_TH, a.planes, tbx, tby, tbz);
    dwt_tile<F, LOM, HIM, PS, NR>(a, s_lo, s_hi, tbx, tby, tbz);
}

// ---- helpers of the persistent inverse-transform kernel (k_idwt_level_pf) -------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// a barrier that waits for LDS traffic only: __syncthreads() would wait for the global loads in flight as well
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define BUF_OOB 0x80000000u  // an offset no plane reaches (the launcher checks): dropped by the descriptor's range check
// Buffer descriptor of `bytes` bytes at p.  p and bytes are workgroup-uniform, but a value that went through a VALU
// division is a VGPR to the instruction selector, and a descriptor in VGPRs costs a waterfall loop per memory
// instruction: its words are read back as scalars here, once.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *p, uint32_t bytes) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
}

// ---- the bottom / right overhang of a float64 level, in PyWavelets' summation order -----------------------------
// pywt's downsampling_convolution adds the taps in ascending order -- except for the outputs that hang over the end
// of the input (jb = 2o+1-N >= 0, the last F/2 or so output rows and columns of a level): there the taps that read the
// signal extension come first, nearest first (tap jb down to 0), then the others ascending.  The sums differ in the
// last bits, and on 8-bit pictures with flat areas (coefficient x q exactly an integer) the truncating quantiser turns
// that into +-1 (tests/golden/blocky_pywt.npz).  k_dwt_level keeps its compile-time ascending order everywhere (the
// overhang code inside it -- as a block-uniform branch, a second instantiation for the edge tiles, one launch or two,
// a side stream -- cost the level 5 to 12 %: measured); this kernel then recomputes just the outputs whose order
// matters (ov_h / ov_w: the last row and column of a bior level), one thread each, straight from global memory, and
// overwrites them.  grid: (ceil(outputs / 256), planes).
template <int F>
__global__ __launch_bounds__(256) void k_dwt_edge(DwtKArgs a) {
    __shared__ double s_f[2][F];  // taps, indexed at run time below
    if (threadIdx.x < F) { s_f[0][threadIdx.x] = a.lo[threadIdx.x]; s_f[1][threadIdx.x] = a.hi[threadIdx.x]; }
    __shared__ uint32_t s_amax;
    if (threadIdx.x == 0) s_amax = 0;
    __syncthreads();
    const int nr = a.out_h - a.ov_h, nc = a.out_w - a.ov_w;        // overhang rows / columns
    const int nA = nr * a.out_w, total = nA + a.ov_h * nc;          // all columns of those rows + the rest of those columns
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y;
    uint32_t amax = 0;
    if (t < total) {
        int oh, ow;
        if (t < nA) { oh = a.ov_h + t / a.out_w; ow = t % a.out_w; }
        else { const int u = t - nA; oh = u / nc; ow = a.ov_w + u % nc; }
        const double *__restrict__ in = a.in + (size_t)plane * a.in_h * a.in_w;
        const int ir = 2 * oh + 1, ic = 2 * ow + 1;
        const int jbr = oh >= a.ov_h ? ir - a.in_h : -1, jbc = ow >= a.ov_w ? ic - a.in_w : -1;
        int gr[F];
#pragma unroll
        for (int r = 0; r < F; r++) gr[r] = ext_index(ir - r, a.in_h, a.mode);
        // Column by column in the order axis -1 asks for; per column the F samples are loaded first (independent loads),
        // then added in the order the row asks for -- a sample is picked by comparing indices, registers cannot be
        // indexed at run time.  (One load at a time, as a plain double loop does it, this kernel took 0.26 ms at
        // 256 x 1080p, 7 % of the level.)
        double aa = 0.0, ad = 0.0, da = 0.0, dd = 0.0;
#pragma unroll F <= 6 ? F : 1
        for (int s1 = 0; s1 < F; s1++) {
            const int j = s1 <= jbc ? jbc - s1 : s1;
            const int gc = ext_index(ic - j, a.in_w, a.mode);
            double xv[F];
#pragma unroll
            for (int r = 0; r < F; r++) xv[r] = (gc < 0 || gr[r] < 0) ? 0.0 : in[(size_t)gr[r] * a.in_w + gc];
            double tl = 0.0, th = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < F; s2++) {
                const int j2 = s2 <= jbr ? jbr - s2 : s2;
                double v = xv[0];
#pragma unroll
                for (int r = 1; r < F; r++) v = (j2 == r) ? xv[r] : v;
                tl += s_f[0][j2] * v;
                th += s_f[1][j2] * v;
            }
            aa += s_f[0][j] * tl; da += s_f[0][j] * th;
            ad += s_f[1][j] * tl; dd += s_f[1][j] * th;
        }
        const int k = plane % a.c;
        const bool has_m = a.mults != nullptr;
        const double mk = has_m ? a.mults[k] : 1.0;
        int32_t *__restrict__ co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
        const int32_t qad = quant(ad, mk, a.q, has_m), qda = quant(da, mk, a.q, has_m), qdd = quant(dd, mk, a.q, has_m);
        if (a.last) {
            const int32_t qaa = quant(aa, mk, a.q, has_m);
            co[(size_t)oh * a.enc_w + ow] = qaa;
            amax = iabs_u(qaa);
        } else {
            a.ll_out[(size_t)plane * a.out_h * a.out_w + (size_t)oh * a.out_w + ow] = aa;
        }
        co[(size_t)oh * a.enc_w + a.off_w + ow] = qad;
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = qda;
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = qdd;
        amax = max(amax, max(iabs_u(qad), max(iabs_u(qda), iabs_u(qdd))));
    }
    if (a.maxabs != nullptr) {
        block_raise_max(&a.maxabs[plane / a.c], amax, &s_amax);
    }
}

// ---- single-precision forward level ------------------------------------------------------------------------------
// PyWavelets transforms float32 (and float16) pixels in float32 with float copies of the filters, and the wrapper
// quantises the float32 array in float32 (in float64 once per-channel scales are applied).  In float32 the order of
// the additions decides quantised coefficients, so this kernel follows pywt's (convolution.template.c,
// downsampling_convolution): ascending taps, except for outputs that hang over the right / bottom end (2o+1 >= N),
// where the taps reading the extension come first, nearest first (tap index 2o+1-N down to 0), then the rest
// ascending.  Same tiling as k_dwt_level; the overhang outputs (the last (F-1)/2 rows and columns of a level) take
// a slower path with run-time tap order.  Level inputs shorter than the filter are refused by the host (pywt runs
// yet another loop for them).
__device__ __forceinline__ int32_t quant_f32(float v, double m, double q, float qf, bool has_m) {
    if (has_m) return (int32_t)((m * (double)v) * q);  // channel_mults[:,None,None] * arr is float64 (wrapper:167-170)
    return (int32_t)(v * qf);                          // arr * q_scale stays float32 (wrapper:9-11)
}

template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(DW_BLOCK) void k_dwt_level_f32(DwtKArgs a) {
    constexpr int NC = 2 * DW_TW + F - 2, NR = 2 * DW32_TH + F - 2, HC = (NC + 1) / 2;
    static_assert(NC <= DW_BLOCK, "one thread per input column");
    constexpr int RS = HC + 1, PS = DW32_TH * RS + 8;
    __shared__ float s_lo[2][PS];
    __shared__ float s_hi[2][PS];
    __shared__ int s_row[NR];
    __shared__ float s_flo[F], s_fhi[F];  // float copies of the filters, for the run-time-ordered overhang sums
    uint32_t tbx, tby, tbz;
    xcd_tile((a.out_w + DW_TW - 1) / DW_TW, (a.out_h + DW32_TH - 1) / DW32_TH, a.planes, tbx, tby, tbz);
    const int plane = (int)tbz;
    const int oh0 = (int)tby * DW32_TH, ow0 = (int)tbx * DW_TW;
    const float *__restrict__ in = reinterpret_cast<const float *>(a.in) + (size_t)plane * a.in_h * a.in_w;
    const int tid = threadIdx.x;
    float flo[F], fhi[F];  // (PyWavelets' single-precision filters: not always the doubles rounded -- the coiflets)
#pragma unroll
    for (int j = 0; j < F; j++) { flo[j] = a.lo_f[j]; fhi[j] = a.hi_f[j]; }

    const int r0 = 2 * oh0 + 2 - F, c0 = 2 * ow0 + 2 - F;
    if (tid < NR) s_row[tid] = ext_index(r0 + tid, a.in_h, a.mode);
    if (tid < F) { s_flo[tid] = a.lo_f[tid]; s_fhi[tid] = a.hi_f[tid]; }
    __shared__ uint32_t s_amax;
    if (tid == 0) s_amax = 0;
    __syncthreads();

    // ---- axis -2 ----
    if (tid < NC) {
        const int gc = ext_index(c0 + tid, a.in_w, a.mode);
        float x[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int gr = s_row[r];
            x[r] = (gc < 0 || gr < 0) ? 0.0f : in[(size_t)gr * a.in_w + gc];
        }
        const int par = tid & 1, hc = tid >> 1;
#pragma unroll
        for (int o = 0; o < DW32_TH; o++) {
            float sl = 0.0f, shh = 0.0f;
            const int i = 2 * (oh0 + o) + 1;
            if (i < a.in_h || a.mode == 4) {  // (constant-edge mode: pywt keeps ascending order over the overhang too)
#pragma unroll
                for (int j = 0; j < F; j++) {
                    if ((LOM >> j) & 1u) sl += flo[j] * x[2 * o + F - 1 - j];
                    if ((HIM >> j) & 1u) shh += fhi[j] * x[2 * o + F - 1 - j];
                }
            } else if (oh0 + o < a.out_h) {  // overhang: extension taps first, nearest first (reloaded: run-time order)
                const int jb = i - a.in_h;
                for (int s = 0; s < F; s++) {
                    const int j = s <= jb ? jb - s : s;
                    const int gr = s_row[2 * o + F - 1 - j];
                    const float v = (gc < 0 || gr < 0) ? 0.0f : in[(size_t)gr * a.in_w + gc];
                    sl += s_flo[j] * v;
                    shh += s_fhi[j] * v;
                }
            }
            s_lo[par][o * RS + hc] = sl;
            s_hi[par][o * RS + hc] = shh;
        }
    }
    __syncthreads();

    // ---- axis -1 from LDS ----
    const int k = plane % a.c;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    const float qf = (float)a.q;
    int32_t *__restrict__ co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
    float *__restrict__ llo = a.last ? nullptr : reinterpret_cast<float *>(a.ll_out) + (size_t)plane * a.out_h * a.out_w;
    uint32_t amax = 0;
#pragma unroll
    for (int u = 0; u < DW32_TH * DW_TW / DW_BLOCK; u++) {
        const int p = tid + u * DW_BLOCK;
        const int o = p / DW_TW, wcol = p % DW_TW;
        const int oh = oh0 + o, ow = ow0 + wcol;
        if (oh >= a.out_h || ow >= a.out_w) continue;
        float aa = 0.0f, ad = 0.0f, da = 0.0f, dd = 0.0f;
        const int i = 2 * ow + 1;
        if (i < a.in_w || a.mode == 4) {
#pragma unroll
            for (int j = 0; j < F; j++) {
                const float vl = s_lo[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                const float vh = s_hi[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                if ((LOM >> j) & 1u) { aa += flo[j] * vl; da += flo[j] * vh; }
                if ((HIM >> j) & 1u) { ad += fhi[j] * vl; dd += fhi[j] * vh; }
            }
        } else {
            const int jb = i - a.in_w;
            for (int s = 0; s < F; s++) {
                const int j = s <= jb ? jb - s : s;
                const float vl = s_lo[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                const float vh = s_hi[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                aa += s_flo[j] * vl; da += s_flo[j] * vh;
                ad += s_fhi[j] * vl; dd += s_fhi[j] * vh;
            }
        }
        const int32_t qad = quant_f32(ad, mk, a.q, qf, has_m), qda = quant_f32(da, mk, a.q, qf, has_m),
                      qdd = quant_f32(dd, mk, a.q, qf, has_m);
        if (a.last) {
            const int32_t qaa = quant_f32(aa, mk, a.q, qf, has_m);
            co[(size_t)oh * a.enc_w + ow] = qaa;
            amax = max(amax, iabs_u(qaa));
        } else {
            llo[(size_t)oh * a.out_w + ow] = aa;
        }
        co[(size_t)oh * a.enc_w + a.off_w + ow] = qad;
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = qda;
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = qdd;
        amax = max(amax, max(iabs_u(qad), max(iabs_u(qda), iabs_u(qdd))));
    }
    if (a.maxabs != nullptr) {
        block_raise_max(&a.maxabs[plane / a.c], amax, &s_amax);
    }
}

// ---- colour ---------------------------------------------------------------------------------------------------------
#define SPOW_FN __device__ __forceinline__
#define SPOW_FMA(a, b, c) fma((a), (b), (c))
#define SPOW_RINT(a) rint(a)
#define SPOW_LDEXP(a, n) ldexp((a), (n))
#define SPOW_TABLE_QUAL __device__ const
#include "spow_tables.h"
#include "spow.h"

// the three tables of the power function, copied to LDS by every kernel that converts colours (per-lane table reads)
struct SpowLds {
    double inv[SPOW_N], log2c[SPOW_N], exp2t[SPOW_N];
};
__device__ __forceinline__ void spow_lds_fill(SpowLds &t, int tid) {  // needs a barrier before the first use
    if (tid < SPOW_N) { t.inv[tid] = SPOW_INV[tid]; t.log2c[tid] = SPOW_LOG2C[tid]; t.exp2t[tid] = SPOW_EXP2[tid]; }
}

// One pixel of the colour model change.  Shared by the stand-alone kernel (k_color3) and the fused level-1 kernels, and
// this file is compiled without multiply-add contraction: the three produce the same bits.  (The CPU checker,
// oracle/color_oracle.c, is an arithmetic of its own with the C library's pow(): the device is held to it within a
// tolerance, tests/test_gpu_image.py.)  numpy's dot order.
__device__ __forceinline__ void color3_px(const Color3 &c, const SpowLds &t, double u0, double u1, double u2, double &w0,
                                          double &w1, double &w2) {
    double v[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double x = (u0 * c.A[3 * r] + u1 * c.A[3 * r + 1]) + u2 * c.A[3 * r + 2];
#ifdef SPIHT_COLOR_POW   // diagnostic: the device library's pow() (<= 1 ulp, about 250 float64 instruction slots per call)
        const double m = pow(fabs(x), c.p);
        v[r] = x < 0.0 ? -m : (x > 0.0 ? m : 0.0);
#else
        v[r] = spow_signed(x, c.p, t.inv, t.log2c, t.exp2t);
#endif
    }
    w0 = (v[0] * c.M[0] + v[1] * c.M[1]) + v[2] * c.M[2];
    w1 = (v[0] * c.M[3] + v[1] * c.M[4]) + v[2] * c.M[5];
    w2 = (v[0] * c.M[6] + v[1] * c.M[7]) + v[2] * c.M[8];
}

// Stand-alone colour model change of a batch of 3-channel float64 images [B,3,npix]; in place allowed.  The checker of
// the fused kernels, and the path of images that have no transform level.
__global__ __launch_bounds__(256) void k_color3(const double *__restrict__ in, double *__restrict__ out, size_t npix, Color3 c) {
    __shared__ SpowLds s_pw;
    spow_lds_fill(s_pw, threadIdx.x);
    __syncthreads();
    const size_t img = (size_t)blockIdx.y * 3 * npix;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < npix; t += (size_t)gridDim.x * 256) {
        double w0, w1, w2;
        color3_px(c, s_pw, in[img + t], in[img + npix + t], in[img + 2 * npix + t], w0, w1, w2);
        out[img + t] = w0;
        out[img + npix + t] = w1;
        out[img + 2 * npix + t] = w2;
    }
}
extern "C" int spiht_launch_color3(const double *d_in, double *d_out, int B, size_t npix, const double *A, const double *M,
                                   double p, hipStream_t st) {
    Color3 c;
    for (int i = 0; i < 9; i++) { c.A[i] = A[i]; c.M[i] = M[i]; }
    c.p = p;
    const unsigned gx = (unsigned)((npix + 1023) / 1024 < 1 ? 1 : (npix + 1023) / 1024);
    hipLaunchKernelGGL(k_color3, dim3(gx, (unsigned)B), dim3(256), 0, st, d_in, d_out, npix, c);
    return (int)hipGetLastError();
}

// ---- level 1 of a 3-channel image with the colour model change on its loads ---------------------
// The change needs all three planes of a pixel and costs three pow() per pixel -- about as much arithmetic as the
// transform costs memory time -- so every pixel must be converted exactly once: a workgroup owns a strip of C1_SW output
// columns of ALL THREE channels and marches down C1_ROWS output rows.  Thread = input column: per step it loads the
// R, G, B samples of two new rows (C1_PF steps ahead), converts them, slides them into a register window of F rows per
// channel, and filters down the column; the low / high rows of the three channels go through a double-buffered LDS row
// to the axis -1 filter (threads 0..127: aa, ad from the low rows; 128..255: da, dd from the high rows).  One barrier per
// output row; a pixel is loaded and converted once per strip (F-2 rows of overlap between vertically adjacent strips).
// Same sums in the same order as k_dwt_level on converted pixels, overhang order included: bit-identical outputs.
#ifndef C1_PF
#define C1_PF 1   // steps of look-ahead of the raw samples.  256 x 1024x1024: 1 -> 5.6 ms, 2 -> 7.1 (spills at 128 VGPRs wait
#endif            // for the look-ahead loads in front of them); 3 waves / SIMD without spills: 6.1 - 6.3; 2: 7.7
#define C1_ROWS 136
#ifndef C1_WPE
#define C1_WPE 4
#endif
template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C1_WPE, C1_WPE)))  // 128 VGPRs: left alone the compiler interleaves
void k_dwt1_color(DwtKArgs a, uint32_t gx, uint32_t gy) {                       // the six powers of a step over 173 (2 waves / SIMD)
    constexpr int SW = (256 - (F - 2)) / 2;  // output columns per strip: exactly 256 input columns
    constexpr int HC = 128 + 2;
    __shared__ double s_lo[2][3][2][HC];     // [output row parity][channel][column parity][column >> 1]
    __shared__ double s_hi[2][3][2][HC];
    __shared__ SpowLds s_pw;
    __shared__ uint32_t s_amax;
    spow_lds_fill(s_pw, threadIdx.x);
    if (threadIdx.x == 0) s_amax = 0;
    __syncthreads();
    uint32_t tbx, tby, tbz;
    xcd_tile(gx, gy, a.planes / 3, tbx, tby, tbz);
    const int img = (int)tbz;
    const int ow0 = (int)tbx * SW, oa = (int)tby * C1_ROWS;
    const int ob = min(oa + C1_ROWS, a.out_h);
    const size_t npl = (size_t)a.in_h * a.in_w;
    const double *__restrict__ in = a.in + (size_t)img * 3 * npl;
    const int tid = threadIdx.x;
    const int gc = ext_index(2 * ow0 + 2 - F + tid, a.in_w, a.mode);
    auto ld = [&](int r, double &u0, double &u1, double &u2) {  // raw R, G, B of (row r, this column), extension applied
        const int gr = ext_index(r, a.in_h, a.mode);
        const bool z = gc < 0 || gr < 0;
        const size_t o = z ? 0 : (size_t)gr * a.in_w + gc;
        u0 = in[o]; u1 = in[o + npl]; u2 = in[o + 2 * npl];
        if (z) { u0 = 0.0; u1 = 0.0; u2 = 0.0; }
    };
    // "zero" extension pads the CONVERTED signal with zeros (pywt extends what it is given): convert, then zero
    auto cv = [&](int r, double u0, double u1, double u2, double &w0, double &w1, double &w2) {
        color3_px(a.col, s_pw, u0, u1, u2, w0, w1, w2);
        if (gc < 0 || ext_index(r, a.in_h, a.mode) < 0) { w0 = 0.0; w1 = 0.0; w2 = 0.0; }
    };
    double win[3][F];  // rows 2o+2-F .. 2o+1 of output row o, converted
#pragma unroll
    for (int t = 0; t < F; t++) {
        double u0, u1, u2;
        ld(2 * oa + 2 - F + t, u0, u1, u2);
        cv(2 * oa + 2 - F + t, u0, u1, u2, win[0][t], win[1][t], win[2][t]);
    }
    double pq[C1_PF][2][3];  // raw samples of the two rows each of the next C1_PF steps brings in
#pragma unroll
    for (int u = 0; u < C1_PF; u++)
#pragma unroll
        for (int e = 0; e < 2; e++) ld(2 * (oa + 1 + u) + e, pq[u][e][0], pq[u][e][1], pq[u][e][2]);

    const bool has_m = a.mults != nullptr;
    double mk[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) mk[ch] = has_m ? a.mults[ch] : 1.0;
    const int role = tid >> 7, wcol = tid & 127;
    const int ow = ow0 + wcol;
    const bool wr = wcol < SW && ow < a.out_w;
    const size_t cpl = (size_t)a.enc_h * a.enc_w, lpl = (size_t)a.out_h * a.out_w;
    int32_t *__restrict__ co = a.coeffs + (size_t)img * 3 * cpl;
    double *__restrict__ llo = a.last ? nullptr : a.ll_out + (size_t)img * 3 * lpl;
    uint32_t amax = 0;
    const int par = tid & 1, hc = tid >> 1;

    for (int o0 = oa; o0 < ob; o0 += C1_PF) {
#pragma unroll
        for (int u = 0; u < C1_PF; u++) {
            const int o = o0 + u;
            // ---- axis -2 for output row o, three channels ----
            if (o < a.ov_h) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    double sl = 0.0, shh = 0.0;
#pragma unroll
                    for (int j = 0; j < F; j++) {
                        if ((LOM >> j) & 1u) sl += a.lo[j] * win[ch][F - 1 - j];
                        if ((HIM >> j) & 1u) shh += a.hi[j] * win[ch][F - 1 - j];
                    }
                    s_lo[o & 1][ch][par][hc] = sl;
                    s_hi[o & 1][ch][par][hc] = shh;
                }
            } else {  // bottom overhang: PyWavelets' order (k_dwt_level); the window element is picked with selects
                const int jb = 2 * o + 1 - a.in_h;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    double sl = 0.0, shh = 0.0;
                    for (int s2 = 0; s2 < F; s2++) {
                        const int j = s2 <= jb ? jb - s2 : s2;
                        double v = win[ch][0];
#pragma unroll
                        for (int t = 1; t < F; t++) v = (F - 1 - j == t) ? win[ch][t] : v;
                        sl += a.lo[j] * v;
                        shh += a.hi[j] * v;
                    }
                    s_lo[o & 1][ch][par][hc] = sl;
                    s_hi[o & 1][ch][par][hc] = shh;
                }
            }
            // slide the windows: the two rows this step brings in are converted now, their successors requested
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int t = 0; t < F - 2; t++) win[ch][t] = win[ch][t + 2];
#pragma unroll
            for (int e = 0; e < 2; e++)
                cv(2 * (o + 1) + e, pq[u][e][0], pq[u][e][1], pq[u][e][2], win[0][F - 2 + e], win[1][F - 2 + e], win[2][F - 2 + e]);
#pragma unroll
            for (int e = 0; e < 2; e++) ld(2 * (o + 1 + C1_PF) + e, pq[u][e][0], pq[u][e][1], pq[u][e][2]);
            __syncthreads();
            // ---- axis -1: two sub-bands per thread and channel ----
            if (wr && o < ob) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const double(*src)[HC] = role ? s_hi[o & 1][ch] : s_lo[o & 1][ch];
                    double r0 = 0.0, r1 = 0.0;  // role 0: aa, ad   role 1: da, dd
                    if (ow < a.ov_w) {
#pragma unroll
                        for (int j = 0; j < F; j++) {
                            const double v = src[(F - 1 - j) & 1][wcol + ((F - 1 - j) >> 1)];
                            if ((LOM >> j) & 1u) r0 += a.lo[j] * v;
                            if ((HIM >> j) & 1u) r1 += a.hi[j] * v;
                        }
                    } else {  // right overhang
                        const int jb = 2 * ow + 1 - a.in_w;
                        for (int s2 = 0; s2 < F; s2++) {
                            const int j = s2 <= jb ? jb - s2 : s2;
                            const double v = src[(F - 1 - j) & 1][wcol + ((F - 1 - j) >> 1)];
                            r0 += a.lo[j] * v;
                            r1 += a.hi[j] * v;
                        }
                    }
                    int32_t *cc = co + (size_t)ch * cpl;
                    const int32_t q1 = quant(r1, mk[ch], a.q, has_m);
                    amax = max(amax, iabs_u(q1));
                    if (role == 0) {
                        if (a.last) {
                            const int32_t q0 = quant(r0, mk[ch], a.q, has_m);
                            cc[(size_t)o * a.enc_w + ow] = q0;
                            amax = max(amax, iabs_u(q0));
                        } else {
                            llo[(size_t)ch * lpl + (size_t)o * a.out_w + ow] = r0;
                        }
                        cc[(size_t)o * a.enc_w + a.off_w + ow] = q1;                      // 'ad' top-right
                    } else {
                        const int32_t q0 = quant(r0, mk[ch], a.q, has_m);
                        amax = max(amax, iabs_u(q0));
                        cc[(size_t)(a.off_h + o) * a.enc_w + ow] = q0;                     // 'da' bottom-left
                        cc[(size_t)(a.off_h + o) * a.enc_w + a.off_w + ow] = q1;           // 'dd' bottom-right
                    }
                }
            }
        }
    }
    if (a.maxabs != nullptr) {
        block_raise_max(&a.maxabs[img], amax, &s_amax);
    }
}

// ---- forward level for the extension modes that COMPUTE the extended sample (smooth, antisymmetric, antireflect) ----------
// PyWavelets' modes beyond the five index maps (the reference passes SpihtSettings.mode through, spiht_wrapper.py:163).
// A sample outside the signal is a function of the samples at the edge, and along axis -1 pywt extends the INTERMEDIATE
// (axis -2 already filtered) rows -- which is not the filtered extension of the pixels in the last bits -- so these modes
// run as two plain passes, one thread per output, through a float64 intermediate in memory: correctness first, no tiling
// (a path nobody's headline runs on; the tiled kernel above serves the index-map modes).  Summation order as pywt's
// downsampling_convolution: taps ascending, except that on the right overhang of an input at least as long as the filter
// the taps that read the extension come first, nearest first -- smooth, like constant, keeps ascending order there too.
template <typename T>
__device__ __forceinline__ T ext_value(const T *x, int N, size_t sx, int i, int mode) {
    if (mode == 8) {  // periodization: the signal, made even by repeating its last sample, continued periodically
        const int Np = N + (N & 1);
        int m = i % Np;
        if (m < 0) m += Np;
        return x[(size_t)(m < N ? m : N - 1) * sx];
    }
    if (i >= 0 && i < N) return x[(size_t)i * sx];
    if (mode < 5) {  // the index maps (filters longer than the tiled kernels take come through here too)
        const int m = ext_index(i, N, mode);
        return m < 0 ? (T)0 : x[(size_t)m * sx];
    }
    if (mode == 5) {  // smooth: the straight line through the two samples at the edge
        if (N < 2) return x[0];
        if (i < 0) return x[0] + (T)(-i) * (x[0] - x[sx]);
        return x[(size_t)(N - 1) * sx] + (T)(i - N + 1) * (x[(size_t)(N - 1) * sx] - x[(size_t)(N - 2) * sx]);
    }
    if (mode == 6) {  // antisymmetric: half-sample mirror image, every other block of N samples negated
        const int P = 2 * N;
        int m = i % P;
        if (m < 0) m += P;
        const int blk = (i - m) / N + (m >= N ? 1 : 0);
        const T v = x[(size_t)(m < N ? m : P - 1 - m) * sx];
        return (blk & 1) ? -v : v;
    }
    // antireflect: whole-sample mirror image through the edge VALUE; the value a block ends on is the next block's edge
    if (N < 2) return x[0];
    const bool left = i < 0;
    int d = left ? -i : i - N + 1;
    T e = left ? x[0] : x[(size_t)(N - 1) * sx];
    bool away = true;  // the first block walks away from the edge it started at and subtracts; the next one comes back and adds
    for (;;) {
        const int k = d <= N - 1 ? d : N - 1;
        const bool from_left = left ? away : !away;
        const T dlt = from_left ? x[(size_t)k * sx] - x[0] : x[(size_t)(N - 1 - k) * sx] - x[(size_t)(N - 1) * sx];
        const T v = away ? e - dlt : e + dlt;
        if (d <= N - 1) return v;
        e = v;
        d -= N - 1;
        away = !away;
    }
}

// one analysis pass along one axis: out position o of line `line` of plane `plane`.  n_lines lines of length N, element
// stride sx, line stride sl; outputs: lo / hi [plane][L][n_lines] laid out with the same strides roles (so, sol).
// T = float: the single-precision transform PyWavelets runs on float32 / float16 pixels (its own filter values).
struct DwtAxisArgs {
    int32_t F, mode, N, L, n_lines, planes;
    int32_t i0, pad;             // output o reads the window that ends at sample 2 o + i0: 1, or F / 2 under periodization
    size_t sx, sl, plane_in;     // input: element stride along the axis, stride between lines, plane stride
    size_t so, sol, plane_out;   // output alike
    const void *in;
    void *lo, *hi;
    const double *flo, *fhi;     // device: dec_lo, dec_hi (any length: the whole PyWavelets table goes through this path)
    const float *flo_f, *fhi_f;  // ... as the single-precision transform has them
};
template <typename T>
__global__ __launch_bounds__(256) void k_dwt_axis_ext(DwtAxisArgs a) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_plane = (size_t)a.L * a.n_lines;
    if (t >= per_plane * (size_t)a.planes) return;
    const int plane = (int)(t / per_plane);
    const size_t r = t - (size_t)plane * per_plane;
    // consecutive threads along the contiguous direction of the output
    int o, line;
    if (a.so == 1) { line = (int)(r / a.L); o = (int)(r - (size_t)line * a.L); }
    else { o = (int)(r / a.n_lines); line = (int)(r - (size_t)o * a.n_lines); }
    const T *x = reinterpret_cast<const T *>(a.in) + (size_t)plane * a.plane_in + (size_t)line * a.sl;
    const int i = 2 * o + a.i0;
    const int jb = (i >= a.N && a.mode != 5 && a.mode != 4) ? i - a.N : -1;  // (smooth and constant: ascending throughout)
    T sa = 0, sd = 0;
    for (int s2 = 0; s2 < a.F; s2++) {
        const int j = s2 <= jb ? jb - s2 : s2;
        const T v = ext_value<T>(x, a.N, a.sx, i - j, a.mode);
        const T fl = sizeof(T) == 4 ? (T)a.flo_f[j] : (T)a.flo[j], fh = sizeof(T) == 4 ? (T)a.fhi_f[j] : (T)a.fhi[j];
        sa += fl * v;
        sd += fh * v;
    }
    const size_t oo = (size_t)plane * a.plane_out + (size_t)line * a.sol + (size_t)o * a.so;
    reinterpret_cast<T *>(a.lo)[oo] = sa;
    reinterpret_cast<T *>(a.hi)[oo] = sd;
}
// quantise + pack the four sub-bands of a level computed by the two passes: aa / ad from the low rows, da / dd from the high rows
struct DwtPackArgs {
    int32_t c, out_h, out_w, off_h, off_w, enc_h, enc_w, last, planes, pad;
    const void *aa, *ad, *da, *dd;   // [planes, out_h, out_w]
    void *ll_out;
    int32_t *coeffs;
    const double *mults;
    uint32_t *maxabs;
    double q;
};
template <typename T>
__global__ __launch_bounds__(256) void k_dwt_pack_ext(DwtPackArgs a) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)a.out_h * a.out_w;
    uint32_t amax = 0;
    int img = -1;
    if (t < per * (size_t)a.planes) {
        const int plane = (int)(t / per);
        const size_t r = t - (size_t)plane * per;
        const int oh = (int)(r / a.out_w), ow = (int)(r - (size_t)oh * a.out_w);
        const bool has_m = a.mults != nullptr;
        const double mk = has_m ? a.mults[plane % a.c] : 1.0;
        const float qf = (float)a.q;
        auto qz = [&](T v) -> int32_t {  // the wrapper's arithmetic in the array's precision (spiht_wrapper.py:167-172, :9-11)
            if (sizeof(T) == 4) return quant_f32((float)v, mk, a.q, qf, has_m);
            return quant((double)v, mk, a.q, has_m);
        };
        int32_t *co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
        const T *aa = reinterpret_cast<const T *>(a.aa), *ad = reinterpret_cast<const T *>(a.ad);
        const T *da = reinterpret_cast<const T *>(a.da), *dd = reinterpret_cast<const T *>(a.dd);
        const int32_t qad = qz(ad[t]), qda = qz(da[t]), qdd = qz(dd[t]);
        if (a.last) {
            const int32_t qaa = qz(aa[t]);
            co[(size_t)oh * a.enc_w + ow] = qaa;
            amax = iabs_u(qaa);
        } else {
            reinterpret_cast<T *>(a.ll_out)[t] = aa[t];
        }
        co[(size_t)oh * a.enc_w + a.off_w + ow] = qad;
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = qda;
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = qdd;
        amax = max(amax, max(iabs_u(qad), max(iabs_u(qda), iabs_u(qdd))));
        img = plane / a.c;
    }
    if (a.maxabs != nullptr) wave_raise_max(a.maxabs, img, amax);
}
// tmp: 6 arrays of planes*out_h*in_w (2) and planes*out_h*out_w (4) elements (float when a->f32), carved by the caller
// d_filt: the wavelet's filters on the device -- dec_lo, dec_hi, rec_lo, rec_hi (F doubles each), then dec_lo, dec_hi as the
// single-precision transform has them (F floats each)
extern "C" int spiht_launch_dwt_level_ext(const DwtKArgs *a, int planes, void *t_lo, void *t_hi, void *b_aa, void *b_ad,
                                          void *b_da, void *b_dd, const double *d_filt, hipStream_t st) {
    DwtAxisArgs x;
    memset(&x, 0, sizeof(x));
    x.F = a->F; x.mode = a->mode; x.planes = planes;
    x.i0 = a->mode == 8 ? a->F / 2 : 1;
    x.flo = d_filt; x.fhi = d_filt + a->F;
    x.flo_f = reinterpret_cast<const float *>(d_filt + 4 * (size_t)a->F); x.fhi_f = x.flo_f + a->F;
    const bool f32 = a->f32 != 0;
    auto axis = [&](size_t n) {
        if (f32) hipLaunchKernelGGL(k_dwt_axis_ext<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x);
        else hipLaunchKernelGGL(k_dwt_axis_ext<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x);
    };
    // axis -2: lines = columns
    x.N = a->in_h; x.L = a->out_h; x.n_lines = a->in_w;
    x.sx = (size_t)a->in_w; x.sl = 1; x.plane_in = (size_t)a->in_h * a->in_w;
    x.so = (size_t)a->in_w; x.sol = 1; x.plane_out = (size_t)a->out_h * a->in_w;
    x.in = a->in; x.lo = t_lo; x.hi = t_hi;
    axis((size_t)planes * x.L * x.n_lines);
    // axis -1 on the low rows, then on the high rows: lines = rows
    x.N = a->in_w; x.L = a->out_w; x.n_lines = a->out_h;
    x.sx = 1; x.sl = (size_t)a->in_w; x.plane_in = (size_t)a->out_h * a->in_w;
    x.so = 1; x.sol = (size_t)a->out_w; x.plane_out = (size_t)a->out_h * a->out_w;
    const size_t n = (size_t)planes * x.L * x.n_lines;
    x.in = t_lo; x.lo = b_aa; x.hi = b_ad;
    axis(n);
    x.in = t_hi; x.lo = b_da; x.hi = b_dd;
    axis(n);
    DwtPackArgs p;
    memset(&p, 0, sizeof(p));
    p.c = a->c; p.out_h = a->out_h; p.out_w = a->out_w; p.off_h = a->off_h; p.off_w = a->off_w; p.enc_h = a->enc_h; p.enc_w = a->enc_w;
    p.last = a->last; p.planes = planes;
    p.aa = b_aa; p.ad = b_ad; p.da = b_da; p.dd = b_dd;
    p.ll_out = a->ll_out; p.coeffs = a->coeffs; p.mults = a->mults; p.maxabs = a->maxabs; p.q = a->q;
    if (f32) hipLaunchKernelGGL(k_dwt_pack_ext<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_dwt_pack_ext<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

// ---- inverse level under periodization (pywt upsampling_convolution_valid_sf_periodization) ---------------------------------
// 2 L samples back from L + L coefficients: x[n] = sum_k rec[n - 2k + F/2 - 1] c[k mod L], i.e. with p = (n + F/2 - 1) & 1 and
// i = (n + F/2 - 1 - p) / 2 the terms j = 0 .. F/2-1: tap 2j + p against c[(i - j) mod L]; pywt adds every product straight
// into the output sample, the approximation's first, then the detail's; where the window hangs over the right end (i >= L)
// the terms beyond it come first, nearest first -- and sample 0 of a filter with an even number of tap pairs is summed
// like the odd half of the last window (jb = F/4 - 1).  One thread per output sample, two passes (axis -1, then axis -2)
// through an intermediate: the plain form of a mode nobody's headline runs on.
struct IdwtPerArgs {
    int32_t F, L, n_lines, planes;     // L coefficients per line, n_lines lines
    int32_t per, n_out;                // periodization (2 L samples) or the plain synthesis (2 L - F + 2 samples: filters longer than
                                       // the tiled kernels take)
    const double *flo_d, *fhi_d;       // device: rec_lo, rec_hi
    size_t si, so;                     // element stride along the axis: inputs, output
    size_t sla, sld, slo;              // stride between lines: approximation input, detail input, output
    size_t plane_a, plane_d, plane_out;
    const double *ca, *cd;             // float inputs ...
    const int32_t *qa, *qd;            // ... or quantised ones (dequantised on the fly, (v / m) / q as the wrapper does)
    int32_t a_is_q, d_is_q, c, has_m;
    const double *mults;
    double q;
    double *out;
};
__global__ __launch_bounds__(256) void k_idwt_axis_per(IdwtPerArgs a) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_plane = (size_t)a.n_out * a.n_lines;
    if (t >= per_plane * (size_t)a.planes) return;
    const int plane = (int)(t / per_plane);
    const size_t r = t - (size_t)plane * per_plane;
    int n, line;  // consecutive threads along the contiguous direction of the output
    if (a.so == 1) { line = (int)(r / (size_t)a.n_out); n = (int)(r - (size_t)line * a.n_out); }
    else { n = (int)(r / a.n_lines); line = (int)(r - (size_t)n * a.n_lines); }
    const int HF = a.F / 2;
    const bool has_m = a.has_m != 0;
    const double mk = has_m ? a.mults[plane % a.c] : 1.0;
    const size_t ba = (size_t)plane * a.plane_a + (size_t)line * a.sla, bd = (size_t)plane * a.plane_d + (size_t)line * a.sld;
    auto va = [&](int k) -> double { return a.a_is_q ? dequant(a.qa[ba + (size_t)k * a.si], mk, a.q, has_m) : a.ca[ba + (size_t)k * a.si]; };
    auto vd = [&](int k) -> double { return a.d_is_q ? dequant(a.qd[bd + (size_t)k * a.si], mk, a.q, has_m) : a.cd[bd + (size_t)k * a.si]; };
    double res;
    if (a.per) {
        const int s0 = HF - 1, p = (n + s0) & 1, i = (n + s0 - p) / 2;
        int jb = i >= a.L ? i - a.L : -1;
        if (n == 0 && (HF & 1) == 0) jb = a.F / 4 - 1;
        double acc = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const double *f = pass ? a.fhi_d : a.flo_d;
            for (int s2 = 0; s2 < HF; s2++) {
                const int j = s2 <= jb ? jb - s2 : s2;
                int k = (i - j) % a.L;
                if (k < 0) k += a.L;
                acc += f[2 * j + p] * (pass ? vd(k) : va(k));
            }
        }
        res = acc;
    } else {
        // upsampling_convolution_valid_sf: sample n = 2 (i - (F/2 - 1)) + p; the approximation's sum and the detail's sum kept
        // apart, each over j ascending (tap 2j + p against coefficient i - j, those inside the band), then added
        const int p = n & 1, i = n / 2 + HF - 1;
        double sa = 0.0, sd = 0.0;
        for (int j = 0; j < HF; j++) {
            const int k = i - j;
            if (k < 0 || k >= a.L) continue;
            sa += a.flo_d[2 * j + p] * va(k);
            sd += a.fhi_d[2 * j + p] * vd(k);
        }
        res = (0.0 + sa) + sd;
    }
    a.out[(size_t)plane * a.plane_out + (size_t)line * a.slo + (size_t)n * a.so] = res;
}
// t_lo, t_hi: planes * band_h * out_w doubles each (out_w = 2 band_w, out_h = 2 band_h)
// per: periodization (a->out_h / out_w = 2 x band) or the plain synthesis for a filter of any length (2 x band - F + 2);
// d_filt: the wavelet's filters on the device (see spiht_launch_dwt_level_ext)
extern "C" int spiht_launch_idwt_level_per(const IdwtKArgs *a, int planes, double *t_lo, double *t_hi, const double *d_filt, int per,
                                           hipStream_t st) {
    IdwtPerArgs x;
    memset(&x, 0, sizeof(x));
    x.F = a->F; x.planes = planes; x.c = a->c; x.has_m = a->mults != nullptr; x.mults = a->mults; x.q = a->q;
    x.per = per;
    x.flo_d = d_filt + 2 * (size_t)a->F; x.fhi_d = d_filt + 3 * (size_t)a->F;
    const size_t enc_plane = (size_t)a->enc_h * a->enc_w;
    // axis -1 (PyWavelets' idwtn takes the last axis first): band_h lines of band_w coefficients -> out_w samples
    x.L = a->band_w; x.n_lines = a->band_h; x.n_out = a->out_w;
    x.si = 1; x.so = 1; x.slo = (size_t)a->out_w; x.plane_out = (size_t)a->band_h * a->out_w;
    const size_t n1 = (size_t)planes * x.n_out * x.n_lines;
    // (aa, ad) -> low rows: the approximation comes out of the packed array (coarsest level) or from the level before, whose
    // array may be one sample longer than the band in either direction (waverec2 drops it)
    if (a->first) { x.a_is_q = 1; x.qa = a->rec; x.plane_a = enc_plane; x.sla = (size_t)a->enc_w; }
    else { x.a_is_q = 0; x.ca = a->a_in; x.plane_a = (size_t)a->a_h * a->a_w; x.sla = (size_t)a->a_w; }
    x.d_is_q = 1; x.qd = a->rec + a->off_w; x.plane_d = enc_plane; x.sld = (size_t)a->enc_w;
    x.out = t_lo;
    hipLaunchKernelGGL(k_idwt_axis_per, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, x);
    // (da, dd) -> high rows
    x.a_is_q = 1; x.qa = a->rec + (size_t)a->off_h * a->enc_w; x.plane_a = enc_plane; x.sla = (size_t)a->enc_w;
    x.qd = a->rec + (size_t)a->off_h * a->enc_w + a->off_w;
    x.out = t_hi;
    hipLaunchKernelGGL(k_idwt_axis_per, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, x);
    // axis -2: out_w lines (columns) of band_h coefficients -> out_h samples
    x.L = a->band_h; x.n_lines = a->out_w; x.n_out = a->out_h;
    x.si = (size_t)a->out_w; x.sla = 1; x.sld = 1; x.plane_a = x.plane_d = (size_t)a->band_h * a->out_w;
    x.so = (size_t)a->out_w; x.slo = 1; x.plane_out = (size_t)a->out_h * a->out_w;
    x.a_is_q = 0; x.d_is_q = 0; x.ca = t_lo; x.cd = t_hi;
    x.out = a->out;
    const size_t n2 = (size_t)planes * x.n_out * x.n_lines;
    hipLaunchKernelGGL(k_idwt_axis_per, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, x);
    return (int)hipGetLastError();
}

// zero the padding cells of coeffs_to_array: per level the strip below 'ad' and the strip right of 'da'.
// grid: (blocks, nrects, planes)
struct PadRects {
    int32_t n;
    int32_t enc_h, enc_w, pad;
    int32_t r0[2 * SPIHT_MAX_LEVELS], r1[2 * SPIHT_MAX_LEVELS], c0[2 * SPIHT_MAX_LEVELS], c1[2 * SPIHT_MAX_LEVELS];
};
__global__ __launch_bounds__(256) void k_zero_pads(PadRects pr, int32_t *coeffs) {
    const int rc = blockIdx.y;
    const int r0 = pr.r0[rc], r1 = pr.r1[rc], c0 = pr.c0[rc], c1 = pr.c1[rc];
    const int w = c1 - c0, cells = (r1 - r0) * w;
    int32_t *co = coeffs + (size_t)blockIdx.z * pr.enc_h * pr.enc_w;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < cells; t += gridDim.x * blockDim.x) {
        int r = t / w, cidx = t - r * w;
        co[(size_t)(r0 + r) * pr.enc_w + c0 + cidx] = 0;
    }
}

// level 0 of the API (no decomposition): quantise the image itself. grid-stride.
__global__ __launch_bounds__(256) void k_quant_plain(const double *in, int32_t *out, size_t n_per_plane, int planes, int c,
                                                     const double *mults, double q, uint32_t *maxabs) {
    size_t total = n_per_plane * (size_t)planes;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t0 = (size_t)blockIdx.x * blockDim.x; t0 < total; t0 += stride) {  // (t0: the same in every lane of a wavefront's block)
        const size_t t = t0 + threadIdx.x;
        int img = -1;
        uint32_t av = 0;
        if (t < total) {
            int plane = (int)(t / n_per_plane);
            bool has_m = mults != nullptr;
            int32_t v = quant(in[t], has_m ? mults[plane % c] : 1.0, q, has_m);
            out[t] = v;
            img = plane / c;
            av = iabs_u(v);
        }
        if (maxabs != nullptr) wave_raise_max(maxabs, img, av);
    }
}
__global__ __launch_bounds__(256) void k_dequant_plain(const int32_t *in, double *out, size_t n_per_plane, int planes, int c,
                                                       const double *mults, double q) {
    size_t total = n_per_plane * (size_t)planes;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int plane = (int)(t / n_per_plane);
        double v = (double)in[t];
        if (mults != nullptr) v = v / mults[plane % c];
        out[t] = v / q;
    }
}

// ------------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------------
// IW_TH x IW_TW (common.h): output rows / columns per tile.  Level 1 of 256 1080p images, workgroup per tile: 16 rows
// 4.02 ms, 20: 4.05, 24: 4.03, 32: 4.19, 12: 4.22, 40: 4.67, 8: 4.74; persistent kernel beside the list decoder (the
// pipelined schedule): 16 rows 7.3 ms, 20: 7.0, 24: 6.5-6.7, 28: 8.1, 32: 7.6


// grid: (ceil(out_w/TW), ceil(out_h/TH), planes).  LOM / HIM: non-zero taps of rec_lo / rec_hi (a product with a
// zero tap adds exactly nothing to `ca*lo + cd*hi`, so it is skipped).
template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(DW_BLOCK) void k_idwt_level(IdwtKArgs a) {
    // band index k contributes to output n with tap t = n + F - 2 - 2k in [0,F):  k in [n/2, n/2 + F/2 - 1]
    constexpr int HF = F / 2;
    constexpr int KH = IW_TH / 2 + HF - 1;   // band rows staged: outputs m0..m0+TH-1 need k in [m0/2, m0/2+TH/2-1+HF-1]
    constexpr int KW = IW_TW / 2 + HF - 1;   // band cols staged
    constexpr int KHH = IW_TH / 4 + HF - 1;  // band rows one half-tile walks
    __shared__ double s_b[4][KH][KW + 1];    // aa, ad, da, dd (dequantised)
    uint32_t tbx, tby, tbz;
    xcd_tile((a.out_w + IW_TW - 1) / IW_TW, (a.out_h + IW_TH - 1) / IW_TH, a.planes, tbx, tby, tbz);
    const int plane = (int)tbz;
    const int k = plane % a.c;
    const int m0 = (int)tby * IW_TH, n0 = (int)tbx * IW_TW;
    const int kh0 = m0 / 2, kw0 = n0 / 2;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    const bool zero_ok = (!has_m || mk > 0.0) && a.q > 0.0;  // 0/m/q == +0.0 exactly: skip the divisions
    const int32_t *__restrict__ rec = a.rec + (size_t)plane * a.enc_h * a.enc_w;
    const double *__restrict__ ain = a.first ? nullptr : a.a_in + (size_t)plane * a.a_h * a.a_w;
    const int tid = threadIdx.x;
    // L1Flags word of this tile still zero: the decoder wrote nothing into the detail bands it stages -- they are not
    // read (a zero goes through the same arithmetic: same bits)
    const bool occupied = a.flags == nullptr ||
        a.flags[((size_t)plane * ((a.out_h + IW_TH - 1) / IW_TH) + tby) * ((a.out_w + IW_TW - 1) / IW_TW) + tbx] != 0u;

    for (int p = tid; p < KH * KW; p += DW_BLOCK) {
        const int r = p / KW, cidx = p - r * KW;
        const int bi = kh0 + r, bj = kw0 + cidx;
        double vaa = 0.0, vad = 0.0, vda = 0.0, vdd = 0.0;
        if (bi < a.band_h && bj < a.band_w) {
            const int32_t rad = occupied ? rec[(size_t)bi * a.enc_w + a.off_w + bj] : 0;
            const int32_t rda = occupied ? rec[(size_t)(a.off_h + bi) * a.enc_w + bj] : 0;
            const int32_t rdd = occupied ? rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj] : 0;
            if (a.first) {
                const int32_t raa = rec[(size_t)bi * a.enc_w + bj];
                vaa = (raa == 0 && zero_ok) ? 0.0 : dequant(raa, mk, a.q, has_m);
            } else {
                vaa = ain[(size_t)bi * a.a_w + bj];
            }
            vad = (rad == 0 && zero_ok) ? 0.0 : dequant(rad, mk, a.q, has_m);
            vda = (rda == 0 && zero_ok) ? 0.0 : dequant(rda, mk, a.q, has_m);
            vdd = (rdd == 0 && zero_ok) ? 0.0 : dequant(rdd, mk, a.q, has_m);
        }
        s_b[0][r][cidx] = vaa; s_b[1][r][cidx] = vad; s_b[2][r][cidx] = vda; s_b[3][r][cidx] = vdd;
    }
    __syncthreads();

    // thread = (output column nn, half): walks the band rows its 16 output rows need
    const int nn = tid & (IW_TW - 1), half = tid / IW_TW;
    const int n = n0 + nn;
    const int np = n & 1;
    const int cl = nn / 2;            // first contributing band column, tile-relative (= n/2 - kw0)
    // taps along axis -1 for this column's parity: s = 0..HF-1  ->  t = np + F - 2 - 2s
    double tlo[HF], thi[HF];
#pragma unroll
    for (int s = 0; s < HF; s++) {
        tlo[s] = a.lo[np + F - 2 - 2 * s];
        thi[s] = a.hi[np + F - 2 - 2 * s];
    }
    double wl[HF], wh[HF];  // register window: tl/th of the last HF band rows (index HF-1 = newest)
#pragma unroll
    for (int s = 0; s < HF; s++) { wl[s] = 0.0; wh[s] = 0.0; }
    double *__restrict__ out = a.out + (size_t)plane * a.out_h * a.out_w;
    const int rbase = half * (IW_TH / 4);  // first band row (tile-relative) of this half
#pragma unroll
    for (int rr = 0; rr < KHH; rr++) {
        const int r = rbase + rr;
        // axis -1 synthesis of band row r at output column n.  Order of the additions as in pywt's
        // upsampling_convolution_valid_sf: the approximation and the detail contribution are separate sums, each over
        // j = 0..F/2-1 (tap 2j+parity against band column i-j, i.e. DEscending band column), and the detail sum is
        // added to the finished approximation sum -- the result is then bit-identical to pywt.waverec2's.
        double ta = 0.0, td = 0.0, ua = 0.0, ud = 0.0;
#pragma unroll
        for (int j = 0; j < HF; j++) {
            const int s = HF - 1 - j;
            // taps F-2-2s (even columns) and F-1-2s (odd columns): skip a product when both are zero
            constexpr uint32_t PAIR = 3u;
            const bool lnz = ((LOM >> (F - 2 - 2 * s)) & PAIR) != 0, hnz = ((HIM >> (F - 2 - 2 * s)) & PAIR) != 0;
            if (lnz) {
                ta += s_b[0][r][cl + s] * tlo[s];
                ua += s_b[2][r][cl + s] * tlo[s];
            }
            if (hnz) {
                td += s_b[1][r][cl + s] * thi[s];
                ud += s_b[3][r][cl + s] * thi[s];
            }
        }
        const double tl = (0.0 + ta) + td, th = (0.0 + ua) + ud;
#pragma unroll
        for (int s = 0; s < HF - 1; s++) { wl[s] = wl[s + 1]; wh[s] = wh[s + 1]; }
        wl[HF - 1] = tl;
        wh[HF - 1] = th;
        if (rr >= HF - 1) {
            // window holds band rows kb..kb+HF-1 with kb = kh0 + r - (HF-1): output rows 2kb, 2kb+1
            const int m = 2 * (kh0 + r - (HF - 1));
#pragma unroll
            for (int mp = 0; mp < 2; mp++) {
                double sa = 0.0, sd = 0.0;  // same order along axis -2
#pragma unroll
                for (int j = 0; j < HF; j++) {
                    const int s = HF - 1 - j;
                    const bool lnz = (LOM >> (mp + F - 2 - 2 * s)) & 1u, hnz = (HIM >> (mp + F - 2 - 2 * s)) & 1u;
                    if (lnz) sa += wl[s] * a.lo[mp + F - 2 - 2 * s];
                    if (hnz) sd += wh[s] * a.hi[mp + F - 2 - 2 * s];
                }
                const double sacc = (0.0 + sa) + sd;
                if (m + mp < a.out_h && n < a.out_w) out[(size_t)(m + mp) * a.out_w + n] = sacc;
            }
        }
    }
}

// ---- the inverse level as persistent workgroups that fetch one tile ahead ---------------------------------------------
// k_idwt_level is a workgroup per tile: request the tile's bands, wait a memory latency, filter, store, exit.  With few
// workgroups resident on a CU -- three when a list decoder lives there too (the pipelined schedule of
// spiht_amd/batch.py) -- a CU has three tiles in flight and the level is bound by that latency, not by bytes: 8.5 instead
// of 4.0 ms for level 1 of 256 1080p images, and not reading the detail bands at all changed nothing.
// Here a workgroup walks a sequence of tiles and requests tile k+1's samples (into registers: 15 per thread) before it
// filters and stores tile k, so the latency of k+1 runs under the work of k.  The barriers are bare s_barrier behind an
// lgkmcnt wait: __syncthreads() would wait for the loads in flight as well.  Same arithmetic in the same order as
// k_idwt_level: bit-identical output.  grid: IWP_WG workgroups per CU at most; tiles dealt XCD-contiguously.
#ifndef IWP_WG
#define IWP_WG 4
#endif

// FLAGS: a.flags holds one word per tile (common.h: L1Flags); the detail bands of a tile whose word is zero are all zero
// and are not read.  The loads stay unconditional -- a branch around them would make every later wait a wait for
// everything -- and go through a buffer descriptor of the plane instead: an offset beyond it returns 0 without a trip
// to memory.  The word of a tile is fetched when the tile's number becomes known, a tile of work ahead of its use.
template <int F, uint32_t LOM, uint32_t HIM, bool FIRST, bool FLAGS = false>  // FIRST: coarsest level, the approximation comes from the packed array
__global__ __launch_bounds__(DW_BLOCK) void k_idwt_level_pf(IdwtKArgs a, uint32_t gx, uint32_t gy, uint32_t *ctr,
                                                                      TileBase cb) {
    static_assert(!(FIRST && FLAGS), "the flags are those of level 1 of a transform with two levels or more");
    constexpr int HF = F / 2;
    constexpr int KH = IW_TH / 2 + HF - 1, KW = IW_TW / 2 + HF - 1, KHH = IW_TH / 4 + HF - 1;
    constexpr int NE = (KH * KW + DW_BLOCK - 1) / DW_BLOCK;  // staged elements per thread
    __shared__ double s_b[4][KH][KW + 1];                    // aa, ad, da, dd (dequantised)
    const int tid = threadIdx.x;
    // this workgroup's tiles: workgroups are dealt round-robin over the 8 XCDs; XCD x owns the contiguous tile range
    // [x*q + min(x, r), ...) and its workgroups (every 8th) take the tiles of that range in turn
    const uint32_t nt = gx * gy * (uint32_t)a.planes;
    const uint32_t x = blockIdx.x & 7u, q = nt >> 3, r8 = nt & 7u;
    const uint32_t base = x * q + (x < r8 ? x : r8), cnt = q + (x < r8 ? 1u : 0u);
    // Which tile next: with a fixed stride (tile k, k + per, ...) the workgroups drift apart over a few hundred tiles,
    // the tiles in flight on an XCD stop being neighbours and the halo rows two tiles share are fetched from HBM twice
    // (PMC: 40 instead of 31.6 MB read per 1080p image).  So the XCD's workgroups draw their tiles from a counter: what
    // is in flight is always one contiguous window of the range, as with one workgroup per tile.  Thread 0 draws a
    // tile two ahead (the answer travels under a whole tile of work) and passes it on through s_next.
    __shared__ uint32_t s_next[2];
    uint32_t *const myctr = ctr + 32u * x;
    const uint32_t cbase = cb.v[x];
    // "on a CU": who waits for this launch to be resident before launching a neighbour (spiht_ctx_wait_resident) counts these
    if (tid == 0) atomicAdd(ctr + TILECTR_STARTED, 1u);
    uint32_t k, kn;  // position in the XCD's range of the tile being worked on / of the one after
    if (tid == 0) {
        s_next[0] = atomicAdd(myctr, 1u) - cbase;
        s_next[1] = atomicAdd(myctr, 1u) - cbase;
    }
    __syncthreads();
    k = __builtin_amdgcn_readfirstlane(s_next[0]);  // (workgroup-uniform; said so, the tile arithmetic stays scalar)
    kn = __builtin_amdgcn_readfirstlane(s_next[1]);
    __syncthreads();

    struct Stage {  // the samples of one tile on their way from memory: 15 registers per thread
        int32_t rad[NE], rda[NE], rdd[NE], raa[NE];
        double vaa[NE];
        double mk;  // the plane's channel scale (a load as well: it must not be waited for on its own)
    };
    auto request = [&](Stage &g, uint32_t T, uint32_t occ) {  // the loads of tile T (nothing waits for them here)
        const uint32_t bx = T % gx, t2 = T / gx, by = t2 % gy, plane = t2 / gy;
        const int kh0 = (int)(by * IW_TH) / 2, kw0 = (int)(bx * IW_TW) / 2;
        const int32_t *__restrict__ rec = a.rec + (size_t)plane * a.enc_h * a.enc_w;
        const double *__restrict__ ain = FIRST ? nullptr : a.a_in + (size_t)plane * a.a_h * a.a_w;
        const __amdgpu_buffer_rsrc_t rrsrc = plane_rsrc(rec, (uint32_t)a.enc_h * (uint32_t)a.enc_w * 4u);  // (FLAGS only)
        // unconditional loads from clamped positions (a branch around a load makes the compiler wait for it at the
        // join); what lies outside the band is zeroed when the samples go to LDS
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int p = min(tid + e * DW_BLOCK, KH * KW - 1);
            const int rr = p / KW, cidx = p - rr * KW;
            const int bi = min(kh0 + rr, a.band_h - 1), bj = min(kw0 + cidx, a.band_w - 1);
            if (FLAGS) {
                const uint32_t o_ad = ((uint32_t)bi * (uint32_t)a.enc_w + (uint32_t)(a.off_w + bj)) * 4u;
                const uint32_t o_da = ((uint32_t)(a.off_h + bi) * (uint32_t)a.enc_w + (uint32_t)bj) * 4u;
                g.rad[e] = __builtin_amdgcn_raw_buffer_load_b32(rrsrc, occ ? o_ad : BUF_OOB, 0, 0);
                g.rda[e] = __builtin_amdgcn_raw_buffer_load_b32(rrsrc, occ ? o_da : BUF_OOB, 0, 0);
                g.rdd[e] = __builtin_amdgcn_raw_buffer_load_b32(rrsrc, occ ? o_da + (uint32_t)a.off_w * 4u : BUF_OOB, 0, 0);
            } else {
                g.rad[e] = rec[(size_t)bi * a.enc_w + a.off_w + bj];
                g.rda[e] = rec[(size_t)(a.off_h + bi) * a.enc_w + bj];
                g.rdd[e] = rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj];
            }
            if (FIRST) { g.raa[e] = rec[(size_t)bi * a.enc_w + bj]; g.vaa[e] = 0.0; }
            else { g.vaa[e] = ain[(size_t)bi * a.a_w + bj]; g.raa[e] = 0; }
        }
        g.mk = a.mults != nullptr ? a.mults[plane % (uint32_t)a.c] : 1.0;
    };
    // taps of axis -1 for this thread's column parity (n0 is even: the parity is the same in every tile)
    const int nn = tid & (IW_TW - 1), half = tid / IW_TW, np = nn & 1, cl = nn / 2;
    double tlo[HF], thi[HF];
#pragma unroll
    for (int s2 = 0; s2 < HF; s2++) {
        tlo[s2] = a.lo[np + F - 2 - 2 * s2];
        thi[s2] = a.hi[np + F - 2 - 2 * s2];
    }
    const bool has_m = a.mults != nullptr;
    // one tile: its samples (in g) -> LDS, then g is free and takes the loads of the next tile; filter; store
    auto tile = [&](Stage &g, uint32_t kk, uint32_t knext, uint32_t occ_next) {
        const uint32_t T = base + kk;
        const uint32_t bx = T % gx, t2 = T / gx, by = t2 % gy, plane = t2 / gy;
        const int kh0s = (int)(by * IW_TH) / 2, kw0s = (int)(bx * IW_TW) / 2;
        const double mk = g.mk;
        const bool zero_ok = (!has_m || mk > 0.0) && a.q > 0.0;  // 0/m/q == +0.0 exactly: skip the divisions
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int p = tid + e * DW_BLOCK;
            if (p < KH * KW) {
                const int rr = p / KW, cidx = p - rr * KW;
                const bool in = kh0s + rr < a.band_h && kw0s + cidx < a.band_w;
                const double va = FIRST ? ((g.raa[e] == 0 && zero_ok) ? 0.0 : dequant(g.raa[e], mk, a.q, has_m)) : g.vaa[e];
                s_b[0][rr][cidx] = in ? va : 0.0;
                s_b[1][rr][cidx] = (!in || (g.rad[e] == 0 && zero_ok)) ? 0.0 : dequant(g.rad[e], mk, a.q, has_m);
                s_b[2][rr][cidx] = (!in || (g.rda[e] == 0 && zero_ok)) ? 0.0 : dequant(g.rda[e], mk, a.q, has_m);
                s_b[3][rr][cidx] = (!in || (g.rdd[e] == 0 && zero_ok)) ? 0.0 : dequant(g.rdd[e], mk, a.q, has_m);
            }
        }
        // the tile after `knext` is drawn here -- behind the wait for this tile's samples, ahead of the next tile's
        // loads -- and looked at only at the end of the tile (raw: a subtraction here would wait for the answer)
        uint32_t drawn = 0;
        if (tid == 0) drawn = atomicAdd(myctr, 1u);
        request(g, base + min(knext, cnt - 1u), occ_next);  // unconditional (past the end: the last tile again, never used): a
        lds_barrier();                            // branch around loads makes every later wait a wait for all of them
        // ---- thread = (output column nn, half): as k_idwt_level ----
        const int m0 = (int)by * IW_TH, n0 = (int)bx * IW_TW, kh0 = m0 / 2;
        const int n = n0 + nn;
        double wl[HF], wh[HF];
#pragma unroll
        for (int s2 = 0; s2 < HF; s2++) { wl[s2] = 0.0; wh[s2] = 0.0; }
        const int rbase = half * (IW_TH / 4);
        // Stores through a buffer descriptor of the plane: what falls outside the picture (rows past the plane's end by
        // the descriptor's range check, columns past out_w by an offset beyond it) is dropped by the hardware, so the
        // stores stand in straight-line code.  With branches around them the compiler cannot count them, and the wait
        // for the next tile's samples at the top of the loop becomes a wait for these stores as well.
        const uint32_t row_bytes = (uint32_t)a.out_w * 8u;
        const __amdgpu_buffer_rsrc_t orsrc = plane_rsrc(a.out + (size_t)plane * a.out_h * a.out_w, (uint32_t)a.out_h * row_bytes);
        const uint32_t voff0 = n < a.out_w ? (uint32_t)(2 * (kh0 + rbase)) * row_bytes + (uint32_t)n * 8u : BUF_OOB;
#pragma unroll
        for (int rr = 0; rr < KHH; rr++) {
            const int r = rbase + rr;
            double ta = 0.0, td = 0.0, ua = 0.0, ud = 0.0;
#pragma unroll
            for (int j = 0; j < HF; j++) {
                const int s2 = HF - 1 - j;
                constexpr uint32_t PAIR = 3u;
                const bool lnz = ((LOM >> (F - 2 - 2 * s2)) & PAIR) != 0, hnz = ((HIM >> (F - 2 - 2 * s2)) & PAIR) != 0;
                if (lnz) {
                    ta += s_b[0][r][cl + s2] * tlo[s2];
                    ua += s_b[2][r][cl + s2] * tlo[s2];
                }
                if (hnz) {
                    td += s_b[1][r][cl + s2] * thi[s2];
                    ud += s_b[3][r][cl + s2] * thi[s2];
                }
            }
            const double tl = (0.0 + ta) + td, th = (0.0 + ua) + ud;
#pragma unroll
            for (int s2 = 0; s2 < HF - 1; s2++) { wl[s2] = wl[s2 + 1]; wh[s2] = wh[s2 + 1]; }
            wl[HF - 1] = tl;
            wh[HF - 1] = th;
            if (rr >= HF - 1) {
#pragma unroll
                for (int mp = 0; mp < 2; mp++) {
                    double sa = 0.0, sd = 0.0;
#pragma unroll
                    for (int j = 0; j < HF; j++) {
                        const int s2 = HF - 1 - j;
                        const bool lnz = (LOM >> (mp + F - 2 - 2 * s2)) & 1u, hnz = (HIM >> (mp + F - 2 - 2 * s2)) & 1u;
                        if (lnz) sa += wl[s2] * a.lo[mp + F - 2 - 2 * s2];
                        if (hnz) sd += wh[s2] * a.hi[mp + F - 2 - 2 * s2];
                    }
                    const double sacc = (0.0 + sa) + sd;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sacc), orsrc,
                                                          voff0 + (uint32_t)(2 * (rr - (HF - 1)) + mp) * row_bytes, 0, 0);
                }
            }
        }
        if (tid == 0) s_next[0] = drawn - cbase;
        lds_barrier();  // every wave has read the tile before the next one is written over it
    };
    // One tile ahead.  (Two ahead -- a second register set, 116 VGPRs -- leaves room for ONE such workgroup per CU beside
    // a decoder instead of two and was slower there: 9.2-10.6 instead of 7.2 ms for level 1.)
    Stage g0;
    if (k >= cnt) return;
    auto occupancy = [&](uint32_t kk) -> uint32_t {  // the L1Flags word of position kk of this XCD's range (clamped as the request is)
        return FLAGS ? a.flags[base + min(kk, cnt - 1u)] : 1u;
    };
    request(g0, base + k, occupancy(k));
    uint32_t occ_n = occupancy(kn);
    // The first tile stands outside the loop: inside it the wait for a tile's samples can then be "all but the stores
    // issued since" on every path into the loop head (with the first trip inside, nothing follows the samples' loads on
    // the path from above, and the compiler has to make it a wait for everything -- the stores of the tile before).
    tile(g0, k, kn, occ_n);
    k = kn;
    kn = __builtin_amdgcn_readfirstlane(s_next[0]);  // (s_next[0] is written again only behind the
                                                                         // next tile's first barrier)
    occ_n = occupancy(kn);
    while (k < cnt) {
        tile(g0, k, kn, occ_n);
        k = kn;
        kn = __builtin_amdgcn_readfirstlane(s_next[0]);
        occ_n = occupancy(kn);
    }
}

// ---- level 1 of the inverse transform of a 3-channel image with the colour model change on its stores ------------
// k_idwt_level for the three channels of a tile at once (same staging, same sums in the same order), so that the three
// values of an output pixel meet in one thread, which converts them (color3_px: three pow() per pixel) and stores the
// result: the transformed picture in the coded colour model never exists in memory.  Arithmetic-bound, so the tile is
// smaller than k_idwt_level's (IWC_TH rows: 38.6 KB of LDS for the three channels, four workgroups per CU).
#define IWC_TH 8
template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(DW_BLOCK) void k_idwt1_color(IdwtKArgs a) {
    constexpr int HF = F / 2;
    constexpr int KH = IWC_TH / 2 + HF - 1, KW = IW_TW / 2 + HF - 1, KHH = IWC_TH / 4 + HF - 1;
    __shared__ double s_b[3][4][KH][KW + 1];  // per channel: aa, ad, da, dd (dequantised)
    __shared__ SpowLds s_pw;
    spow_lds_fill(s_pw, threadIdx.x);      // (the barrier after the staging loop below covers it)
    uint32_t tbx, tby, tbz;
    xcd_tile((a.out_w + IW_TW - 1) / IW_TW, (a.out_h + IWC_TH - 1) / IWC_TH, a.planes / 3, tbx, tby, tbz);
    const int img = (int)tbz;
    const int m0 = (int)tby * IWC_TH, n0 = (int)tbx * IW_TW;
    const int kh0 = m0 / 2, kw0 = n0 / 2;
    const bool has_m = a.mults != nullptr;
    const int tid = threadIdx.x;
    const size_t cpl = (size_t)a.enc_h * a.enc_w, apl = (size_t)a.a_h * a.a_w, opl = (size_t)a.out_h * a.out_w;

    for (int p = tid; p < 3 * KH * KW; p += DW_BLOCK) {
        const int ch = p / (KH * KW), q = p - ch * (KH * KW);
        const int r = q / KW, cidx = q - r * KW;
        const int bi = kh0 + r, bj = kw0 + cidx;
        const double mk = has_m ? a.mults[ch] : 1.0;
        const bool zero_ok = (!has_m || mk > 0.0) && a.q > 0.0;  // 0/m/q == +0.0 exactly: skip the divisions
        const int32_t *__restrict__ rec = a.rec + ((size_t)img * 3 + ch) * cpl;
        double vaa = 0.0, vad = 0.0, vda = 0.0, vdd = 0.0;
        if (bi < a.band_h && bj < a.band_w) {
            const int32_t rad = rec[(size_t)bi * a.enc_w + a.off_w + bj];
            const int32_t rda = rec[(size_t)(a.off_h + bi) * a.enc_w + bj];
            const int32_t rdd = rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj];
            if (a.first) {
                const int32_t raa = rec[(size_t)bi * a.enc_w + bj];
                vaa = (raa == 0 && zero_ok) ? 0.0 : dequant(raa, mk, a.q, has_m);
            } else {
                vaa = a.a_in[((size_t)img * 3 + ch) * apl + (size_t)bi * a.a_w + bj];
            }
            vad = (rad == 0 && zero_ok) ? 0.0 : dequant(rad, mk, a.q, has_m);
            vda = (rda == 0 && zero_ok) ? 0.0 : dequant(rda, mk, a.q, has_m);
            vdd = (rdd == 0 && zero_ok) ? 0.0 : dequant(rdd, mk, a.q, has_m);
        }
        s_b[ch][0][r][cidx] = vaa; s_b[ch][1][r][cidx] = vad; s_b[ch][2][r][cidx] = vda; s_b[ch][3][r][cidx] = vdd;
    }
    __syncthreads();

    const int nn = tid & (IW_TW - 1), half = tid / IW_TW;
    const int n = n0 + nn, np = n & 1, cl = nn / 2;
    double tlo[HF], thi[HF];
#pragma unroll
    for (int s = 0; s < HF; s++) {
        tlo[s] = a.lo[np + F - 2 - 2 * s];
        thi[s] = a.hi[np + F - 2 - 2 * s];
    }
    double wl[3][HF], wh[3][HF];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
#pragma unroll
        for (int s = 0; s < HF; s++) { wl[ch][s] = 0.0; wh[ch][s] = 0.0; }
    double *__restrict__ out = a.out + (size_t)img * 3 * opl;
    const int rbase = half * (IWC_TH / 4);
#pragma unroll
    for (int rr = 0; rr < KHH; rr++) {
        const int r = rbase + rr;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {  // axis -1 synthesis of band row r at output column n (order: k_idwt_level)
            double ta = 0.0, td = 0.0, ua = 0.0, ud = 0.0;
#pragma unroll
            for (int j = 0; j < HF; j++) {
                const int s = HF - 1 - j;
                constexpr uint32_t PAIR = 3u;
                const bool lnz = ((LOM >> (F - 2 - 2 * s)) & PAIR) != 0, hnz = ((HIM >> (F - 2 - 2 * s)) & PAIR) != 0;
                if (lnz) {
                    ta += s_b[ch][0][r][cl + s] * tlo[s];
                    ua += s_b[ch][2][r][cl + s] * tlo[s];
                }
                if (hnz) {
                    td += s_b[ch][1][r][cl + s] * thi[s];
                    ud += s_b[ch][3][r][cl + s] * thi[s];
                }
            }
            const double tl = (0.0 + ta) + td, th = (0.0 + ua) + ud;
#pragma unroll
            for (int s = 0; s < HF - 1; s++) { wl[ch][s] = wl[ch][s + 1]; wh[ch][s] = wh[ch][s + 1]; }
            wl[ch][HF - 1] = tl;
            wh[ch][HF - 1] = th;
        }
        if (rr >= HF - 1) {
            const int m = 2 * (kh0 + r - (HF - 1));
#pragma unroll
            for (int mp = 0; mp < 2; mp++) {
                double px[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    double sa = 0.0, sd = 0.0;
#pragma unroll
                    for (int j = 0; j < HF; j++) {
                        const int s = HF - 1 - j;
                        const bool lnz = (LOM >> (mp + F - 2 - 2 * s)) & 1u, hnz = (HIM >> (mp + F - 2 - 2 * s)) & 1u;
                        if (lnz) sa += wl[ch][s] * a.lo[mp + F - 2 - 2 * s];
                        if (hnz) sd += wh[ch][s] * a.hi[mp + F - 2 - 2 * s];
                    }
                    px[ch] = (0.0 + sa) + sd;
                }
                if (m + mp < a.out_h && n < a.out_w) {
                    double w0, w1, w2;
                    color3_px(a.col, s_pw, px[0], px[1], px[2], w0, w1, w2);
                    const size_t o = (size_t)(m + mp) * a.out_w + n;
                    out[o] = w0;
                    out[o + opl] = w1;
                    out[o + 2 * opl] = w2;
                }
            }
        }
    }
}

// ---- host launchers -----------------------------------------------------------------------------

template <int F, uint32_t LOM, uint32_t HIM>
static int launch_dwt_FM(DwtKArgs a, int planes, hipStream_t st) {
    a.planes = planes;
    a.ov_h = a.out_h;
    a.ov_w = a.out_w;
    if (a.f32) {
        uint32_t ntf = (uint32_t)((a.out_w + DW_TW - 1) / DW_TW) * (uint32_t)((a.out_h + DW32_TH - 1) / DW32_TH) * (uint32_t)planes;
        hipLaunchKernelGGL((k_dwt_level_f32<F, LOM, HIM>), dim3(ntf), dim3(DW_BLOCK), 0, st, a);
        return (int)hipGetLastError();
    }
    if (a.color) {  // level 1 of a 3-channel image, colour model change on the loads
        constexpr int SW = (256 - (F - 2)) / 2;
        int z = 0;
        while (z < F && a.lo[z] == 0.0 && a.hi[z] == 0.0) z++;
        if (a.mode != 4) a.ov_h = min(a.out_h, (a.in_h + z + 2) / 2);
        if (a.mode != 4) a.ov_w = min(a.out_w, (a.in_w + z + 2) / 2);
        const uint32_t gx = (uint32_t)((a.out_w + SW - 1) / SW), gy = (uint32_t)((a.out_h + C1_ROWS - 1) / C1_ROWS);
        hipLaunchKernelGGL((k_dwt1_color<F, LOM, HIM>), dim3(gx * gy * (uint32_t)(planes / 3)), dim3(256), 0, st, a, gx, gy);
        return (int)hipGetLastError();
    }
    // Outputs summed in PyWavelets' overhang order: those with jb = 2o+1-N >= 0 (constant-edge mode keeps ascending order).
    // Inputs shorter than the filter take the same order -- right-hand extension taps first, nearest first, then ascending
    // through the signal and on into the left-hand extension (checked against PyWavelets: tests/golden/short_pywt.npz).
    // The order is tap jb, jb-1, ..., 0, jb+1, ...: with z leading taps that are zero in both filters it gives the same
    // bits as ascending order until jb >= z+2 (a zero tap adds nothing and the first two non-zero terms commute).
    int z = 0;
    while (z < F && a.lo[z] == 0.0 && a.hi[z] == 0.0) z++;
    if (a.mode != 4) a.ov_h = min(a.out_h, (a.in_h + z + 2) / 2);
    if (a.mode != 4) a.ov_w = min(a.out_w, (a.in_w + z + 2) / 2);
    const uint32_t nt = (uint32_t)((a.out_w + DW_TW - 1) / DW_TW) * (uint32_t)((a.out_h + DW_TH - 1) / DW_TH) * (uint32_t)planes;
    hipLaunchKernelGGL((k_dwt_level<F, LOM, HIM>), dim3(nt), dim3(DWF_BLOCK), 0, st, a);
    const int n_edge = (a.out_h - a.ov_h) * a.out_w + a.ov_h * (a.out_w - a.ov_w);
    if (n_edge > 0) hipLaunchKernelGGL(k_dwt_edge<F>, dim3((n_edge + 255) / 256, planes), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
// specialised for the zero-tap pattern of the known filter bank of that length, generic otherwise
template <int F, uint32_t LOM, uint32_t HIM>
static int launch_dwt_F(const DwtKArgs &a, int planes, hipStream_t st) {
    uint32_t lom = 0, him = 0;
    for (int j = 0; j < F; j++) {
        if (a.lo[j] != 0.0) lom |= 1u << j;
        if (a.hi[j] != 0.0) him |= 1u << j;
    }
    if (lom == LOM && him == HIM) return launch_dwt_FM<F, LOM, HIM>(a, planes, st);
    return launch_dwt_FM<F, (1u << F) - 1u, (1u << F) - 1u>(a, planes, st);
}
template <int F, uint32_t LOM, uint32_t HIM>
static int launch_idwt_FM(IdwtKArgs a, int planes, hipStream_t st, TileCtr *tc) {
    a.planes = planes;
    if (a.color) {  // level 1 of a 3-channel image, colour model change on the stores
        uint32_t ntc = (uint32_t)((a.out_w + IW_TW - 1) / IW_TW) * (uint32_t)((a.out_h + IWC_TH - 1) / IWC_TH) * (uint32_t)(planes / 3);
        hipLaunchKernelGGL((k_idwt1_color<F, LOM, HIM>), dim3(ntc), dim3(DW_BLOCK), 0, st, a);
        return (int)hipGetLastError();
    }
    const uint32_t gx = (uint32_t)((a.out_w + IW_TW - 1) / IW_TW), gy = (uint32_t)((a.out_h + IW_TH - 1) / IW_TH);
    const uint32_t nt = gx * gy * (uint32_t)planes;
    // a level with several tiles per workgroup slot (20 000 tiles and more: at 1080p, level 1 of 10 images): persistent
    // workgroups that fetch a tile ahead
    constexpr uint32_t pf_min = 20000u;
    const int num_cu = tc ? tc->num_cu : 0;
    if (tc && tc->dev && num_cu >= 2 && nt >= pf_min && (uint64_t)(a.out_h + IW_TH) * a.out_w * 8u < (1ull << 31)) {  // (>= 8 workgroups: one per tile range at least)
        const int g = tc->wg_per_cu > 0 ? tc->wg_per_cu : IWP_WG;
        const uint32_t G = (uint32_t)(num_cu * g);
        // A caller that asks for g < IWP_WG workgroups per CU wants the rest of the CU for somebody else -- the pipelined
        // schedule's list decoder, whose workgroup finds no registers beside four of these.  The grid is exactly g per CU,
        // but where the workgroups land is the dispatcher's business: a CU that took g + 1 while the decoder's were arriving
        // keeps a decoder workgroup waiting for the whole level (13.9 instead of 9.3 ms).  So the workgroups
        // are made too large for a (g + 1)-th: LDS they do not use (dynamic, behind the kernel's own), just over 1 / (g + 1)
        // of the CU's.  The placement is then the same whoever arrives first -- no timer between the launches.
        uint32_t pad = 0;
        if (g < IWP_WG && tc->lds_per_cu > 0) {
            constexpr int HFk = F / 2;
            constexpr uint32_t own = 4u * (IW_TH / 2 + HFk - 1) * (IW_TW / 2 + HFk) * 8u + 64u;  // s_b + s_next, rounded up
            const uint32_t want = (uint32_t)tc->lds_per_cu / (uint32_t)(g + 1) + 512u;
            if (want > own && want <= 65536u) pad = want - own;
        }
        TileBase cb;
        uint32_t *ctr = tc->dev;
        for (uint32_t x = 0; x < 8; x++) {
            cb.v[x] = tc->base[x];
            // XCD x draws one number per tile of its range and two more per workgroup (the look-ahead past the end)
            tc->base[x] += (nt >> 3) + (x < (nt & 7u) ? 1u : 0u) + 2u * ((G + 7u - x) >> 3);
        }
        tc->started += G;
        if (a.first) hipLaunchKernelGGL((k_idwt_level_pf<F, LOM, HIM, true>), dim3(G), dim3(DW_BLOCK), pad, st, a, gx, gy, ctr, cb);
        else if (a.flags && (uint64_t)a.enc_h * a.enc_w * 4u < (1ull << 31))
            hipLaunchKernelGGL((k_idwt_level_pf<F, LOM, HIM, false, true>), dim3(G), dim3(DW_BLOCK), pad, st, a, gx, gy, ctr, cb);
        else { a.flags = nullptr; hipLaunchKernelGGL((k_idwt_level_pf<F, LOM, HIM, false>), dim3(G), dim3(DW_BLOCK), pad, st, a, gx, gy, ctr, cb); }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {  // the launch did not happen: counters and book-keeping start over together
            (void)hipMemsetAsync(tc->dev, 0, TILECTR_WORDS * sizeof(uint32_t), st);
            for (uint32_t x = 0; x < 8; x++) tc->base[x] = 0;
            tc->started = 0;
        }
        return (int)e;
    }
    hipLaunchKernelGGL((k_idwt_level<F, LOM, HIM>), dim3(nt), dim3(DW_BLOCK), 0, st, a);
    return (int)hipGetLastError();
}
template <int F, uint32_t LOM, uint32_t HIM>
static int launch_idwt_F(const IdwtKArgs &a, int planes, hipStream_t st, TileCtr *tc) {
    uint32_t lom = 0, him = 0;
    for (int j = 0; j < F; j++) {
        if (a.lo[j] != 0.0) lom |= 1u << j;
        if (a.hi[j] != 0.0) him |= 1u << j;
    }
    if (lom == LOM && him == HIM) return launch_idwt_FM<F, LOM, HIM>(a, planes, st, tc);
    return launch_idwt_FM<F, (1u << F) - 1u, (1u << F) - 1u>(a, planes, st, tc);
}

extern "C" int spiht_launch_dwt_level(const DwtKArgs *a, int planes, hipStream_t st) {
    switch (a->F) {
    case 2: return launch_dwt_F<2, 0x3u, 0x3u>(*a, planes, st);            // haar
    case 6: return launch_dwt_F<6, 0x3Eu, 0x0Eu>(*a, planes, st);          // bior2.2
    case 10: return launch_dwt_F<10, 0x3FEu, 0x0FEu>(*a, planes, st);      // bior4.4
    case 18: return launch_dwt_F<18, 0x3FFFEu, 0x3FF8u>(*a, planes, st);   // bior6.8
    // every other even length up to SPIHT_MAX_TAPS (db / sym / coif / the other bior and rbio banks): all taps taken
    case 4: return launch_dwt_F<4, 0xFu, 0xFu>(*a, planes, st);
    case 8: return launch_dwt_F<8, 0xFFu, 0xFFu>(*a, planes, st);
    case 12: return launch_dwt_F<12, 0xFFFu, 0xFFFu>(*a, planes, st);
    case 14: return launch_dwt_F<14, 0x3FFFu, 0x3FFFu>(*a, planes, st);
    case 16: return launch_dwt_F<16, 0xFFFFu, 0xFFFFu>(*a, planes, st);
    case 20: return launch_dwt_F<20, 0xFFFFFu, 0xFFFFFu>(*a, planes, st);
    default: return -1;
    }
}
// tc: tile counters of the calling context (nullptr: fixed-stride tile order in the persistent kernel)
extern "C" int spiht_launch_idwt_level(const IdwtKArgs *a, int planes, hipStream_t st, TileCtr *tc) {
    switch (a->F) {
    case 2: return launch_idwt_F<2, 0x3u, 0x3u>(*a, planes, st, tc);            // haar
    case 6: return launch_idwt_F<6, 0x0Eu, 0x3Eu>(*a, planes, st, tc);          // bior2.2 rec_lo / rec_hi
    case 10: return launch_idwt_F<10, 0x0FEu, 0x3FEu>(*a, planes, st, tc);      // bior4.4
    case 18: return launch_idwt_F<18, 0x3FF8u, 0x3FFFEu>(*a, planes, st, tc);   // bior6.8
    case 4: return launch_idwt_F<4, 0xFu, 0xFu>(*a, planes, st, tc);
    case 8: return launch_idwt_F<8, 0xFFu, 0xFFu>(*a, planes, st, tc);
    case 12: return launch_idwt_F<12, 0xFFFu, 0xFFFu>(*a, planes, st, tc);
    case 14: return launch_idwt_F<14, 0x3FFFu, 0x3FFFu>(*a, planes, st, tc);
    case 16: return launch_idwt_F<16, 0xFFFFu, 0xFFFFu>(*a, planes, st, tc);
    case 20: return launch_idwt_F<20, 0xFFFFFu, 0xFFFFFu>(*a, planes, st, tc);
    default: return -1;
    }
}
// pad strips of coeffs_to_array for `L` levels: hs/ws band sizes and offh/offw block offsets (index 1..L)
extern "C" int spiht_launch_zero_pads(int L, const int64_t *hs, const int64_t *ws, const int64_t *offh, const int64_t *offw,
                                      int enc_h, int enc_w, int32_t *coeffs, int planes, hipStream_t st) {
    PadRects pr;
    pr.n = 0; pr.enc_h = enc_h; pr.enc_w = enc_w; pr.pad = 0;
    for (int l = 1; l <= L; l++) {
        if (offh[l] > hs[l]) {  // below 'ad'
            pr.r0[pr.n] = (int)hs[l]; pr.r1[pr.n] = (int)offh[l]; pr.c0[pr.n] = (int)offw[l]; pr.c1[pr.n] = (int)(offw[l] + ws[l]);
            pr.n++;
        }
        if (offw[l] > ws[l]) {  // right of 'da'
            pr.r0[pr.n] = (int)offh[l]; pr.r1[pr.n] = (int)(offh[l] + hs[l]); pr.c0[pr.n] = (int)ws[l]; pr.c1[pr.n] = (int)offw[l];
            pr.n++;
        }
    }
    if (pr.n == 0) return 0;
    hipLaunchKernelGGL(k_zero_pads, dim3(8, pr.n, planes), dim3(256), 0, st, pr, coeffs);
    return (int)hipGetLastError();
}
extern "C" int spiht_launch_quant_plain(const double *in, int32_t *out, size_t n_per_plane, int planes, int c,
                                        const double *mults, double q, uint32_t *maxabs, hipStream_t st) {
    hipLaunchKernelGGL(k_quant_plain, dim3(1024), dim3(256), 0, st, in, out, n_per_plane, planes, c, mults, q, maxabs);
    return (int)hipGetLastError();
}
extern "C" int spiht_launch_dequant_plain(const int32_t *in, double *out, size_t n_per_plane, int planes, int c,
                                          const double *mults, double q, hipStream_t st) {
    hipLaunchKernelGGL(k_dequant_plain, dim3(1024), dim3(256), 0, st, in, out, n_per_plane, planes, c, mults, q);
    return (int)hipGetLastError();
}
