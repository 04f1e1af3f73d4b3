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

#ifndef DW_TH
#define DW_TH 12      // output rows per tile.  Level 1 of 256 1080p images: 12 rows 3.8-3.95 ms, 16 rows 4.05-4.2, 14: 4.1,
#endif                // 10: 4.4, 20: 4.5, 24: 4.2 (25.9 KB of LDS per workgroup at 12 rows: six workgroups per CU)
#define DW32_TH 16    // ... of the single-precision kernel (12 rows: 3.9 instead of 2.85 ms)
#ifndef DW_TW
#define DW_TW 64      // output cols per tile
#endif
#define DW_BLOCK 256

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Remap the linear block id so
// that each XCD walks one contiguous range of tiles: neighbouring tiles (shared halo columns/rows on the read
// side, shared partial cache lines on the write side) then meet in the same L2.  Speed only, never correctness.
__device__ __forceinline__ void xcd_tile(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t &bx, uint32_t &by, uint32_t &bz) {
    const uint32_t nt = gx * gy * gz, L = blockIdx.x;
    const uint32_t q = nt >> 3, r = nt & 7u, x = L & 7u, j = L >> 3;
    const uint32_t T = x * q + (x < r ? x : r) + j;
    bx = T % gx;
    const uint32_t t2 = T / gx;
    by = t2 % gy;
    bz = t2 / gy;
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

// grid: (ceil(out_w/TW), ceil(out_h/TH), planes).  LOM / HIM: bit j set = tap j of dec_lo / dec_hi is non-zero;
// a zero tap contributes exactly nothing (0*x added to the running sum), so skipping it changes no bit and
// removes a third (bior2.2) to a fifth of the float64 arithmetic.
// EDGE: the instantiation for the tiles that hold the bottom / right overhang outputs, which are summed in PyWavelets'
// order (below); the interior tiles run the instantiation without that code (it costs the fast path 10 % when it is
// merely present: measured).  The two launches cover the tile grid between them (DwtKArgs::et_*).
template <int F, uint32_t LOM, uint32_t HIM, bool EDGE>
__global__ __launch_bounds__(DW_BLOCK) void k_dwt_level(DwtKArgs a) {
    constexpr int NC = 2 * DW_TW + F - 2;  // input columns needed by the tile
    constexpr int NR = 2 * DW_TH + F - 2;  // input rows needed
    constexpr int HC = (NC + 1) / 2;       // columns per parity plane
    static_assert(NC <= DW_BLOCK, "one thread per input column");
    // two column-parity planes; the padding makes the plane stride an odd multiple of 16 banks, so the even and odd
    // lanes of one ds_write_b64 lane group land on different banks
    constexpr int RS = HC + 1, PS = DW_TH * RS + (24 - (DW_TH * RS) % 16) % 16;  // PS % 16 == 8
    __shared__ double s_lo[2][PS];
    __shared__ double s_hi[2][PS];
    __shared__ int s_row[NR];
    uint32_t tbx, tby, tbz;
    if (!EDGE) {
        xcd_tile((uint32_t)a.et_x, (uint32_t)a.et_y, a.planes, tbx, tby, tbz);  // tiles left of et_x and above et_y
    } else {
        // the other tiles of a plane, numbered: the column strip right of et_x (all tile rows), then the rest of the
        // bottom strip
        const uint32_t gx = (a.out_w + DW_TW - 1) / DW_TW, gy = (a.out_h + DW_TH - 1) / DW_TH;
        const uint32_t sw = gx - (uint32_t)a.et_x, ns = sw * gy, ne = ns + (uint32_t)a.et_x * (gy - (uint32_t)a.et_y);
        uint32_t e, dummy;
        xcd_tile(ne, 1, a.planes, e, dummy, tbz);
        if (e < ns) { tbx = (uint32_t)a.et_x + e % sw; tby = e / sw; }
        else { e -= ns; tbx = e % (uint32_t)a.et_x; tby = (uint32_t)a.et_y + e / (uint32_t)a.et_x; }
    }
    const int plane = (int)tbz;
    const int oh0 = (int)tby * DW_TH, ow0 = (int)tbx * DW_TW;
    const double *__restrict__ in = a.in + (size_t)plane * a.in_h * a.in_w;
    const int tid = threadIdx.x;

    // input row needed for output row o, tap j: 2*o + 1 - j ; first needed row r0 = 2*oh0 + 1 - (F-1)
    const int r0 = 2 * oh0 + 2 - F, c0 = 2 * ow0 + 2 - F;
    if (tid < NR) s_row[tid] = ext_index(r0 + tid, a.in_h, a.mode);
    __syncthreads();

    // ---- axis -2: thread <-> input column; all loads first, then the filter ----
    if (tid < NC) {
        const int gc = ext_index(c0 + tid, a.in_w, a.mode);
        double x[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int gr = s_row[r];
            x[r] = (gc < 0 || gr < 0) ? 0.0 : in[(size_t)gr * a.in_w + gc];
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
        // Bottom overhang, PyWavelets' order (downsampling_convolution): for an output that hangs over the end of the
        // input (2o+1 >= N) the taps that read the signal extension come first, nearest first (tap 2o+1-N down to 0),
        // then the others ascending.  Only the last row or two of a level differ in a bit from the ascending sums above
        // (ov_h: set by the launcher), so only the bottom tile row comes here (a block-uniform branch); the samples are
        // loaded again -- run-time tap order would turn x[] into scratch memory.
        if (EDGE && oh0 + DW_TH > a.ov_h) {
            for (int o = max(a.ov_h - oh0, 0); o < DW_TH && oh0 + o < a.out_h; o++) {
                const int jb = 2 * (oh0 + o) + 1 - a.in_h;
                double sl = 0.0, shh = 0.0;
                for (int s2 = 0; s2 < F; s2++) {
                    const int j = s2 <= jb ? jb - s2 : s2;
                    const int gr = s_row[2 * o + F - 1 - j];
                    const double v = (gc < 0 || gr < 0) ? 0.0 : in[(size_t)gr * a.in_w + gc];
                    sl += a.lo[j] * v;
                    shh += a.hi[j] * v;
                }
                s_lo[par][o * RS + hc] = sl;
                s_hi[par][o * RS + hc] = shh;
            }
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
#pragma unroll
    for (int u = 0; u < (DW_TH * DW_TW + DW_BLOCK - 1) / DW_BLOCK; u++) {
        const int p = tid + u * DW_BLOCK;
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
        if (EDGE && ow >= a.ov_w) {  // right overhang: the same order along this axis (the last column or two of a level)
            const int jb = 2 * ow + 1 - a.in_w;
            aa = 0.0; ad = 0.0; da = 0.0; dd = 0.0;
            for (int s2 = 0; s2 < F; s2++) {
                const int j = s2 <= jb ? jb - s2 : s2;
                const double vl = s_lo[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                const double vh = s_hi[(F - 1 - j) & 1][o * RS + wcol + ((F - 1 - j) >> 1)];
                aa += a.lo[j] * vl; da += a.lo[j] * vh;
                ad += a.hi[j] * vl; dd += a.hi[j] * vh;
            }
        }
        const int32_t qad = quant(ad, mk, a.q, has_m), qda = quant(da, mk, a.q, has_m), qdd = quant(dd, mk, a.q, has_m);
        if (a.last) {
            const int32_t qaa = quant(aa, mk, a.q, has_m);
            co[(size_t)oh * a.enc_w + ow] = qaa;
            amax = max(amax, iabs_u(qaa));
        } else {
            llo[(size_t)oh * a.out_w + ow] = aa;
        }
        co[(size_t)oh * a.enc_w + a.off_w + ow] = qad;                 // 'ad' top-right
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = qda;               // 'da' bottom-left
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = qdd;     // 'dd' bottom-right
        amax = max(amax, max(iabs_u(qad), max(iabs_u(qda), iabs_u(qdd))));
    }
    if (a.maxabs != nullptr) {
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
        if ((tid & 63) == 0 && amax) atomicMax(&a.maxabs[plane / a.c], amax);
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
    float flo[F], fhi[F];
#pragma unroll
    for (int j = 0; j < F; j++) { flo[j] = (float)a.lo[j]; fhi[j] = (float)a.hi[j]; }

    const int r0 = 2 * oh0 + 2 - F, c0 = 2 * ow0 + 2 - F;
    if (tid < NR) s_row[tid] = ext_index(r0 + tid, a.in_h, a.mode);
    if (tid < F) { s_flo[tid] = (float)a.lo[tid]; s_fhi[tid] = (float)a.hi[tid]; }
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
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
        if ((tid & 63) == 0 && amax) atomicMax(&a.maxabs[plane / a.c], amax);
    }
}

#ifdef SPIHT_DIAG  // experiment kept for reference (equal at best), not in the product build
// ---- row-marching variant of the forward level ------------------------------------------------------------------
// A workgroup owns a strip of MW_SW output columns and marches down MW_ROWS output rows: thread = input column keeps
// the F rows its column filter needs in registers (two new rows per step, prefetched MW_PF steps ahead, so every
// input sample is loaded exactly once per strip), the low/high outputs of the step go through a double-buffered LDS
// row to the axis -1 filter: threads 0..127 produce (aa, ad) from the low row, threads 128..255 (da, dd) from the
// high row.  One barrier per output row; loads are a continuous stream with no vertical halo.
#define MW_PF 8      // steps of look-ahead for the two rows a step loads
#define MW_ROWS 136  // output rows per workgroup
template <int F, uint32_t LOM, uint32_t HIM>
__global__ __launch_bounds__(256) void k_dwt_march(DwtKArgs a, uint32_t gx, uint32_t gy) {
    constexpr int SW = (256 - (F - 2)) / 2;  // output columns per strip: exactly 256 input columns
    constexpr int HC = 128 + 2;
    __shared__ double s_lo[2][2][HC];        // [step parity][column parity][column >> 1]
    __shared__ double s_hi[2][2][HC];
    uint32_t tbx, tby, tbz;
    xcd_tile(gx, gy, a.planes, tbx, tby, tbz);
    const int plane = (int)tbz;
    const int ow0 = (int)tbx * SW, oa = (int)tby * MW_ROWS;
    const int ob = min(oa + MW_ROWS, a.out_h);
    const double *__restrict__ in = a.in + (size_t)plane * a.in_h * a.in_w;
    const int tid = threadIdx.x;
    const int gc = ext_index(2 * ow0 + 2 - F + tid, a.in_w, a.mode);
    auto ld = [&](int r) -> double {
        const int gr = ext_index(r, a.in_h, a.mode);
        return (gc < 0 || gr < 0) ? 0.0 : in[(size_t)gr * a.in_w + gc];
    };
    // window: rows 2o+2-F .. 2o+1 of output row o
    double win[F];
#pragma unroll
    for (int t = 0; t < F; t++) win[t] = ld(2 * oa + 2 - F + t);
    double pq[MW_PF][2];
#pragma unroll
    for (int u = 0; u < MW_PF; u++) { pq[u][0] = ld(2 * (oa + 1 + u)); pq[u][1] = ld(2 * (oa + 1 + u) + 1); }

    const int k = plane % a.c;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    int32_t *__restrict__ co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
    double *__restrict__ llo = a.last ? nullptr : a.ll_out + (size_t)plane * a.out_h * a.out_w;
    const int role = tid >> 7, wcol = tid & 127;
    const int ow = ow0 + wcol;
    const bool wr = wcol < SW && ow < a.out_w;
    uint32_t amax = 0;
    const int par = tid & 1, hc = tid >> 1;

    for (int o0 = oa; o0 < ob; o0 += MW_PF) {
#pragma unroll
        for (int u = 0; u < MW_PF; u++) {
            const int o = o0 + u;
            // ---- axis -2 for output row o ----
            double sl = 0.0, shh = 0.0;
#pragma unroll
            for (int j = 0; j < F; j++) {
                if ((LOM >> j) & 1u) sl += a.lo[j] * win[F - 1 - j];
                if ((HIM >> j) & 1u) shh += a.hi[j] * win[F - 1 - j];
            }
            s_lo[u & 1][par][hc] = sl;
            s_hi[u & 1][par][hc] = shh;
            // slide the window and refill the look-ahead slot
#pragma unroll
            for (int t = 0; t < F - 2; t++) win[t] = win[t + 2];
            win[F - 2] = pq[u][0];
            win[F - 1] = pq[u][1];
            pq[u][0] = ld(2 * (o + 1 + MW_PF));
            pq[u][1] = ld(2 * (o + 1 + MW_PF) + 1);
            __syncthreads();
            // ---- axis -1: two sub-bands per thread ----
            if (wr && o < ob) {
                const double(*src)[HC] = role ? s_hi[u & 1] : s_lo[u & 1];
                double r0 = 0.0, r1 = 0.0;  // role 0: aa, ad   role 1: da, dd
#pragma unroll
                for (int j = 0; j < F; j++) {
                    const double v = src[(F - 1 - j) & 1][wcol + ((F - 1 - j) >> 1)];
                    if ((LOM >> j) & 1u) r0 += a.lo[j] * v;
                    if ((HIM >> j) & 1u) r1 += a.hi[j] * v;
                }
                const int32_t q1 = quant(r1, mk, a.q, has_m);
                amax = max(amax, iabs_u(q1));
                if (role == 0) {
                    if (a.last) {
                        const int32_t q0 = quant(r0, mk, a.q, has_m);
                        co[(size_t)o * a.enc_w + ow] = q0;
                        amax = max(amax, iabs_u(q0));
                    } else {
                        llo[(size_t)o * a.out_w + ow] = r0;
                    }
                    co[(size_t)o * a.enc_w + a.off_w + ow] = q1;                      // 'ad' top-right
                } else {
                    const int32_t q0 = quant(r0, mk, a.q, has_m);
                    amax = max(amax, iabs_u(q0));
                    co[(size_t)(a.off_h + o) * a.enc_w + ow] = q0;                     // 'da' bottom-left
                    co[(size_t)(a.off_h + o) * a.enc_w + a.off_w + ow] = q1;           // 'dd' bottom-right
                }
            }
        }
    }
    if (a.maxabs != nullptr) {
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
        if ((tid & 63) == 0 && amax) atomicMax(&a.maxabs[plane / a.c], amax);
    }
}
#endif  // SPIHT_DIAG

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
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int plane = (int)(t / n_per_plane);
        bool has_m = mults != nullptr;
        int32_t v = quant(in[t], has_m ? mults[plane % c] : 1.0, q, has_m);
        out[t] = v;
        if (maxabs != nullptr && v != 0) atomicMax(&maxabs[plane / c], iabs_u(v));
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
#ifndef IW_TH
#define IW_TH 16    // output rows per tile (two halves of 8, one per half of the workgroup; a multiple of 4).  Level 1 of 256
#endif              // 1080p images: 16 rows 4.02 ms, 20: 4.05, 24: 4.03, 32: 4.19, 12: 4.22, 40: 4.67, 8: 4.74
#define IW_TW 128   // output cols per tile, one thread per column per half

__device__ __forceinline__ double dequant(int32_t r, double m, double q, bool has_m) {
    double v = (double)r;
    if (has_m) v = v / m;
    return v / q;
}

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

    for (int p = tid; p < KH * KW; p += DW_BLOCK) {
        const int r = p / KW, cidx = p - r * KW;
        const int bi = kh0 + r, bj = kw0 + cidx;
        double vaa = 0.0, vad = 0.0, vda = 0.0, vdd = 0.0;
        if (bi < a.band_h && bj < a.band_w) {
#ifdef IDWT_EXP_SKIP  // timing experiment only (wrong results): what the inverse transform costs without its detail-band reads
            const bool skp = a.band_h > IDWT_EXP_SKIP;
            const int32_t rad = skp ? 0 : rec[(size_t)bi * a.enc_w + a.off_w + bj];
            const int32_t rda = skp ? 0 : rec[(size_t)(a.off_h + bi) * a.enc_w + bj];
            const int32_t rdd = skp ? 0 : rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj];
#else
            const int32_t rad = rec[(size_t)bi * a.enc_w + a.off_w + bj];
            const int32_t rda = rec[(size_t)(a.off_h + bi) * a.enc_w + bj];
            const int32_t rdd = rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj];
#endif
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
#ifdef SPIHT_DIAG
    static const int use_march = [] { const char *e = getenv("SPIHT_DWT_MARCH"); return e ? atoi(e) : 0; }();
    if (use_march) {
        constexpr int SW = (256 - (F - 2)) / 2;
        const uint32_t gx = (uint32_t)((a.out_w + SW - 1) / SW), gy = (uint32_t)((a.out_h + MW_ROWS - 1) / MW_ROWS);
        hipLaunchKernelGGL((k_dwt_march<F, LOM, HIM>), dim3(gx * gy * (uint32_t)planes), dim3(256), 0, st, a, gx, gy);
        return (int)hipGetLastError();
    }
#endif
    // Outputs summed in PyWavelets' overhang order: those with jb = 2o+1-N >= 0 on an axis whose input is at least as
    // long as the filter (constant-edge mode keeps ascending order; shorter inputs go through another loop of pywt's).
    // The order is tap jb, jb-1, ..., 0, jb+1, ...: with z leading taps that are zero in both filters it gives the same
    // bits as ascending order until jb >= z+2 (a zero tap adds nothing and the first two non-zero terms commute).
    int z = 0;
    while (z < F && a.lo[z] == 0.0 && a.hi[z] == 0.0) z++;
    if (a.mode != 4 && a.in_h >= F) a.ov_h = min(a.out_h, (a.in_h + z + 2) / 2);
    if (a.mode != 4 && a.in_w >= F) a.ov_w = min(a.out_w, (a.in_w + z + 2) / 2);
    const int gx = (a.out_w + DW_TW - 1) / DW_TW, gy = (a.out_h + DW_TH - 1) / DW_TH;
    a.et_x = min(gx, a.ov_w / DW_TW);  // first tile column / row that holds such outputs (gx / gy: none)
    a.et_y = min(gy, a.ov_h / DW_TH);
    if (a.ov_w >= a.out_w) a.et_x = gx;
    if (a.ov_h >= a.out_h) a.et_y = gy;
    const uint32_t n_in = (uint32_t)a.et_x * (uint32_t)a.et_y, n_edge = (uint32_t)gx * (uint32_t)gy - n_in;
    if (n_in) hipLaunchKernelGGL((k_dwt_level<F, LOM, HIM, false>), dim3(n_in * (uint32_t)planes), dim3(DW_BLOCK), 0, st, a);
    if (n_edge) hipLaunchKernelGGL((k_dwt_level<F, LOM, HIM, true>), dim3(n_edge * (uint32_t)planes), dim3(DW_BLOCK), 0, st, a);
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
static int launch_idwt_FM(IdwtKArgs a, int planes, hipStream_t st) {
    a.planes = planes;
    uint32_t nt = (uint32_t)((a.out_w + IW_TW - 1) / IW_TW) * (uint32_t)((a.out_h + IW_TH - 1) / IW_TH) * (uint32_t)planes;
    hipLaunchKernelGGL((k_idwt_level<F, LOM, HIM>), dim3(nt), dim3(DW_BLOCK), 0, st, a);
    return (int)hipGetLastError();
}
template <int F, uint32_t LOM, uint32_t HIM>
static int launch_idwt_F(const IdwtKArgs &a, int planes, hipStream_t st) {
    uint32_t lom = 0, him = 0;
    for (int j = 0; j < F; j++) {
        if (a.lo[j] != 0.0) lom |= 1u << j;
        if (a.hi[j] != 0.0) him |= 1u << j;
    }
    if (lom == LOM && him == HIM) return launch_idwt_FM<F, LOM, HIM>(a, planes, st);
    return launch_idwt_FM<F, (1u << F) - 1u, (1u << F) - 1u>(a, planes, st);
}

extern "C" int spiht_launch_dwt_level(const DwtKArgs *a, int planes, hipStream_t st) {
    switch (a->F) {
    case 2: return launch_dwt_F<2, 0x3u, 0x3u>(*a, planes, st);            // haar
    case 6: return launch_dwt_F<6, 0x3Eu, 0x0Eu>(*a, planes, st);          // bior2.2
    case 10: return launch_dwt_F<10, 0x3FEu, 0x0FEu>(*a, planes, st);      // bior4.4
    case 18: return launch_dwt_F<18, 0x3FFFEu, 0x3FF8u>(*a, planes, st);   // bior6.8
    default: return -1;
    }
}
extern "C" int spiht_launch_idwt_level(const IdwtKArgs *a, int planes, hipStream_t st) {
    switch (a->F) {
    case 2: return launch_idwt_F<2, 0x3u, 0x3u>(*a, planes, st);            // haar
    case 6: return launch_idwt_F<6, 0x0Eu, 0x3Eu>(*a, planes, st);          // bior2.2 rec_lo / rec_hi
    case 10: return launch_idwt_F<10, 0x0FEu, 0x3FEu>(*a, planes, st);      // bior4.4
    case 18: return launch_idwt_F<18, 0x3FF8u, 0x3FFFEu>(*a, planes, st);   // bior6.8
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
