// 2-D separable DWT / inverse DWT for the SPIHT image path (gfx950), float64.
//
// Replaces the PyWavelets calls of the reference wrapper
//   encode: pywt.wavedec2 -> coeffs_to_array -> channel_mults*arr -> quantize   (spiht_wrapper.py:163-172)
//   decode: rec/channel_mults -> dequantize -> array_to_coeffs -> waverec2       (spiht_wrapper.py:259-276)
// One launch per decomposition level, both axes fused:
//   forward: a workgroup owns a TH x TW tile of output positions of all four sub-bands.  The axis -2
//            (row-direction) filter runs straight from global memory, one input column per thread with
//            the column walked downwards so every input sample is loaded once per tile; the low/high
//            intermediates go to LDS; the axis -1 filter reads them back and writes LL as float64 (input
//            of the next level) and the three detail bands already quantised (int32, truncation toward
//            zero) into their place in the zero-padded Mallat array -- coeffs_to_array and the two
//            quantise passes of the wrapper cost no extra HBM traffic.
//   inverse: the mirror image; detail bands are dequantised on load ((rec / m_k) / q as the wrapper
//            does), axis -1 synthesis goes to LDS, axis -2 synthesis writes float64.
// Arithmetic follows the published pywt definitions in the same summation order as oracle/dwt_oracle.c
// (ascending tap index, separate multiply and add: this file is compiled with -ffp-contract=off), so
// GPU and oracle agree bit for bit and both agree with pywt to a few ulp.
// HBM-bound: per level, reads 8 B per input sample, writes 8 B (LL) + 3*4 B (details) per output position.
#include "common.h"

#define DW_TH 16      // output rows per tile
#define DW_TW 64      // output cols per tile
#define DW_BLOCK 256


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

// grid: (ceil(out_w/TW), ceil(out_h/TH), planes)
template <int F>
__global__ __launch_bounds__(DW_BLOCK) void k_dwt_level(DwtKArgs a) {
    constexpr int NC = 2 * DW_TW + F - 2;  // input columns needed by the tile
    constexpr int NR = 2 * DW_TH + F - 2;  // input rows needed
    __shared__ double s_lo[DW_TH][NC + 1];
    __shared__ double s_hi[DW_TH][NC + 1];
    const int plane = blockIdx.z;
    const int oh0 = blockIdx.y * DW_TH, ow0 = blockIdx.x * DW_TW;
    const double *in = a.in + (size_t)plane * a.in_h * a.in_w;
    const int tid = threadIdx.x;

    // ---- axis -2: thread <-> input column; sliding window down the rows ----
    // input row needed for output row o, tap j: 2*o + 1 - j ; first needed row r0 = 2*oh0 + 1 - (F-1)
    const int r0 = 2 * oh0 + 2 - F, c0 = 2 * ow0 + 2 - F;
    for (int col = tid; col < NC; col += DW_BLOCK) {
        const int gc = ext_index(c0 + col, a.in_w, a.mode);
        double win[F];  // win[t] = x~[r0 + base + t]
#pragma unroll
        for (int t = 0; t < F - 2; t++) {
            int gr = ext_index(r0 + t, a.in_h, a.mode);
            win[t] = (gc < 0 || gr < 0) ? 0.0 : in[(size_t)gr * a.in_w + gc];
        }
#pragma unroll
        for (int o = 0; o < DW_TH; o++) {
            // rows r0 + 2o + F-2, r0 + 2o + F-1 enter the window
            int gr1 = ext_index(r0 + 2 * o + F - 2, a.in_h, a.mode);
            int gr2 = ext_index(r0 + 2 * o + F - 1, a.in_h, a.mode);
            win[F - 2] = (gc < 0 || gr1 < 0) ? 0.0 : in[(size_t)gr1 * a.in_w + gc];
            win[F - 1] = (gc < 0 || gr2 < 0) ? 0.0 : in[(size_t)gr2 * a.in_w + gc];
            // out[o] = sum_j f[j] * x~[2(oh0+o)+1-j];  x~[2(oh0+o)+1-j] = win[F-1-j]
            double sl = 0.0, shh = 0.0;
#pragma unroll
            for (int j = 0; j < F; j++) {
                sl += a.lo[j] * win[F - 1 - j];
                shh += a.hi[j] * win[F - 1 - j];
            }
            s_lo[o][col] = sl;
            s_hi[o][col] = shh;
#pragma unroll
            for (int t = 0; t < F - 2; t++) win[t] = win[t + 2];
        }
    }
    __syncthreads();

    // ---- axis -1 from LDS; 4 sub-bands per output position ----
    const int k = plane % a.c;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    int32_t *co = a.coeffs + (size_t)plane * a.enc_h * a.enc_w;
    double *llo = a.last ? nullptr : a.ll_out + (size_t)plane * a.out_h * a.out_w;
    for (int p = tid; p < DW_TH * DW_TW; p += DW_BLOCK) {
        const int o = p / DW_TW, wcol = p % DW_TW;
        const int oh = oh0 + o, ow = ow0 + wcol;
        if (oh >= a.out_h || ow >= a.out_w) continue;
        // x~ index 2*ow+1-j  ->  LDS column (2*ow+1-j) - c0 = 2*wcol + F-1 - j
        double aa = 0.0, ad = 0.0, da = 0.0, dd = 0.0;
#pragma unroll
        for (int j = 0; j < F; j++) {
            const double vl = s_lo[o][2 * wcol + F - 1 - j];
            const double vh = s_hi[o][2 * wcol + F - 1 - j];
            aa += a.lo[j] * vl;
            ad += a.hi[j] * vl;
            da += a.lo[j] * vh;
            dd += a.hi[j] * vh;
        }
        if (a.last) co[(size_t)oh * a.enc_w + ow] = quant(aa, mk, a.q, has_m);
        else llo[(size_t)oh * a.out_w + ow] = aa;
        co[(size_t)oh * a.enc_w + a.off_w + ow] = quant(ad, mk, a.q, has_m);                 // 'ad' top-right
        co[(size_t)(a.off_h + oh) * a.enc_w + ow] = quant(da, mk, a.q, has_m);               // 'da' bottom-left
        co[(size_t)(a.off_h + oh) * a.enc_w + a.off_w + ow] = quant(dd, mk, a.q, has_m);     // 'dd' bottom-right
    }
}

// level 0 of the API (no decomposition): quantise the image itself. grid-stride.
__global__ __launch_bounds__(256) void k_quant_plain(const double *in, int32_t *out, size_t n_per_plane, int planes, int c,
                                                     const double *mults, double q) {
    size_t total = n_per_plane * (size_t)planes;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int plane = (int)(t / n_per_plane);
        bool has_m = mults != nullptr;
        out[t] = quant(in[t], has_m ? mults[plane % c] : 1.0, q, has_m);
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
#define IW_TH 32   // output rows per tile
#define IW_TW 64   // output cols per tile


__device__ __forceinline__ double dequant(int32_t r, double m, double q, bool has_m) {
    double v = (double)r;
    if (has_m) v = v / m;
    return v / q;
}

// grid: (ceil(out_w/TW), ceil(out_h/TH), planes)
template <int F>
__global__ __launch_bounds__(DW_BLOCK) void k_idwt_level(IdwtKArgs a) {
    // band indices k contributing to outputs n in [n0, n0+T): t = n+F-2-2k in [0,F)  ->
    //   k >= (n0-1)/2 (ceil) ... k <= (n0+T-1+F-2)/2 (floor)
    constexpr int KH = IW_TH / 2 + F / 2 + 1;   // band rows staged
    constexpr int KW = IW_TW / 2 + F / 2 + 1;   // band cols staged
    __shared__ double s_a[4][KH][KW + 1];       // aa, ad, da, dd (dequantised)
    __shared__ double s_tl[KH][IW_TW + 1];
    __shared__ double s_th[KH][IW_TW + 1];
    const int plane = blockIdx.z;
    const int k = plane % a.c;
    const int m0 = blockIdx.y * IW_TH, n0 = blockIdx.x * IW_TW;
    const int kh0 = m0 > 0 ? (m0 - 1 + 1) / 2 : 0;  // ceil((m0-1)/2) for m0>=1; 0 for m0=0
    const int kw0 = n0 > 0 ? (n0 - 1 + 1) / 2 : 0;
    const bool has_m = a.mults != nullptr;
    const double mk = has_m ? a.mults[k] : 1.0;
    const int32_t *rec = a.rec + (size_t)plane * a.enc_h * a.enc_w;
    const double *ain = a.first ? nullptr : a.a_in + (size_t)plane * a.a_h * a.a_w;
    const int tid = threadIdx.x;

    for (int p = tid; p < KH * KW; p += DW_BLOCK) {
        const int r = p / KW, cidx = p % KW;
        const int bi = kh0 + r, bj = kw0 + cidx;
        double vaa = 0.0, vad = 0.0, vda = 0.0, vdd = 0.0;
        if (bi < a.band_h && bj < a.band_w) {
            vaa = a.first ? dequant(rec[(size_t)bi * a.enc_w + bj], mk, a.q, has_m) : ain[(size_t)bi * a.a_w + bj];
            vad = dequant(rec[(size_t)bi * a.enc_w + a.off_w + bj], mk, a.q, has_m);
            vda = dequant(rec[(size_t)(a.off_h + bi) * a.enc_w + bj], mk, a.q, has_m);
            vdd = dequant(rec[(size_t)(a.off_h + bi) * a.enc_w + a.off_w + bj], mk, a.q, has_m);
        }
        s_a[0][r][cidx] = vaa; s_a[1][r][cidx] = vad; s_a[2][r][cidx] = vda; s_a[3][r][cidx] = vdd;
    }
    __syncthreads();

    // ---- axis -1 synthesis: tl = idwt(aa, ad), th = idwt(da, dd) along columns ----
    for (int p = tid; p < KH * IW_TW; p += DW_BLOCK) {
        const int r = p / IW_TW, nn = p % IW_TW;
        const int n = n0 + nn;
        double tl = 0.0, th = 0.0;
        if (n < a.out_w && kh0 + r < a.band_h) {
            // ascending band index kk with tap t = n + F - 2 - 2kk in [0, F)
            int kk_lo = (n - 1 + 1) / 2;  // ceil((n-1)/2) for n >= 0
            if (n == 0) kk_lo = 0;
            for (int kk = kk_lo; kk < a.band_w; kk++) {
                const int t = n + F - 2 - 2 * kk;
                if (t < 0) break;
                if (t >= F) continue;
                const int lc = kk - kw0;
                tl += s_a[0][r][lc] * a.lo[t] + s_a[1][r][lc] * a.hi[t];
                th += s_a[2][r][lc] * a.lo[t] + s_a[3][r][lc] * a.hi[t];
            }
        }
        s_tl[r][nn] = tl;
        s_th[r][nn] = th;
    }
    __syncthreads();

    // ---- axis -2 synthesis ----
    double *out = a.out + (size_t)plane * a.out_h * a.out_w;
    for (int p = tid; p < IW_TH * IW_TW; p += DW_BLOCK) {
        const int mm = p / IW_TW, nn = p % IW_TW;
        const int m = m0 + mm, n = n0 + nn;
        if (m >= a.out_h || n >= a.out_w) continue;
        double s = 0.0;
        int kk_lo = (m == 0) ? 0 : m / 2;  // ceil((m-1)/2)
        for (int kk = kk_lo; kk < a.band_h; kk++) {
            const int t = m + F - 2 - 2 * kk;
            if (t < 0) break;
            if (t >= F) continue;
            const int lr = kk - kh0;
            s += s_tl[lr][nn] * a.lo[t] + s_th[lr][nn] * a.hi[t];
        }
        out[(size_t)m * a.out_w + n] = s;
    }
}

// ---- host launchers -----------------------------------------------------------------------------

template <int F>
static int launch_dwt_F(const DwtKArgs &a, int planes, hipStream_t st) {
    dim3 grid((a.out_w + DW_TW - 1) / DW_TW, (a.out_h + DW_TH - 1) / DW_TH, planes);
    hipLaunchKernelGGL(k_dwt_level<F>, grid, dim3(DW_BLOCK), 0, st, a);
    return (int)hipGetLastError();
}
template <int F>
static int launch_idwt_F(const IdwtKArgs &a, int planes, hipStream_t st) {
    dim3 grid((a.out_w + IW_TW - 1) / IW_TW, (a.out_h + IW_TH - 1) / IW_TH, planes);
    hipLaunchKernelGGL(k_idwt_level<F>, grid, dim3(DW_BLOCK), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int spiht_launch_dwt_level(const DwtKArgs *a, int planes, hipStream_t st) {
    switch (a->F) {
    case 2: return launch_dwt_F<2>(*a, planes, st);
    case 6: return launch_dwt_F<6>(*a, planes, st);
    case 10: return launch_dwt_F<10>(*a, planes, st);
    case 18: return launch_dwt_F<18>(*a, planes, st);
    default: return -1;
    }
}
extern "C" int spiht_launch_idwt_level(const IdwtKArgs *a, int planes, hipStream_t st) {
    switch (a->F) {
    case 2: return launch_idwt_F<2>(*a, planes, st);
    case 6: return launch_idwt_F<6>(*a, planes, st);
    case 10: return launch_idwt_F<10>(*a, planes, st);
    case 18: return launch_idwt_F<18>(*a, planes, st);
    default: return -1;
    }
}
extern "C" int spiht_launch_quant_plain(const double *in, int32_t *out, size_t n_per_plane, int planes, int c,
                                        const double *mults, double q, hipStream_t st) {
    hipLaunchKernelGGL(k_quant_plain, dim3(1024), dim3(256), 0, st, in, out, n_per_plane, planes, c, mults, q);
    return (int)hipGetLastError();
}
extern "C" int spiht_launch_dequant_plain(const int32_t *in, double *out, size_t n_per_plane, int planes, int c,
                                          const double *mults, double q, hipStream_t st) {
    hipLaunchKernelGGL(k_dequant_plain, dim3(1024), dim3(256), 0, st, in, out, n_per_plane, planes, c, mults, q);
    return (int)hipGetLastError();
}
