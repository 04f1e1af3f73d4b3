/*
 * oracle/dwt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (float64, single thread) of the arithmetic the reference's
 * wrapper obtains from PyWavelets (third-party, pinned ==1.5.0 in
 * /root/reference/requirements.txt:8; 1.1.1 is what exists in this image):
 *
 *   encode side  spiht/spiht_wrapper.py:163-172
 *       pywt.wavedec2 -> pywt.coeffs_to_array -> channel_mults * arr -> quantize
 *   decode side  spiht/spiht_wrapper.py:259-276
 *       rec / channel_mults -> dequantize -> pywt.array_to_coeffs -> pywt.waverec2
 *   geometry     spiht/spiht_wrapper.py:92-139 (pywt.wavedecn_shapes)
 *
 * The published definitions restated here:
 *   analysis   cA[o] = sum_j dec_lo[j] * xe[2o+1-j],  o in [0,(N+F-1)/2)
 *   synthesis  x[n]  = sum_k cA[k]*rec_lo[n+F-2-2k] + cD[k]*rec_hi[n+F-2-2k],  n in [0,2L-F+2)
 *   dwt2 filters axis -2 first, then axis -1; idwt2 runs axis -1 first, then -2;
 *   waverec2 drops the last row/col of the running approximation when it is one
 *   longer than the next detail band; coeffs_to_array packs bands Mallat-style
 *   with zero padding.
 *
 * Pinning: tests/golden/ holds arrays captured from pywt 1.1.1 through the
 * reference's own wrapper (tests/golden/make_golden.py); agreement is to <= a
 * few ulp (pywt sums boundary taps in another order), quantised outputs equal.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXF 102

typedef struct {
    const char *name;
    int F;
    double dec_lo[MAXF], dec_hi[MAXF], rec_lo[MAXF], rec_hi[MAXF];
    float dec_lo_f[MAXF], dec_hi_f[MAXF];  /* the analysis filters PyWavelets uses on float32 input: the doubles rounded, except
                                            * for the coiflets, which it builds from a float table (wavelets.c) */
} wavelet_t;

/* pywt.Wavelet(name).filter_bank, repr(float), of every discrete wavelet, up to 102 taps (generated
 * from PyWavelets 1.1.1 by tools/gen_wavelets.py oracle; cross-checked against the library by tests/golden/make_golden.py) */
static const wavelet_t WAVELETS[] = {
#include "wavelets_table.h"
};
#define NWAVELETS ((int)(sizeof(WAVELETS) / sizeof(WAVELETS[0])))

/* modes: 0 reflect, 1 symmetric, 2 periodic, 3 zero, 4 constant */
enum { MODE_REFLECT = 0, MODE_SYMMETRIC = 1, MODE_PERIODIC = 2, MODE_ZERO = 3, MODE_CONSTANT = 4,
       /* ... and the extension modes of PyWavelets that are not index maps: the extended sample is computed */
       MODE_SMOOTH = 5, MODE_ANTISYMMETRIC = 6, MODE_ANTIREFLECT = 7,
       /* ... and periodization: another length rule (ceil(N / 2) coefficients, 2 L samples back) and its own alignment */
       MODE_PERIODIZATION = 8 };

int orc_wavelet_id(const char *name) {
    for (int i = 0; i < NWAVELETS; i++)
        if (!strcmp(WAVELETS[i].name, name)) return i;
    if (!strcmp(name, "db1")) return orc_wavelet_id("haar");
    return -1;
}
int orc_wavelet_len(int wid) { return (wid >= 0 && wid < NWAVELETS) ? WAVELETS[wid].F : -1; }
int orc_wavelet_filters(int wid, double *dec_lo, double *dec_hi, double *rec_lo, double *rec_hi) {
    if (wid < 0 || wid >= NWAVELETS) return -1;
    const wavelet_t *wv = &WAVELETS[wid];
    memcpy(dec_lo, wv->dec_lo, sizeof(double) * wv->F);
    memcpy(dec_hi, wv->dec_hi, sizeof(double) * wv->F);
    memcpy(rec_lo, wv->rec_lo, sizeof(double) * wv->F);
    memcpy(rec_hi, wv->rec_hi, sizeof(double) * wv->F);
    return wv->F;
}

/* pywt common.c dwt_max_level: floor(log2(len / (F-1))) with integer division */
int orc_dwt_max_level(int64_t len, int F) {
    if (F <= 1 || len < F - 1) return 0;
    int64_t q = len / (F - 1);
    int l = 0;
    while (q > 1) { q >>= 1; l++; }
    return l;
}

/* wavedec2 level resolution (pywt _multilevel._check_level): None (-1) -> min over the two axes */
int orc_resolve_level(int64_t H, int64_t W, int F, int level) {
    if (level >= 0) return level;
    int a = orc_dwt_max_level(H, F), b = orc_dwt_max_level(W, F);
    return a < b ? a : b;
}

/* Geometry (wrapper:92-139 / pywt.wavedecn_shapes): per-level band sizes hs[l], ws[l] for
 * l = 0..level (0 = image), LL size, and the padded coefficient-array size.  Returns level used. */
int orc_geometry(int64_t H, int64_t W, int F, int level, int64_t *hs, int64_t *ws, int64_t *ll_h,
                 int64_t *ll_w, int64_t *enc_h, int64_t *enc_w) {
    int L = orc_resolve_level(H, W, F, level);
    hs[0] = H; ws[0] = W;
    for (int l = 1; l <= L; l++) {
        hs[l] = (hs[l - 1] + F - 1) / 2;
        ws[l] = (ws[l - 1] + F - 1) / 2;
    }
    *ll_h = hs[L]; *ll_w = ws[L];
    int64_t ah = hs[L], aw = ws[L];
    for (int l = L; l >= 1; l--) { ah += hs[l]; aw += ws[l]; }
    *enc_h = ah; *enc_w = aw;
    return L;
}

/* ... for an extension mode: periodization keeps ceil(N / 2) coefficients per level -- the rule above with a two-tap
 * filter -- while the level count still follows the real filter length (pywt.dwt_max_level knows no mode) */
int orc_geometry_mode(int64_t H, int64_t W, int F, int mode, int level, int64_t *hs, int64_t *ws, int64_t *ll_h, int64_t *ll_w,
                      int64_t *enc_h, int64_t *enc_w) {
    const int L = orc_resolve_level(H, W, F, level);
    return orc_geometry(H, W, mode == MODE_PERIODIZATION ? 2 : F, L, hs, ws, ll_h, ll_w, enc_h, enc_w);
}

/* signal extension index map; returns -1 for "zero" */
static inline int64_t ext_index(int64_t i, int64_t N, int mode) {
    if (i >= 0 && i < N) return i;
    switch (mode) {
    case MODE_REFLECT: {
        if (N == 1) return 0;
        int64_t P = 2 * (N - 1);
        int64_t m = i % P; if (m < 0) m += P;
        return m < N ? m : P - m;
    }
    case MODE_SYMMETRIC: {
        int64_t P = 2 * N;
        int64_t m = i % P; if (m < 0) m += P;
        return m < N ? m : P - 1 - m;
    }
    case MODE_PERIODIC: {
        int64_t m = i % N; if (m < 0) m += N;
        return m;
    }
    case MODE_CONSTANT: return i < 0 ? 0 : N - 1;
    default: return -1;
    }
}

/* Sample i of the extended signal for the modes that compute it (pywt convolution.template.c, downsampling_convolution):
 *   smooth         straight line through the two samples at the edge: x[0] + k (x[0] - x[1]) at distance k to the left,
 *                  x[N-1] + k (x[N-1] - x[N-2]) to the right (N == 1: the edge value);
 *   antisymmetric  the half-sample mirror image negated: blocks of N samples, every other one with the opposite sign
 *                  (pywt subtracts the product: the same bits as adding the product with the negated sample);
 *   antireflect    the whole-sample mirror image through the edge VALUE: le - (x[k] - x[0]) at distance k <= N-1 to the
 *                  left with le = x[0]; the value reached at k = N-1 is the edge value of the next block, which runs
 *                  back through the signal, le + (x[N-1-k] - x[N-1]); likewise to the right from re = x[N-1]. */
static double ext_value(const double *x, int64_t N, int64_t sx, int64_t i, int mode) {
    if (i >= 0 && i < N) return x[i * sx];
    if (mode == MODE_SMOOTH) {
        if (N < 2) return x[0];
        if (i < 0) return x[0] + (double)(-i) * (x[0] - x[sx]);
        return x[(N - 1) * sx] + (double)(i - N + 1) * (x[(N - 1) * sx] - x[(N - 2) * sx]);
    }
    if (mode == MODE_ANTISYMMETRIC) {
        int64_t P = 2 * N, m = i % P;
        if (m < 0) m += P;
        int64_t b = (i - m) / N + (m >= N);   /* block number floor(i / N) */
        const double v = x[(m < N ? m : P - 1 - m) * sx];
        return (b & 1) ? -v : v;
    }
    /* antireflect */
    if (N < 2) return x[0];
    const int left = i < 0;
    int64_t d = left ? -i : i - N + 1;
    double e = left ? x[0] : x[(N - 1) * sx];
    int fwd = 1;  /* the first block walks away from the edge it started at */
    for (;;) {
        const int64_t k = d <= N - 1 ? d : N - 1;
        /* a block that starts at the left edge walks x[1], x[2], ...; one that starts at the right edge x[N-2], x[N-3], ... */
        const int from_left = left ? fwd : !fwd;
        const double t = from_left ? e - (x[k * sx] - x[0]) : e - (x[(N - 1 - k) * sx] - x[(N - 1) * sx]);
        const double t2 = from_left ? e + (x[k * sx] - x[0]) : e + (x[(N - 1 - k) * sx] - x[(N - 1) * sx]);
        /* first block of either side subtracts (mirror through the edge value); the block after it adds */
        const double v = fwd ? t : t2;
        if (d <= N - 1) return v;
        e = v;
        d -= N - 1;
        fwd = !fwd;
    }
}

/* 1-D analysis along a strided line.  Order of the additions as in pywt's downsampling_convolution
 * (convolution.template.c): taps in ascending order, except for the outputs that hang over the right end of the input
 * (2o+1 >= N): there the taps that read the signal extension come first, nearest first (filter index 2o+1-N down to 0),
 * then the others ascending -- through the signal and, when the input is shorter than the filter, on into the left-hand
 * extension (tests/golden/short_pywt.npz).  Constant-edge and smooth add their extension taps in ascending order too.
 * The truncating quantiser sees the difference on piecewise-constant 8-bit pictures: tests/golden/blocky_pywt.npz holds
 * pywt's arrays bit for bit. */
static void dwt_line(const double *x, int64_t N, int64_t sx, const double *lo, const double *hi, int F,
                     int mode, double *ca, double *cd, int64_t so) {
    int64_t L = (N + F - 1) / 2;
    for (int64_t o = 0; o < L; o++) {
        double a = 0.0, d = 0.0;
        /* (smooth, like constant, adds its extension taps in ascending order) */
        int64_t i = 2 * o + 1, jb = (i >= N && mode != MODE_CONSTANT && mode != MODE_SMOOTH) ? i - N : -1;
        for (int s = 0; s < F; s++) {
            int j = s <= jb ? (int)(jb - s) : s;
            double v;
            if (mode >= MODE_SMOOTH) {
                v = ext_value(x, N, sx, i - j, mode);
            } else {
                int64_t idx = ext_index(i - j, N, mode);
                v = idx < 0 ? 0.0 : x[idx * sx];
            }
            a += lo[j] * v;
            d += hi[j] * v;
        }
        ca[o * so] = a;
        cd[o * so] = d;
    }
}

/* Periodization (pywt downsampling_convolution_periodization): the signal, made even by repeating its last sample, is
 * extended periodically; output o = sum_j f[j] xe[F/2 + 2o - j], o < ceil(N / 2).  Order of the additions as everywhere in
 * that file: ascending taps, except that where the window hangs over the right end the taps that read beyond it come
 * first, nearest first. */
static void dwt_line_per(const double *x, int64_t N, int64_t sx, const double *lo, const double *hi, int F, double *ca,
                         double *cd, int64_t so) {
    const int64_t L = (N + 1) / 2, Np = N + (N & 1);
    for (int64_t o = 0; o < L; o++) {
        double a = 0.0, d = 0.0;
        const int64_t i = F / 2 + 2 * o, jb = i >= N ? i - N : -1;
        for (int s = 0; s < F; s++) {
            const int j = s <= jb ? (int)(jb - s) : s;
            int64_t m = (i - j) % Np;
            if (m < 0) m += Np;
            const double v = x[(m < N ? m : N - 1) * sx];
            a += lo[j] * v;
            d += hi[j] * v;
        }
        ca[o * so] = a;
        cd[o * so] = d;
    }
}
/* ... and back (upsampling_convolution_valid_sf_periodization): 2 L samples, x[n] = sum_k rec[n - 2k + F/2 - 1] c[k mod L],
 * i.e. with p = (n + F/2 - 1) & 1 and i = (n + F/2 - 1 - p) / 2 the terms j = 0 .. F/2-1: tap 2j + p against c[(i - j) mod L].
 * pywt adds every product straight into the output sample, the approximation's first, then the detail's. */
static void idwt_line_per(const double *ca, const double *cd, int64_t L, int64_t si, const double *lo, const double *hi, int F,
                          double *x, int64_t so) {
    const int64_t s0 = F / 2 - 1;
    for (int64_t n = 0; n < 2 * L; n++) {
        const int p = (int)((n + s0) & 1);
        const int64_t i = (n + s0 - p) / 2;
        /* where the window hangs over the right end (i >= L) the terms beyond it come first, nearest first -- and sample 0
         * of a filter with an even number of tap PAIRS is computed by pywt together with sample 2L-1, as the odd half of
         * that window (i = L + F/4 - 1) */
        int64_t jb = i >= L ? i - L : -1;
        if (n == 0 && (F / 2) % 2 == 0) jb = F / 4 - 1;
        double acc = 0.0;
        for (int pass = 0; pass < 2; pass++) {
            const double *c = pass ? cd : ca, *f = pass ? hi : lo;
            for (int s = 0; s < F / 2; s++) {
                const int j = s <= jb ? (int)(jb - s) : s;
                int64_t k = (i - j) % L;
                if (k < 0) k += L;
                acc += f[2 * j + p] * c[k * si];
            }
        }
        x[n * so] = acc;
    }
}

/* 1-D synthesis along a strided line; output length 2L-F+2.  Order of the additions as in pywt's
 * upsampling_convolution_valid_sf (convolution.template.c): the approximation part and the detail part are two separate
 * sums, each over j = 0..F/2-1 with taps 2j (even outputs) / 2j+1 (odd outputs) against input i-j, and the detail sum is
 * added to the finished approximation sum (idwt: `output += ...` twice). */
static void idwt_line(const double *ca, const double *cd, int64_t L, int64_t si, const double *lo,
                      const double *hi, int F, double *x, int64_t so) {
    int64_t N = 2 * L - F + 2;
    for (int64_t n = 0; n < N; n++) {
        /* output n = 2(i - (F/2-1)) + p  with p = n & 1:  i = n/2 + F/2 - 1;  term j uses tap 2j+p and input i-j */
        int64_t i = n / 2 + F / 2 - 1;
        int p = (int)(n & 1);
        double sa = 0.0, sd = 0.0;
        for (int j = 0; j < F / 2; j++) {
            int64_t k = i - j;
            if (k < 0 || k >= L) continue;
            sa += lo[2 * j + p] * ca[k * si];
            sd += hi[2 * j + p] * cd[k * si];
        }
        x[n * so] = (0.0 + sa) + sd;
    }
}

/* one 2-D analysis level on one channel: in [h,w] -> aa, ad, da, dd each [h2,w2] (axis -2 first) */
static int dwt2_level(const double *in, int64_t h, int64_t w, const wavelet_t *wv, int mode, double *aa,
                      double *ad, double *da, double *dd) {
    int F = wv->F;
    const int per = mode == MODE_PERIODIZATION;
    int64_t h2 = per ? (h + 1) / 2 : (h + F - 1) / 2, w2 = per ? (w + 1) / 2 : (w + F - 1) / 2;
    double *ta = (double *)malloc(sizeof(double) * h2 * w), *td = (double *)malloc(sizeof(double) * h2 * w);
    if (!ta || !td) { free(ta); free(td); return -1; }
    if (per) {
        for (int64_t j = 0; j < w; j++) dwt_line_per(in + j, h, w, wv->dec_lo, wv->dec_hi, F, ta + j, td + j, w);
        for (int64_t i = 0; i < h2; i++) {
            dwt_line_per(ta + i * w, w, 1, wv->dec_lo, wv->dec_hi, F, aa + i * w2, ad + i * w2, 1);
            dwt_line_per(td + i * w, w, 1, wv->dec_lo, wv->dec_hi, F, da + i * w2, dd + i * w2, 1);
        }
        free(ta); free(td);
        return 0;
    }
    for (int64_t j = 0; j < w; j++) dwt_line(in + j, h, w, wv->dec_lo, wv->dec_hi, F, mode, ta + j, td + j, w);
    for (int64_t i = 0; i < h2; i++) {
        dwt_line(ta + i * w, w, 1, wv->dec_lo, wv->dec_hi, F, mode, aa + i * w2, ad + i * w2, 1);
        dwt_line(td + i * w, w, 1, wv->dec_lo, wv->dec_hi, F, mode, da + i * w2, dd + i * w2, 1);
    }
    free(ta); free(td);
    return 0;
}

/* wavedec2 + coeffs_to_array (float64 array out, zero padded) : wrapper:163-165 */
int orc_wavedec2_array(const double *img, int64_t c, int64_t H, int64_t W, int wid, int mode, int level,
                       double *arr /* [c,enc_h,enc_w] */) {
    if (wid < 0 || wid >= NWAVELETS) return -1;
    const wavelet_t *wv = &WAVELETS[wid];
    int64_t hs[64], ws[64], ll_h, ll_w, eh, ew;
    int L = orc_geometry_mode(H, W, wv->F, mode, level, hs, ws, &ll_h, &ll_w, &eh, &ew);
    memset(arr, 0, sizeof(double) * c * eh * ew);
    /* band offsets: level l detail block sits at accumulated offset A_l */
    int64_t offh[64], offw[64];
    int64_t ah = ll_h, aw = ll_w;
    for (int l = L; l >= 1; l--) { offh[l] = ah; offw[l] = aw; ah += hs[l]; aw += ws[l]; }
    for (int64_t k = 0; k < c; k++) {
        double *cur = (double *)malloc(sizeof(double) * H * W);
        if (!cur) return -2;
        memcpy(cur, img + k * H * W, sizeof(double) * H * W);
        double *out = arr + k * eh * ew;
        for (int l = 1; l <= L; l++) {
            int64_t h2 = hs[l], w2 = ws[l];
            double *aa = (double *)malloc(sizeof(double) * h2 * w2 * 4);
            if (!aa) { free(cur); return -2; }
            double *ad = aa + h2 * w2, *da = ad + h2 * w2, *dd = da + h2 * w2;
            if (dwt2_level(cur, hs[l - 1], ws[l - 1], wv, mode, aa, ad, da, dd)) { free(aa); free(cur); return -2; }
            for (int64_t i = 0; i < h2; i++)
                for (int64_t j = 0; j < w2; j++) {
                    out[i * ew + offw[l] + j] = ad[i * w2 + j];              /* 'ad': top-right    */
                    out[(offh[l] + i) * ew + j] = da[i * w2 + j];            /* 'da': bottom-left  */
                    out[(offh[l] + i) * ew + offw[l] + j] = dd[i * w2 + j];  /* 'dd': bottom-right */
                }
            free(cur);
            cur = (double *)malloc(sizeof(double) * h2 * w2);
            if (!cur) { free(aa); return -2; }
            memcpy(cur, aa, sizeof(double) * h2 * w2);
            free(aa);
        }
        for (int64_t i = 0; i < ll_h; i++)
            for (int64_t j = 0; j < ll_w; j++) out[i * ew + j] = cur[i * ws[L] + j];
        free(cur);
    }
    return 0;
}

/* wrapper:167-172, :9-11 : int32((m_k * x) * q), truncation toward zero (numpy astype) */
void orc_quantize(const double *arr, int64_t c, int64_t n_per_c, const double *mults /* or NULL */,
                  double q, int32_t *out) {
    for (int64_t k = 0; k < c; k++)
        for (int64_t t = 0; t < n_per_c; t++) {
            double v = arr[k * n_per_c + t];
            if (mults) v = mults[k] * v;
            v = v * q;
            out[k * n_per_c + t] = (int32_t)v;
        }
}

/* wrapper:270-274, :13-14 : (rec / m_k) / q */
void orc_dequantize(const int32_t *rec, int64_t c, int64_t n_per_c, const double *mults, double q,
                    double *out) {
    for (int64_t k = 0; k < c; k++)
        for (int64_t t = 0; t < n_per_c; t++) {
            double v = (double)rec[k * n_per_c + t];
            if (mults) v = v / mults[k];
            out[k * n_per_c + t] = v / q;
        }
}

/* output image size of waverec2 for (H,W,F,level): sizes follow the trim rule */
void orc_waverec2_shape_mode(int64_t H, int64_t W, int F, int mode, int level, int64_t *Ho, int64_t *Wo) {
    int64_t hs[64], ws[64], ll_h, ll_w, eh, ew;
    int L = orc_geometry_mode(H, W, F, mode, level, hs, ws, &ll_h, &ll_w, &eh, &ew);
    const int Fg = mode == MODE_PERIODIZATION ? 2 : F;
    int64_t ah = ll_h, aw = ll_w;
    for (int l = L; l >= 1; l--) {
        if (ah == hs[l] + 1) ah--;
        if (aw == ws[l] + 1) aw--;
        ah = 2 * hs[l] - Fg + 2;
        aw = 2 * ws[l] - Fg + 2;
    }
    *Ho = ah; *Wo = aw;
}
void orc_waverec2_shape(int64_t H, int64_t W, int F, int level, int64_t *Ho, int64_t *Wo) {
    orc_waverec2_shape_mode(H, W, F, MODE_REFLECT, level, Ho, Wo);
}

/* array_to_coeffs + waverec2 : wrapper:275-276.  out is [c,Ho,Wo] from orc_waverec2_shape_mode.  (The extension mode
 * matters to the way back only when it is periodization.) */
int orc_waverec2_array_mode(const double *arr, int64_t c, int64_t H, int64_t W, int wid, int mode, int level, double *out);
int orc_waverec2_array(const double *arr, int64_t c, int64_t H, int64_t W, int wid, int level, double *out) {
    return orc_waverec2_array_mode(arr, c, H, W, wid, MODE_REFLECT, level, out);
}
int orc_waverec2_array_mode(const double *arr, int64_t c, int64_t H, int64_t W, int wid, int mode, int level, double *out) {
    if (wid < 0 || wid >= NWAVELETS) return -1;
    const wavelet_t *wv = &WAVELETS[wid];
    const int per = mode == MODE_PERIODIZATION;
    int F = wv->F;
    int64_t hs[64], ws[64], ll_h, ll_w, eh, ew;
    int L = orc_geometry_mode(H, W, F, mode, level, hs, ws, &ll_h, &ll_w, &eh, &ew);
    int64_t offh[64], offw[64];
    int64_t ah0 = ll_h, aw0 = ll_w;
    for (int l = L; l >= 1; l--) { offh[l] = ah0; offw[l] = aw0; ah0 += hs[l]; aw0 += ws[l]; }
    int64_t Ho, Wo;
    orc_waverec2_shape_mode(H, W, F, mode, level, &Ho, &Wo);
    for (int64_t k = 0; k < c; k++) {
        const double *in = arr + k * eh * ew;
        int64_t ah = ll_h, aw = ll_w;
        double *a = (double *)malloc(sizeof(double) * ah * aw);
        if (!a) return -2;
        for (int64_t i = 0; i < ah; i++)
            for (int64_t j = 0; j < aw; j++) a[i * aw + j] = in[i * ew + j];
        for (int l = L; l >= 1; l--) {
            int64_t h2 = hs[l], w2 = ws[l];
            /* trim rule (_multilevel.py waverec2): a_len == d_len + 1 -> drop last */
            int64_t uh = (ah == h2 + 1) ? ah - 1 : ah, uw = (aw == w2 + 1) ? aw - 1 : aw;
            if (uh != h2 || uw != w2) { free(a); return -3; }
            int64_t wo = per ? 2 * w2 : 2 * w2 - F + 2, ho = per ? 2 * h2 : 2 * h2 - F + 2;
            /* axis -1 first: (aa,ad) -> lo rows ; (da,dd) -> hi rows */
            double *tl = (double *)malloc(sizeof(double) * h2 * wo), *th = (double *)malloc(sizeof(double) * h2 * wo);
            double *rowd = (double *)malloc(sizeof(double) * w2 * 3);
            double *nx = (double *)malloc(sizeof(double) * ho * wo);
            if (!tl || !th || !rowd || !nx) { free(tl); free(th); free(rowd); free(nx); free(a); return -2; }
            for (int64_t i = 0; i < h2; i++) {
                const double *ad = in + i * ew + offw[l];
                const double *da = in + (offh[l] + i) * ew;
                const double *dd = in + (offh[l] + i) * ew + offw[l];
                if (per) {
                    idwt_line_per(a + i * aw, ad, w2, 1, wv->rec_lo, wv->rec_hi, F, tl + i * wo, 1);
                    idwt_line_per(da, dd, w2, 1, wv->rec_lo, wv->rec_hi, F, th + i * wo, 1);
                } else {
                    idwt_line(a + i * aw, ad, w2, 1, wv->rec_lo, wv->rec_hi, F, tl + i * wo, 1);
                    idwt_line(da, dd, w2, 1, wv->rec_lo, wv->rec_hi, F, th + i * wo, 1);
                }
            }
            for (int64_t j = 0; j < wo; j++) {
                if (per) idwt_line_per(tl + j, th + j, h2, wo, wv->rec_lo, wv->rec_hi, F, nx + j, wo);
                else idwt_line(tl + j, th + j, h2, wo, wv->rec_lo, wv->rec_hi, F, nx + j, wo);
            }
            free(tl); free(th); free(rowd); free(a);
            a = nx; ah = ho; aw = wo;
        }
        memcpy(out + k * Ho * Wo, a, sizeof(double) * Ho * Wo);
        free(a);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Single-precision forward path.  PyWavelets transforms float32 (and float16) input in float32
 * (_check_dtype), with float copies of the filters, and the wrapper then quantises the float32 array
 * (spiht_wrapper.py:163-172 with :9-11; `arr * q_scale` stays float32, numpy scalar rules) -- unless
 * per-channel scales are given: `channel_mults[:,None,None] * coeffs_arr` is float64 from there on.
 *
 * In float32 the ORDER of the additions matters for the quantised result, so this restates the order of
 * pywt's convolution.template.c (downsampling_convolution, input at least as long as the filter): for
 * outputs that do not hang over the right end, taps in ascending order; for the right overhang (i >= N),
 * first the taps that read the extension, nearest first (filter index i-N down to 0), then the others
 * ascending.  Checked against pywt 1.1.1: bit-identical float32 lines (tests/golden: wrapper32).
 * ------------------------------------------------------------------------------------------------ */
/* ext_value in single precision (the template instantiated with float: every intermediate is a float) */
static float ext_value_f(const float *x, int64_t N, int64_t sx, int64_t i, int mode) {
    if (mode == MODE_PERIODIZATION) {
        int64_t Np = N + (N & 1), m = i % Np;
        if (m < 0) m += Np;
        return x[(m < N ? m : N - 1) * sx];
    }
    if (i >= 0 && i < N) return x[i * sx];
    if (mode == MODE_SMOOTH) {
        if (N < 2) return x[0];
        volatile float df, pr, v;
        if (i < 0) { df = x[0] - x[sx]; pr = (float)(-i) * df; v = x[0] + pr; return v; }
        df = x[(N - 1) * sx] - x[(N - 2) * sx]; pr = (float)(i - N + 1) * df; v = x[(N - 1) * sx] + pr;
        return v;
    }
    if (mode == MODE_ANTISYMMETRIC) {
        int64_t P = 2 * N, m = i % P;
        if (m < 0) m += P;
        int64_t b = (i - m) / N + (m >= N);
        const float v = x[(m < N ? m : P - 1 - m) * sx];
        return (b & 1) ? -v : v;
    }
    if (N < 2) return x[0];
    const int left = i < 0;
    int64_t d = left ? -i : i - N + 1;
    volatile float e = left ? x[0] : x[(N - 1) * sx];
    int away = 1;
    for (;;) {
        const int64_t k = d <= N - 1 ? d : N - 1;
        const int from_left = left ? away : !away;
        volatile float dl = from_left ? x[k * sx] - x[0] : x[(N - 1 - k) * sx] - x[(N - 1) * sx];
        volatile float v = away ? e - dl : e + dl;
        if (d <= N - 1) return v;
        e = v;
        d -= N - 1;
        away = !away;
    }
}

static void dwt_line_f(const float *x, int64_t N, int64_t sx, const float *lo, const float *hi, int F, int mode,
                       float *ca, float *cd, int64_t so) {
    const int per = mode == MODE_PERIODIZATION;
    int64_t L = per ? (N + 1) / 2 : (N + F - 1) / 2;
    for (int64_t o = 0; o < L; o++) {
        volatile float a = 0.0f, d = 0.0f; /* volatile: no contraction, no reassociation, no excess precision */
        /* constant-edge mode: pywt adds the replicated-edge taps in ascending order too, i.e. plain ascending everywhere
         * (and so does smooth) */
        int64_t i = per ? F / 2 + 2 * o : 2 * o + 1, jb = (i >= N && mode != MODE_CONSTANT && mode != MODE_SMOOTH) ? i - N : -1;
        for (int s = 0; s < F; s++) {
            int j = s <= jb ? (int)(jb - s) : s;
            float v;
            if (mode >= MODE_SMOOTH) {
                v = ext_value_f(x, N, sx, i - j, mode);
            } else {
                int64_t idx = ext_index(i - j, N, mode);
                v = idx < 0 ? 0.0f : x[idx * sx];
            }
            volatile float pa = lo[j] * v, pd = hi[j] * v;
            a = a + pa;
            d = d + pd;
        }
        ca[o * so] = a;
        cd[o * so] = d;
    }
}

/* float32 wavedec2 + coeffs_to_array; arr [c,enc_h,enc_w] float, zero padded. */
int orc_wavedec2_array_f32(const float *img, int64_t c, int64_t H, int64_t W, int wid, int mode, int level, float *arr) {
    if (wid < 0 || wid >= NWAVELETS) return -1;
    const wavelet_t *wv = &WAVELETS[wid];
    int F = wv->F;
    int64_t hs[64], ws[64], ll_h, ll_w, eh, ew;
    int L = orc_geometry_mode(H, W, F, mode, level, hs, ws, &ll_h, &ll_w, &eh, &ew);
    /* (inputs shorter than the filter take the same order of additions: checked against PyWavelets, tests/golden) */
    float lo[MAXF], hi[MAXF];
    for (int j = 0; j < F; j++) { lo[j] = wv->dec_lo_f[j]; hi[j] = wv->dec_hi_f[j]; }
    memset(arr, 0, sizeof(float) * c * eh * ew);
    int64_t offh[64], offw[64];
    int64_t ah = ll_h, aw = ll_w;
    for (int l = L; l >= 1; l--) { offh[l] = ah; offw[l] = aw; ah += hs[l]; aw += ws[l]; }
    for (int64_t k = 0; k < c; k++) {
        float *cur = (float *)malloc(sizeof(float) * H * W);
        if (!cur) return -2;
        memcpy(cur, img + k * H * W, sizeof(float) * H * W);
        float *out = arr + k * eh * ew;
        for (int l = 1; l <= L; l++) {
            int64_t h = hs[l - 1], w = ws[l - 1], h2 = hs[l], w2 = ws[l];
            float *ta = (float *)malloc(sizeof(float) * h2 * w * 2), *aa = (float *)malloc(sizeof(float) * h2 * w2 * 4);
            if (!ta || !aa) { free(ta); free(aa); free(cur); return -2; }
            float *td = ta + h2 * w, *ad = aa + h2 * w2, *da = ad + h2 * w2, *dd = da + h2 * w2;
            for (int64_t j = 0; j < w; j++) dwt_line_f(cur + j, h, w, lo, hi, F, mode, ta + j, td + j, w);  /* axis -2 */
            for (int64_t i = 0; i < h2; i++) {                                                                 /* axis -1 */
                dwt_line_f(ta + i * w, w, 1, lo, hi, F, mode, aa + i * w2, ad + i * w2, 1);
                dwt_line_f(td + i * w, w, 1, lo, hi, F, mode, da + i * w2, dd + i * w2, 1);
            }
            for (int64_t i = 0; i < h2; i++)
                for (int64_t j = 0; j < w2; j++) {
                    out[i * ew + offw[l] + j] = ad[i * w2 + j];
                    out[(offh[l] + i) * ew + j] = da[i * w2 + j];
                    out[(offh[l] + i) * ew + offw[l] + j] = dd[i * w2 + j];
                }
            free(cur);
            cur = (float *)malloc(sizeof(float) * h2 * w2);
            if (!cur) { free(ta); free(aa); return -2; }
            memcpy(cur, aa, sizeof(float) * h2 * w2);
            free(ta); free(aa);
        }
        for (int64_t i = 0; i < ll_h; i++)
            for (int64_t j = 0; j < ll_w; j++) out[i * ew + j] = cur[i * ws[L] + j];
        free(cur);
    }
    return 0;
}

/* wrapper:167-172 on a float32 array: without channel scales the product with q_scale is float32 (numpy keeps the
 * array's dtype against a Python scalar); with them the array is promoted to float64 first */
void orc_quantize_f32(const float *arr, int64_t c, int64_t n_per_c, const double *mults /* or NULL */, double q, int32_t *out) {
    const float qf = (float)q;
    for (int64_t k = 0; k < c; k++)
        for (int64_t t = 0; t < n_per_c; t++) {
            if (mults) {
                double v = mults[k] * (double)arr[k * n_per_c + t];
                v = v * q;
                out[k * n_per_c + t] = (int32_t)v;
            } else {
                volatile float v = arr[k * n_per_c + t] * qf;
                out[k * n_per_c + t] = (int32_t)v;
            }
        }
}
