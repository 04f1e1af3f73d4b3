/*
 * oracle/color_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU twin of the GPU's colour model change (spiht_amd/csrc/dwt.hip: color3_px / k_color3; reference call sites
 * spiht/spiht_wrapper.py:158-160, :278-279 -> spiht/color_models.py:6-13 -> colour-science 0.4.4, which is absent here:
 * COLOUR PARITY IS UNPINNED).  It includes the very header the kernels include (csrc/spow.h, spow_tables.h) and
 * performs the same float64 operations in the same order (this file is built with -ffp-contract=off, fused
 * multiply-adds are explicit), so the device result can be checked bit for bit, and the power function's accuracy
 * can be measured on the CPU.
 */
#include <math.h>
#include <stdint.h>

#define SPOW_TABLE_QUAL static const
#include "../spiht_amd/csrc/spow_tables.h"
#include "../spiht_amd/csrc/spow.h"

double orc_spow(double x, double p) { return spow_signed(x, p, SPOW_INV, SPOW_LOG2C, SPOW_EXP2); }

/* in / out: [3][npix] planes; A, M: row-major 3x3; w = M * spow(A * u, p) per pixel, numpy's dot order */
void orc_color3(const double *in, double *out, int64_t npix, const double *A, const double *M, double p) {
    for (int64_t t = 0; t < npix; t++) {
        const double u0 = in[t], u1 = in[npix + t], u2 = in[2 * npix + t];
        double v[3];
        for (int r = 0; r < 3; r++) {
            const double x = (u0 * A[3 * r] + u1 * A[3 * r + 1]) + u2 * A[3 * r + 2];
            v[r] = orc_spow(x, p);
        }
        out[t] = (v[0] * M[0] + v[1] * M[1]) + v[2] * M[2];
        out[npix + t] = (v[0] * M[3] + v[1] * M[4]) + v[2] * M[5];
        out[2 * npix + t] = (v[0] * M[6] + v[1] * M[7]) + v[2] * M[8];
    }
}
