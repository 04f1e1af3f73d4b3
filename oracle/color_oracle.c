/*
 * oracle/color_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The colour model change of the reference's wrapper (spiht/spiht_wrapper.py:158-160, :278-279 ->
 * spiht/color_models.py:6-13 -> colour-science 0.4.4 `colour.convert`, which is absent here: COLOUR PARITY IS UNPINNED
 * beyond the published known answer of the XYZ -> IPT half, tests/test_oracle.py) restated on the CPU with the C
 * library's pow(): per pixel w = M * spow(A * u, p), spow(x, p) = sign(x) |x|^p (colour-science's `spow`), numpy's
 * dot order.  Independent of the product: nothing of spiht_amd/ is included -- the GPU kernels use their own power
 * function (csrc/spow.h) and are held to this one within a stated number of units in the last place.
 */
#include <math.h>
#include <stdint.h>

static double spow_libm(double x, double p) {
    if (x == 0.0) return 0.0;
    const double m = pow(fabs(x), p);
    return x < 0.0 ? -m : m;
}

/* in / out: [3][npix] planes; A, M: row-major 3x3 */
void orc_color3(const double *in, double *out, int64_t npix, const double *A, const double *M, double p) {
    for (int64_t t = 0; t < npix; t++) {
        const double u0 = in[t], u1 = in[npix + t], u2 = in[2 * npix + t];
        double v[3];
        for (int r = 0; r < 3; r++) {
            const double x = (u0 * A[3 * r] + u1 * A[3 * r + 1]) + u2 * A[3 * r + 2];
            v[r] = spow_libm(x, p);
        }
        out[t] = (v[0] * M[0] + v[1] * M[1]) + v[2] * M[2];
        out[npix + t] = (v[0] * M[3] + v[1] * M[4]) + v[2] * M[5];
        out[2 * npix + t] = (v[0] * M[6] + v[1] * M[7]) + v[2] * M[8];
    }
}
