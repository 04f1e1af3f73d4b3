/* tests/native/spow_probe.c -- unit-test hook, built by tests/test_oracle.py with gcc: the power function of the GPU's
 * colour kernels (spiht_amd/csrc/spow.h, a host/device header) compiled for the CPU, so that its accuracy can be
 * measured against 60-digit arithmetic.  A test of the product header -- not part of the oracle. */
#include <math.h>
#include <stdint.h>
#define SPOW_FN static inline
#define SPOW_FMA(a, b, c) fma((a), (b), (c))
#define SPOW_RINT(a) rint(a)
#define SPOW_LDEXP(a, n) ldexp((a), (n))
#define SPOW_TABLE_QUAL static const
#include "../../spiht_amd/csrc/spow_tables.h"
#include "../../spiht_amd/csrc/spow.h"
double probe_spow(double x, double p) { return spow_signed(x, p, SPOW_INV, SPOW_LOG2C, SPOW_EXP2); }
