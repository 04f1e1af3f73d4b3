// HBM bandwidth ceilings for the DWT's traffic mix (read float64, write float64 + 3x int32), by access width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each thread: reads NR doubles per 4 outputs; variant W = elements per lane per access
template <int W>
__global__ void k_mix(const double *__restrict__ in, double *__restrict__ ll, int *__restrict__ b0, int *__restrict__ b1,
                      int *__restrict__ b2, size_t nq) {
    // nq = number of output positions; input has 4*nq doubles
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = t * W; q + W <= nq; q += stride * W) {
        double acc[W];
#pragma unroll
        for (int w = 0; w < W; w++) acc[w] = 0;
        // 4 input doubles per output position, read as W-wide vectors from 4 streams
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double *p = in + (size_t)s * nq + q;
            if (W == 1) acc[0] += p[0];
            if (W == 2) { double2 v = *reinterpret_cast<const double2 *>(p); acc[0] += v.x; acc[1] += v.y; }
            if (W == 4) { double4 v = *reinterpret_cast<const double4 *>(p); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        }
        if (W == 1) { ll[q] = acc[0]; b0[q] = (int)acc[0]; b1[q] = (int)(acc[0] * 2); b2[q] = (int)(acc[0] * 3); }
        if (W == 2) {
            *reinterpret_cast<double2 *>(ll + q) = make_double2(acc[0], acc[1]);
            *reinterpret_cast<int2 *>(b0 + q) = make_int2((int)acc[0], (int)acc[1]);
            *reinterpret_cast<int2 *>(b1 + q) = make_int2((int)(acc[0] * 2), (int)(acc[1] * 2));
            *reinterpret_cast<int2 *>(b2 + q) = make_int2((int)(acc[0] * 3), (int)(acc[1] * 3));
        }
        if (W == 4) {
            *reinterpret_cast<double4 *>(ll + q) = make_double4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<int4 *>(b0 + q) = make_int4((int)acc[0], (int)acc[1], (int)acc[2], (int)acc[3]);
            *reinterpret_cast<int4 *>(b1 + q) = make_int4((int)(acc[0] * 2), (int)(acc[1] * 2), (int)(acc[2] * 2), (int)(acc[3] * 2));
            *reinterpret_cast<int4 *>(b2 + q) = make_int4((int)(acc[0] * 3), (int)(acc[1] * 3), (int)(acc[2] * 3), (int)(acc[3] * 3));
        }
    }
}

template <int W>
static void run(const char *name, double *in, double *ll, int *b0, int *b1, int *b2, size_t nq, int blocks) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_mix<W>, dim3(blocks), dim3(256), 0, 0, in, ll, b0, b1, b2, nq);
    CHK(hipEventRecord(a));
    const int it = 5;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL(k_mix<W>, dim3(blocks), dim3(256), 0, 0, in, ll, b0, b1, b2, nq);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)nq * (32 + 8 + 12) * it;
    printf("%-28s blocks %6d : %.1f GB/s\n", name, blocks, bytes / (ms * 1e-3) / 1e9);
}

int main() {
    size_t nq = (size_t)400 << 20;  // 400 Mi output positions: 12.8 GB read, 8 GB written
    double *in, *ll; int *b0, *b1, *b2;
    CHK(hipMalloc(&in, nq * 4 * 8)); CHK(hipMalloc(&ll, nq * 8));
    CHK(hipMalloc(&b0, nq * 4)); CHK(hipMalloc(&b1, nq * 4)); CHK(hipMalloc(&b2, nq * 4));
    CHK(hipMemset(in, 0, nq * 4 * 8));
    for (int blocks : {2048, 8192, 65536}) {
        run<1>("8B loads / 4B+8B stores", in, ll, b0, b1, b2, nq, blocks);
        run<2>("16B loads / 8B+16B stores", in, ll, b0, b1, b2, nq, blocks);
        run<4>("32B loads / 16B+32B stores", in, ll, b0, b1, b2, nq, blocks);
    }
    return 0;
}
