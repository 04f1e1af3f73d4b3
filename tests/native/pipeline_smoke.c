/* tests/native/pipeline_smoke.c -- a caller of the C ABI that is not Python (built by tests/test_gpu_image.py with gcc
 * against include/spiht_hip.h and libspiht_hip.so): the pipelined round trip of INTEGRATION.md on synthetic pictures.
 * Prints, per step, a checksum of the streams and of the decoded pictures; the test compares them with what the fused
 * Python calls give for the same pixels.   usage: pipeline_smoke B c H W level max_bits steps */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spiht_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        int _s = (x);                                                                      \
        if (_s != SPIHT_OK) {                                                              \
            fprintf(stderr, "%s -> %d (%s; %s)\n", #x, _s, spiht_strerror(_s), spiht_last_hip_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static uint64_t fnv(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 8) return 2;
    const int64_t B = atoll(argv[1]), c = atoll(argv[2]), H = atoll(argv[3]), W = atoll(argv[4]);
    const int level = atoi(argv[5]);
    const uint64_t max_bits = strtoull(argv[6], 0, 10);
    const int steps = atoi(argv[7]);
    spiht_ctx *ctx;
    CK(spiht_ctx_create(0, &ctx));
    spiht_pipeline *pl;
    CK(spiht_pipeline_create_on(ctx, 0, B, c, H, W, spiht_wavelet_id("bior2.2"), spiht_mode_id("reflect"), level, 50.0, NULL, max_bits, &pl));
    uint64_t slot;
    int64_t rh, rw;
    CK(spiht_pipeline_info(pl, &slot, &rh, &rw));
    const size_t npx = (size_t)B * c * H * W, nrec = (size_t)B * c * rh * rw;
    double *img = (double *)malloc(npx * 8), *rec = (double *)malloc(nrec * 8);
    uint8_t *streams = (uint8_t *)malloc((size_t)B * slot), *maxn = (uint8_t *)malloc(B);
    uint64_t *nbits = (uint64_t *)malloc(B * 8);
    void **d_img = (void **)malloc(steps * sizeof(void *)), **d_rec = (void **)malloc(steps * sizeof(void *));
    void **d_out = (void **)malloc(steps * sizeof(void *)), **d_nb = (void **)malloc(steps * sizeof(void *));
    void **d_mn = (void **)malloc(steps * sizeof(void *));
    for (int s = 0; s < steps; s++) {
        /* pictures: a deterministic smooth pattern + a little hash noise, 8-bit levels / 255 (the test makes the same) */
        for (size_t t = 0; t < npx; t++) {
            const size_t x = t % W, y = (t / W) % H, k = t / ((size_t)W * H);
            const uint32_t hsh = (uint32_t)((t + 1) * 2654435761u + (uint32_t)s * 40503u);
            const int v = (int)((x * 3 + y * 5 + k * 17 + s * 29) % 200) + (int)(hsh >> 28);
            img[t] = (double)v / 255.0;
        }
        CK(spiht_dev_alloc(ctx, npx * 8, &d_img[s]));
        CK(spiht_dev_alloc(ctx, nrec * 8, &d_rec[s]));
        CK(spiht_dev_alloc(ctx, (uint64_t)B * slot, &d_out[s]));
        CK(spiht_dev_alloc(ctx, (uint64_t)B * 8, &d_nb[s]));
        CK(spiht_dev_alloc(ctx, (uint64_t)B, &d_mn[s]));
        CK(spiht_dev_upload(ctx, d_img[s], img, npx * 8));
    }
    CK(spiht_ctx_synchronize(ctx));
    for (int s = 0; s < steps; s++)  /* queues kernels only */
        CK(spiht_pipeline_submit(pl, (const double *)d_img[s], (uint8_t *)d_out[s], (uint64_t *)d_nb[s], (uint8_t *)d_mn[s], (double *)d_rec[s]));
    CK(spiht_pipeline_synchronize(pl));
    for (int s = 0; s < steps; s++) {
        CK(spiht_dev_download(ctx, streams, d_out[s], (uint64_t)B * slot));
        CK(spiht_dev_download(ctx, nbits, d_nb[s], (uint64_t)B * 8));
        CK(spiht_dev_download(ctx, maxn, d_mn[s], (uint64_t)B));
        CK(spiht_dev_download(ctx, rec, d_rec[s], nrec * 8));
        uint64_t hs = 1469598103934665603ull;
        for (int64_t b = 0; b < B; b++) {  /* the bytes that belong to the stream, image by image */
            hs ^= fnv(streams + (size_t)b * slot, (size_t)((nbits[b] + 7) / 8));
            hs *= 1099511628211ull;
        }
        printf("step %d streams %016llx nbits0 %llu maxn0 %u pictures %016llx\n", s, (unsigned long long)hs,
               (unsigned long long)nbits[0], (unsigned)maxn[0], (unsigned long long)fnv(rec, nrec * 8));
    }
    spiht_pipeline_destroy(pl);
    spiht_ctx_destroy(ctx);
    return 0;
}
