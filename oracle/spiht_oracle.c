/*
 * oracle/spiht_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the reference's SPIHT bit-plane coder
 * (/root/reference/src/encoder_decoder.rs and the bit<->byte packing of
 * /root/reference/src/lib.rs).  It exists to check the HIP path and to serve
 * as the CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Pinning status: the Rust reference cannot be built in this image (no
 * cargo/rustc, crates not vendored), and the reference ships no golden
 * bitstream.  The restatement is pinned by
 *   (i)   the known answers in the reference's own Rust unit tests
 *         (encoder_decoder.rs:851-862, 865-875, 988-1009),
 *   (ii)  the round-trip properties of encoder_decoder.rs:878-985,
 *   (iii) bit lists captured from the reference's importable pure-Python
 *         twin spiht/spiht_py.py (tests/golden/), which shares the list
 *         logic and differs only in the two predicates selected by
 *         `rule` below (spiht_py.py:35-39 and :118).
 *
 * `rule`: 0 = Rust semantics (the contract), 1 = "py-compat" (spiht_py.py).
 *
 * Each function cites the reference lines it follows.  This is deliberately
 * the slow literal algorithm (recursive significance search, FIFO lists,
 * one length check per pushed bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_LL 1     /* assert!(ll_h > 1); assert!(ll_w > 1)  encoder_decoder.rs:160-161,310-311 */
#define ORC_ERR_EMPTY 2  /* .max().unwrap() on an empty array      encoder_decoder.rs:165 */
#define ORC_ERR_NOMEM 3

/* ---- array view with element strides (lib.rs:27 `x.as_array()` is a strided view) ---- */
typedef struct {
    const int32_t *p;
    int64_t c, h, w;
    int64_t sc, sh, sw;
} view3;

static inline int32_t at(const view3 *a, int64_t k, int64_t i, int64_t j) {
    return a->p[k * a->sc + i * a->sh + j * a->sw];
}

/* ---- growable bit vector (bitvec::BitVec stand-in) ---- */
typedef struct {
    uint8_t *bits; /* one bit per byte */
    uint64_t len, cap;
} bitvec;

static int bv_push(bitvec *b, int bit) {
    if (b->len == b->cap) {
        uint64_t nc = b->cap ? b->cap * 2 : 4096;
        uint8_t *nb = (uint8_t *)realloc(b->bits, nc);
        if (!nb) return -1;
        b->bits = nb;
        b->cap = nc;
    }
    b->bits[b->len++] = (uint8_t)(bit != 0);
    return 0;
}

/* ---- FIFO of (type,k,i,j) (VecDeque stand-in) ---- */
typedef struct {
    uint8_t t;
    uint32_t k, i, j;
} ent;
typedef struct {
    ent *e;
    uint64_t head, tail, cap;
} fifo;

static int fifo_push(fifo *f, uint8_t t, uint32_t k, uint32_t i, uint32_t j) {
    if (f->tail == f->cap) {
        if (f->head > f->cap / 2) { /* compact */
            memmove(f->e, f->e + f->head, (f->tail - f->head) * sizeof(ent));
            f->tail -= f->head;
            f->head = 0;
        } else {
            uint64_t nc = f->cap ? f->cap * 2 : 1024;
            ent *ne = (ent *)realloc(f->e, nc * sizeof(ent));
            if (!ne) return -1;
            f->e = ne;
            f->cap = nc;
        }
    }
    f->e[f->tail].t = t;
    f->e[f->tail].k = k;
    f->e[f->tail].i = i;
    f->e[f->tail].j = j;
    f->tail++;
    return 0;
}
static inline uint64_t fifo_len(const fifo *f) { return f->tail - f->head; }
static inline void fifo_free(fifo *f) {
    free(f->e);
    memset(f, 0, sizeof(*f));
}

/* encoder_decoder.rs:7-12 (rule 0) / spiht_py.py:35-39 (rule 1) */
static int has_descendents_past_offspring(int64_t i, int64_t j, int64_t h, int64_t w, int rule) {
    if (rule == 1) {
        if (2 * i + 1 >= h || 2 * j + 1 >= w) return 0;
        return 1;
    }
    if ((i * 2 + 1) * 2 + 1 >= h || (j * 2 + 1) * 2 + 1 >= w) return 0;
    return 1;
}

/* encoder_decoder.rs:14-29 */
int32_t orc_set_bit(int32_t x, uint8_t n, int bit) {
    int sign = x >= 0;
    if (bit) {
        if (sign) return x | (int32_t)(1u << n);
        return -((-x) | (int32_t)(1u << n));
    } else {
        if (sign) return x & ~(int32_t)(1u << n);
        return -((-x) & ~(int32_t)(1u << n));
    }
}

static inline int32_t iabs32(int32_t x) { return x < 0 ? -x : x; }

/* encoder_decoder.rs:31-34 */
int orc_is_bit_set(int32_t x, uint8_t n) { return (iabs32(x) & (int32_t)(1u << n)) != 0; }

/* encoder_decoder.rs:37-41 */
int orc_is_element_sig(int32_t x, uint8_t n) { return iabs32(x) >= (int32_t)(1u << n); }

/* encoder_decoder.rs:43-75; returns 0 for None, 1 for Some and fills o[4][2] */
int orc_get_offspring(int64_t i, int64_t j, int64_t h, int64_t w, int64_t ll_h, int64_t ll_w,
                      int64_t o[4][2]) {
    if (i < ll_h && j < ll_w) {
        if (i % 2 == 0 && j % 2 == 0) return 0;
        int64_t sub_i = i / 2 * 2, sub_j = j / 2 * 2;
        int64_t chunk_i = i % 2, chunk_j = j % 2;
        o[0][0] = chunk_i * ll_h + sub_i;     o[0][1] = chunk_j * ll_w + sub_j;
        o[1][0] = chunk_i * ll_h + sub_i;     o[1][1] = chunk_j * ll_w + sub_j + 1;
        o[2][0] = chunk_i * ll_h + sub_i + 1; o[2][1] = chunk_j * ll_w + sub_j;
        o[3][0] = chunk_i * ll_h + sub_i + 1; o[3][1] = chunk_j * ll_w + sub_j + 1;
        return 1;
    }
    if (2 * i + 1 >= h || 2 * j + 1 >= w) return 0;
    o[0][0] = 2 * i;     o[0][1] = 2 * j;
    o[1][0] = 2 * i;     o[1][1] = 2 * j + 1;
    o[2][0] = 2 * i + 1; o[2][1] = 2 * j;
    o[3][0] = 2 * i + 1; o[3][1] = 2 * j + 1;
    return 1;
}

/* encoder_decoder.rs:78-99 */
static int is_set_sig(const view3 *a, int64_t k, int64_t i, int64_t j, uint8_t n, int64_t ll_h,
                      int64_t ll_w) {
    if (orc_is_element_sig(at(a, k, i, j), n)) return 1;
    int64_t o[4][2];
    if (orc_get_offspring(i, j, a->h, a->w, ll_h, ll_w, o)) {
        for (int q = 0; q < 4; q++)
            if (is_set_sig(a, k, o[q][0], o[q][1], n, ll_h, ll_w)) return 1;
    }
    return 0;
}

/* encoder_decoder.rs:101-121 */
static int is_l_sig(const view3 *a, int64_t k, int64_t i, int64_t j, uint8_t n, int64_t ll_h,
                    int64_t ll_w) {
    int64_t o[4][2], s[4][2];
    if (orc_get_offspring(i, j, a->h, a->w, ll_h, ll_w, o)) {
        for (int q = 0; q < 4; q++) {
            if (orc_get_offspring(o[q][0], o[q][1], a->h, a->w, ll_h, ll_w, s)) {
                for (int r = 0; r < 4; r++)
                    if (is_set_sig(a, k, s[r][0], s[r][1], n, ll_h, ll_w)) return 1;
            }
        }
    }
    return 0;
}

/* `(max as f32).log2() as u8` encoder_decoder.rs:166 (rule 0);
 * `math.floor(math.log2(int(max)))` spiht_py.py:118 (rule 1). */
uint8_t orc_start_plane(int32_t maxabs, int rule) {
    if (maxabs <= 0) return 0; /* log2(0) = -inf, `as u8` saturates to 0 */
    if (rule == 1) return (uint8_t)floor(log2((double)maxabs));
    float l = log2f((float)maxabs);
    if (!(l > 0.0f)) return 0;
    if (l >= 255.0f) return 255;
    return (uint8_t)l;
}

/* encoder_decoder.rs:155-303.  On success *bits_out is a malloc'd array with one
 * bit per byte, *nbits_out its length. */
int orc_encode_bits(const int32_t *x, int64_t c, int64_t h, int64_t w, int64_t sc, int64_t sh,
                    int64_t sw, int64_t ll_h, int64_t ll_w, uint64_t max_bits, int rule,
                    uint8_t **bits_out, uint64_t *nbits_out, uint8_t *max_n_out) {
    view3 a = {x, c, h, w, sc, sh, sw};
    if (!(ll_h > 1) || !(ll_w > 1)) return ORC_ERR_LL;
    if (c <= 0 || h <= 0 || w <= 0) return ORC_ERR_EMPTY;

    bitvec data = {0, 0, 0};
    int32_t max = 0;
    for (int64_t k = 0; k < c; k++)
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++) {
                int32_t v = iabs32(at(&a, k, i, j));
                if (v > max) max = v;
            }
    uint8_t n = orc_start_plane(max, rule);
    uint8_t max_n = n;
    int rc = ORC_OK;

    fifo lsp = {0}, lip = {0}, lis = {0}, lip_retain = {0}, lis_retain = {0};
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++)
            for (int64_t k = 0; k < c; k++)
                if (fifo_push(&lip, 0, (uint32_t)k, (uint32_t)i, (uint32_t)j)) goto nomem;
    /* type: 1 = A, 0 = B */
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++) {
            if (i % 2 == 0 && j % 2 == 0) continue;
            for (int64_t k = 0; k < c; k++)
                if (fifo_push(&lis, 1, (uint32_t)k, (uint32_t)i, (uint32_t)j)) goto nomem;
        }

#define PUSH_BIT(b)                                   \
    do {                                              \
        if (bv_push(&data, (b))) goto nomem;          \
        if (data.len == max_bits) goto done;          \
    } while (0)

    for (;;) {
        uint64_t lsp_len = fifo_len(&lsp);

        lip_retain.head = lip_retain.tail = 0;
        while (fifo_len(&lip)) {
            ent e = lip.e[lip.head++];
            int32_t v = at(&a, e.k, e.i, e.j);
            int is_sig = orc_is_element_sig(v, n);
            PUSH_BIT(is_sig);
            if (is_sig) {
                if (fifo_push(&lsp, 0, e.k, e.i, e.j)) goto nomem;
                PUSH_BIT(v >= 0);
            } else {
                if (fifo_push(&lip_retain, 0, e.k, e.i, e.j)) goto nomem;
            }
        }
        { fifo t = lip; lip = lip_retain; lip_retain = t; }

        lis_retain.head = lis_retain.tail = 0;
        while (fifo_len(&lis)) {
            ent e = lis.e[lis.head++];
            int64_t o[4][2];
            if (e.t) {
                int desc_sig = 0;
                int has = orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o);
                if (has)
                    for (int q = 0; q < 4; q++)
                        if (is_set_sig(&a, e.k, o[q][0], o[q][1], n, ll_h, ll_w)) {
                            desc_sig = 1;
                            break;
                        }
                PUSH_BIT(desc_sig);
                if (desc_sig) {
                    for (int q = 0; q < 4; q++) {
                        int32_t v = at(&a, e.k, o[q][0], o[q][1]);
                        int sig = orc_is_element_sig(v, n);
                        PUSH_BIT(sig);
                        if (sig) {
                            if (fifo_push(&lsp, 0, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                            PUSH_BIT(v >= 0);
                        } else {
                            if (fifo_push(&lip, 0, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                        }
                    }
                    if (has_descendents_past_offspring(e.i, e.j, h, w, rule))
                        if (fifo_push(&lis, 0, e.k, e.i, e.j)) goto nomem;
                } else {
                    if (fifo_push(&lis_retain, e.t, e.k, e.i, e.j)) goto nomem;
                }
            } else {
                int l_sig = is_l_sig(&a, e.k, e.i, e.j, n, ll_h, ll_w);
                PUSH_BIT(l_sig);
                if (l_sig) {
                    if (orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o))
                        for (int q = 0; q < 4; q++)
                            if (fifo_push(&lis, 1, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                } else {
                    if (fifo_push(&lis_retain, e.t, e.k, e.i, e.j)) goto nomem;
                }
            }
        }
        { fifo t = lis; lis = lis_retain; lis_retain = t; }

        for (uint64_t q = 0; q < lsp_len; q++) {
            ent e = lsp.e[lsp.head + q];
            PUSH_BIT(orc_is_bit_set(at(&a, e.k, e.i, e.j), n));
        }

        if (n == 0) break;
        n -= 1;
    }
#undef PUSH_BIT
done:
    *bits_out = data.bits;
    *nbits_out = data.len;
    *max_n_out = max_n;
    goto cleanup;
nomem:
    rc = ORC_ERR_NOMEM;
    free(data.bits);
cleanup:
    fifo_free(&lsp); fifo_free(&lip); fifo_free(&lis); fifo_free(&lip_retain); fifo_free(&lis_retain);
    return rc;
}

/* encoder_decoder.rs:307-454.  `bits` has one bit per byte. `out` is c*h*w, C-contiguous,
 * zeroed here (Array3::zeros, :308). */
int orc_decode_bits(const uint8_t *bits, uint64_t nbits, uint8_t n, int64_t c, int64_t h, int64_t w,
                    int64_t ll_h, int64_t ll_w, int rule, int32_t *out) {
    if (!(ll_h > 1) || !(ll_w > 1)) return ORC_ERR_LL;
    memset(out, 0, (size_t)(c * h * w) * sizeof(int32_t));
    int rc = ORC_OK;
    uint64_t cur = 0;
#define REC(k, i, j) out[((int64_t)(k) * h + (int64_t)(i)) * w + (int64_t)(j)]
#define POP_BIT(dst)                      \
    do {                                  \
        if (cur >= nbits) goto done;      \
        (dst) = bits[cur++];              \
    } while (0)

    fifo lsp = {0}, lip = {0}, lis = {0}, lip_retain = {0}, lis_retain = {0};
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++)
            for (int64_t k = 0; k < c; k++)
                if (fifo_push(&lip, 0, (uint32_t)k, (uint32_t)i, (uint32_t)j)) goto nomem;
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++) {
            if (i % 2 == 0 && j % 2 == 0) continue;
            for (int64_t k = 0; k < c; k++)
                if (fifo_push(&lis, 1, (uint32_t)k, (uint32_t)i, (uint32_t)j)) goto nomem;
        }

    for (;;) {
        uint64_t lsp_len = fifo_len(&lsp);
        /* :364-370 */
        int32_t base_sig = (n == 0) ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));

        lip_retain.head = lip_retain.tail = 0;
        while (fifo_len(&lip)) {
            ent e = lip.e[lip.head++];
            int is_sig, sb;
            POP_BIT(is_sig);
            if (is_sig) {
                if (fifo_push(&lsp, 0, e.k, e.i, e.j)) goto nomem;
                POP_BIT(sb);
                REC(e.k, e.i, e.j) = base_sig * (sb * 2 - 1);
            } else {
                if (fifo_push(&lip_retain, 0, e.k, e.i, e.j)) goto nomem;
            }
        }
        { fifo t = lip; lip = lip_retain; lip_retain = t; }

        lis_retain.head = lis_retain.tail = 0;
        while (fifo_len(&lis)) {
            ent e = lis.e[lis.head++];
            int64_t o[4][2];
            if (e.t) {
                int desc_sig;
                POP_BIT(desc_sig);
                if (desc_sig) {
                    if (orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o)) {
                        for (int q = 0; q < 4; q++) {
                            int sig, sb;
                            POP_BIT(sig);
                            if (sig) {
                                if (fifo_push(&lsp, 0, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                                POP_BIT(sb);
                                REC(e.k, o[q][0], o[q][1]) = (sb * 2 - 1) * base_sig;
                            } else {
                                if (fifo_push(&lip, 0, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                            }
                        }
                    }
                    if (has_descendents_past_offspring(e.i, e.j, h, w, rule))
                        if (fifo_push(&lis, 0, e.k, e.i, e.j)) goto nomem;
                } else {
                    if (fifo_push(&lis_retain, e.t, e.k, e.i, e.j)) goto nomem;
                }
            } else {
                int l_sig;
                POP_BIT(l_sig);
                if (l_sig) {
                    if (orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o))
                        for (int q = 0; q < 4; q++)
                            if (fifo_push(&lis, 1, e.k, (uint32_t)o[q][0], (uint32_t)o[q][1])) goto nomem;
                } else {
                    if (fifo_push(&lis_retain, e.t, e.k, e.i, e.j)) goto nomem;
                }
            }
        }
        { fifo t = lis; lis = lis_retain; lis_retain = t; }

        for (uint64_t q = 0; q < lsp_len; q++) {
            ent e = lsp.e[lsp.head + q];
            int b;
            POP_BIT(b);
            REC(e.k, e.i, e.j) = orc_set_bit(REC(e.k, e.i, e.j), n, b);
        }

        if (n == 0) break;
        n -= 1;
    }
#undef POP_BIT
#undef REC
done:
    goto cleanup;
nomem:
    rc = ORC_ERR_NOMEM;
cleanup:
    fifo_free(&lsp); fifo_free(&lip); fifo_free(&lis); fifo_free(&lip_retain); fifo_free(&lis_retain);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * decode_with_metadata (encoder_decoder.rs:631-841).  Parity of the metadata rows is UNPINNED: the
 * reference holds no expected rows anywhere (its tests only compare the decoded array, :929-966,
 * test_spiht.py:19-28) and the Rust core cannot be built here; this is a restatement of the source.
 * ------------------------------------------------------------------------------------------------ */

/* CoefficientMetadata, encoder_decoder.rs:123-151 (depth is a u8 and wraps in a release build) */
typedef struct {
    uint8_t t, depth, filter;
    uint32_t k, i, j;
} ment;
typedef struct {
    ment *e;
    uint64_t head, tail, cap;
} mfifo;

static int mfifo_push(mfifo *f, ment m) {
    if (f->tail == f->cap) {
        if (f->head > f->cap / 2) {
            memmove(f->e, f->e + f->head, (f->tail - f->head) * sizeof(ment));
            f->tail -= f->head;
            f->head = 0;
        } else {
            uint64_t nc = f->cap ? f->cap * 2 : 1024;
            ment *ne = (ment *)realloc(f->e, nc * sizeof(ment));
            if (!ne) return -1;
            f->e = ne;
            f->cap = nc;
        }
    }
    f->e[f->tail++] = m;
    return 0;
}
static inline uint64_t mfifo_len(const mfifo *f) { return f->tail - f->head; }

/* encoder_decoder.rs:133-150 */
static uint8_t offspring_filter(const ment *m) {
    if (m->filter == 0) {
        if (m->i % 2 == 1 && m->j % 2 == 1) return 3; /* DD */
        if (m->i % 2 == 0 && m->j % 2 != 0) return 2; /* AD */
        return 1;                                     /* DA */
    }
    return m->filter;
}

/* Rust `f32 as i32`: truncation toward zero, saturating, NaN -> 0 */
static int32_t f32_as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

#define ORC_ERR_SLICE 4 /* index out of bounds in get_local_position (:603) -> Rust panic */

/* encoder_decoder.rs:593-613.  top = {start_i,end_i,start_j,end_j}; other = [level][3][4] same order.
 * volatile keeps gcc from contracting or reassociating the f32 steps. */
static int local_position(const ment *m, const int64_t *top, const int64_t *other, int64_t level, int32_t *lh,
                          int32_t *lw) {
    volatile float local_h, local_w;
    if (m->depth == (uint8_t)level) {
        local_h = (float)m->i / (float)top[1];
        local_w = (float)m->j / (float)top[3];
    } else {
        uint8_t depth_i = (uint8_t)((uint8_t)level - 1 - m->depth);
        uint64_t filter_i = (uint64_t)m->filter - 1;
        if (depth_i >= level || filter_i >= 3) return ORC_ERR_SLICE;
        const int64_t *s = other + ((int64_t)depth_i * 3 + (int64_t)filter_i) * 4;
        volatile float a = (float)m->i - (float)s[0];
        volatile float b = (float)(uint64_t)(s[1] - s[0]);
        local_h = a / b;
        a = (float)m->j - (float)s[2];
        b = (float)(uint64_t)(s[3] - s[2]);
        local_w = a / b;
    }
    volatile float th = local_h * 200000.0f, tw = local_w * 200000.0f;
    volatile float uh = th - 100000.0f, uw = tw - 100000.0f;
    *lh = f32_as_i32(uh);
    *lw = f32_as_i32(uw);
    return ORC_OK;
}

/* encoder_decoder.rs:631-841.  `bits` one bit per byte; out: c*h*w (zeroed here); meta: (nbits+1)*8 (zeroed here). */
int orc_decode_with_metadata_bits(const uint8_t *bits, uint64_t nbits, uint8_t n, int64_t c, int64_t h, int64_t w,
                                  int64_t ll_h, int64_t ll_w, const int64_t *top, const int64_t *other, int64_t level,
                                  int32_t *out, int32_t *meta) {
    memset(out, 0, (size_t)(c * h * w) * sizeof(int32_t));
    memset(meta, 0, (size_t)(nbits + 1) * 8 * sizeof(int32_t));
    if (!(ll_h > 1) || !(ll_w > 1)) return ORC_ERR_LL;
    int rc = ORC_OK;
    uint64_t cur = 0;
#define REC(k, i, j) out[((int64_t)(k) * h + (int64_t)(i)) * w + (int64_t)(j)]
#define POP_BIT(dst)                 \
    do {                             \
        if (cur >= nbits) goto done; \
        (dst) = bits[cur++];         \
    } while (0)
/* :664-684; `metadata_arr.len()` (:668) is the element count, so that guard never trips before pop_bit's */
#define ASSIGN(action, m)                                               \
    do {                                                                \
        int32_t lh_, lw_;                                               \
        rc = local_position(&(m), top, other, level, &lh_, &lw_);       \
        if (rc) goto cleanup;                                           \
        int32_t *row = meta + cur * 8;                                  \
        row[0] = (action); row[1] = lh_; row[2] = lw_;                  \
        row[3] = (int32_t)(m).k; row[4] = (m).filter; row[5] = (m).depth; \
        row[6] = n; row[7] = REC((m).k, (m).i, (m).j);                  \
    } while (0)

    mfifo lsp = {0}, lip = {0}, lis = {0}, lip_retain = {0}, lis_retain = {0};
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++)
            for (int64_t k = 0; k < c; k++) {
                ment m = {0, (uint8_t)level, 0, (uint32_t)k, (uint32_t)i, (uint32_t)j};
                if (mfifo_push(&lip, m)) goto nomem;
            }
    for (int64_t i = 0; i < ll_h; i++)
        for (int64_t j = 0; j < ll_w; j++) {
            if (i % 2 == 0 && j % 2 == 0) continue;
            for (int64_t k = 0; k < c; k++) {
                ment m = {1, (uint8_t)level, 0, (uint32_t)k, (uint32_t)i, (uint32_t)j};
                if (mfifo_push(&lis, m)) goto nomem;
            }
        }

    for (;;) {
        uint64_t lsp_len = mfifo_len(&lsp);
        int32_t base_sig = (n == 0) ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));

        lip_retain.head = lip_retain.tail = 0;
        while (mfifo_len(&lip)) {
            ment e = lip.e[lip.head++];
            int is_sig, sb;
            ASSIGN(0, e);
            POP_BIT(is_sig);
            if (is_sig) {
                ASSIGN(1, e);
                POP_BIT(sb);
                REC(e.k, e.i, e.j) = base_sig * (sb * 2 - 1);
                if (mfifo_push(&lsp, e)) goto nomem;
            } else {
                if (mfifo_push(&lip_retain, e)) goto nomem;
            }
        }
        { mfifo t = lip; lip = lip_retain; lip_retain = t; }

        lis_retain.head = lis_retain.tail = 0;
        while (mfifo_len(&lis)) {
            ment e = lis.e[lis.head++];
            int64_t o[4][2];
            if (e.t) {
                int desc_sig;
                ASSIGN(2, e);
                POP_BIT(desc_sig);
                if (desc_sig) {
                    if (orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o)) {
                        for (int q = 0; q < 4; q++) {
                            ment nc = {0, (uint8_t)(e.depth - 1), offspring_filter(&e), e.k, (uint32_t)o[q][0],
                                       (uint32_t)o[q][1]};
                            int sig, sb;
                            ASSIGN(3, nc);
                            POP_BIT(sig);
                            if (sig) {
                                ASSIGN(4, nc);
                                POP_BIT(sb);
                                REC(nc.k, nc.i, nc.j) = (sb * 2 - 1) * base_sig;
                                if (mfifo_push(&lsp, nc)) goto nomem;
                            } else {
                                if (mfifo_push(&lip, nc)) goto nomem;
                            }
                        }
                    }
                    if (has_descendents_past_offspring(e.i, e.j, h, w, 0)) {
                        ment b = e;
                        b.t = 0;
                        if (mfifo_push(&lis, b)) goto nomem;
                    }
                } else {
                    if (mfifo_push(&lis_retain, e)) goto nomem;
                }
            } else {
                int l_sig;
                ASSIGN(5, e);
                POP_BIT(l_sig);
                if (l_sig) {
                    if (orc_get_offspring(e.i, e.j, h, w, ll_h, ll_w, o))
                        for (int q = 0; q < 4; q++) {
                            ment nc = {1, (uint8_t)(e.depth - 1), offspring_filter(&e), e.k, (uint32_t)o[q][0],
                                       (uint32_t)o[q][1]};
                            if (mfifo_push(&lis, nc)) goto nomem;
                        }
                } else {
                    if (mfifo_push(&lis_retain, e)) goto nomem;
                }
            }
        }
        { mfifo t = lis; lis = lis_retain; lis_retain = t; }

        for (uint64_t q = 0; q < lsp_len; q++) {
            ment e = lsp.e[lsp.head + q];
            int b;
            ASSIGN(6, e);
            POP_BIT(b);
            REC(e.k, e.i, e.j) = orc_set_bit(REC(e.k, e.i, e.j), n, b);
        }

        if (n == 0) break;
        n -= 1;
    }
#undef POP_BIT
#undef ASSIGN
#undef REC
done:
    goto cleanup;
nomem:
    rc = ORC_ERR_NOMEM;
cleanup:
    free(lsp.e); free(lip.e); free(lis.e); free(lip_retain.e); free(lis_retain.e);
    return rc;
}

/* lib.rs:47-56 with bytes_to_bits lib.rs:15-21 */
int orc_decode_with_metadata(const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h, int64_t w,
                             int64_t ll_h, int64_t ll_w, const int64_t *top, const int64_t *other, int64_t level,
                             int32_t *out, int32_t *meta) {
    uint64_t nbits = nbytes * 8;
    uint8_t *bits = (uint8_t *)malloc(nbits ? nbits : 1);
    if (!bits) return ORC_ERR_NOMEM;
    for (uint64_t t = 0; t < nbits; t++) bits[t] = (data[t >> 3] >> (t & 7)) & 1;
    int rc = orc_decode_with_metadata_bits(bits, nbits, n, c, h, w, ll_h, ll_w, top, other, level, out, meta);
    free(bits);
    return rc;
}

void orc_free(void *p) { free(p); }

/* lib.rs:24-32: encode + `data.chunks(8).map(load_le::<u8>)` (bit t -> bit t%8 of byte t/8).
 * *bytes_out is malloc'd, length ceil(nbits/8). */
int orc_encode(const int32_t *x, int64_t c, int64_t h, int64_t w, int64_t sc, int64_t sh, int64_t sw,
               int64_t ll_h, int64_t ll_w, uint64_t max_bits, int rule, uint8_t **bytes_out,
               uint64_t *nbits_out, uint8_t *max_n_out) {
    uint8_t *bits = 0;
    uint64_t nbits = 0;
    int rc = orc_encode_bits(x, c, h, w, sc, sh, sw, ll_h, ll_w, max_bits, rule, &bits, &nbits, max_n_out);
    if (rc) return rc;
    uint64_t nbytes = (nbits + 7) / 8;
    uint8_t *out = (uint8_t *)calloc(nbytes ? nbytes : 1, 1);
    if (!out) { free(bits); return ORC_ERR_NOMEM; }
    for (uint64_t t = 0; t < nbits; t++)
        if (bits[t]) out[t >> 3] |= (uint8_t)(1u << (t & 7));
    free(bits);
    *bytes_out = out;
    *nbits_out = nbits;
    return ORC_OK;
}

/* lib.rs:35-42 with bytes_to_bits lib.rs:15-21: ALL 8*nbytes bits are data. */
int orc_decode(const uint8_t *data, uint64_t nbytes, uint8_t n, int64_t c, int64_t h, int64_t w,
               int64_t ll_h, int64_t ll_w, int rule, int32_t *out) {
    uint64_t nbits = nbytes * 8;
    uint8_t *bits = (uint8_t *)malloc(nbits ? nbits : 1);
    if (!bits) return ORC_ERR_NOMEM;
    for (uint64_t t = 0; t < nbits; t++) bits[t] = (data[t >> 3] >> (t & 7)) & 1;
    int rc = orc_decode_bits(bits, nbits, n, c, h, w, ll_h, ll_w, rule, out);
    free(bits);
    return rc;
}

/* Set significance of every node, evaluated with the reference's own recursion (test helper for the GPU's
 * significance pyramid, which replaces it):
 *   dcode[k,i,j] = 1 + the largest plane n at which the type-A test of encoder_decoder.rs:228-237 passes for
 *                  node (k,i,j) -- any offspring o with is_set_sig(o, n), :78-99 -- or 0 if it passes at no plane;
 *   lcode[k,i,j] = the same for the type-B test is_l_sig (:101-121);
 *   has[k,i,j]   = 1 iff get_offspring (:43-75) returns Some for the node.
 * "significant at plane n"  <=>  code > n.  Nodes without offspring get 0. */
int orc_set_codes(const int32_t *x, int64_t c, int64_t h, int64_t w, int64_t ll_h, int64_t ll_w, uint8_t *dcode,
                  uint8_t *lcode, uint8_t *has) {
    view3 a = {x, c, h, w, h * w, w, 1};
    if (!(ll_h > 1) || !(ll_w > 1)) return ORC_ERR_LL;
    for (int64_t k = 0; k < c; k++)
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++) {
                const int64_t t = (k * h + i) * w + j;
                int64_t o[4][2];
                dcode[t] = 0; lcode[t] = 0;
                has[t] = (uint8_t)orc_get_offspring(i, j, h, w, ll_h, ll_w, o);
                if (!has[t]) continue;
                for (int n = 30; n >= 0 && !dcode[t]; n--)
                    for (int q = 0; q < 4; q++)
                        if (is_set_sig(&a, k, o[q][0], o[q][1], (uint8_t)n, ll_h, ll_w)) { dcode[t] = (uint8_t)(n + 1); break; }
                for (int n = 30; n >= 0 && !lcode[t]; n--)
                    if (is_l_sig(&a, k, i, j, (uint8_t)n, ll_h, ll_w)) lcode[t] = (uint8_t)(n + 1);
            }
    return ORC_OK;
}
