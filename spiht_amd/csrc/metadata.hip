// decode_with_metadata (gfx950): turns the position-indexed trace written by k_decode<true> into the
// reference's int32[nbits+1, 8] table (/root/reference/src/encoder_decoder.rs:616-684):
//   [action, local_h, local_w, channel, filter, depth, n, value of the coefficient before the bit is read]
// Seven of the eight columns are a pure function of the trace record (k_meta_rows).  The eighth, the running
// value of the coefficient, depends on every earlier write to the same coefficient -- and on trees with
// duplicated nodes (odd ll_h / ll_w) two list entries write the same cell -- so the records are
// stable-sorted by node index (rocPRIM radix sort; position order is kept inside a node) and one thread per node
// replays that node's few writes in stream order (k_meta_fold).  This is the sequential semantics of the
// reference for any byte string, not only for encoder-produced streams.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#define META_KEY_NONE 0x10000000u  // sorts after every node index (< 2^28)

__global__ __launch_bounds__(256) void k_meta_keys(const uint32_t *__restrict__ tr_ent, const uint8_t *__restrict__ tr_act,
                                                   uint64_t rows, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < rows; q += (uint64_t)gridDim.x * blockDim.x) {
        keys[q] = tr_act[q] == TR_NONE ? META_KEY_NONE : (tr_ent[q] & ENT_IDX_META);
        vals[q] = (uint32_t)q;
    }
}

// Rust `f32 as i32`: toward zero, saturating, NaN -> 0
__device__ __forceinline__ int32_t f32_as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)v;
}

// get_local_position, encoder_decoder.rs:593-613: single-precision divide, multiply, subtract, each rounded
__device__ __forceinline__ int32_t local_coord(uint32_t x, int32_t start, int32_t extent) {
    const float num = __fsub_rn((float)x, (float)start);
    const float frac = __fdiv_rn(num, (float)extent);
    return f32_as_i32(__fsub_rn(__fmul_rn(frac, 200000.0f), 100000.0f));
}

__global__ __launch_bounds__(256) void k_meta_rows(MetaArgs a) {
    const Geom g = a.g;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < a.rows; q += (uint64_t)gridDim.x * blockDim.x) {
        int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
        const uint32_t ac = a.tr_act[q];
        if (ac != TR_NONE) {
            const uint32_t e = a.tr_ent[q];
            const uint32_t idx = e & ENT_IDX_META;
            const uint32_t filter = (e >> ENT_FILT_SHIFT) & 3u;
            const uint32_t k = fdiv(idx, g.div_hw);
            const uint32_t rem = idx - k * g.hw;
            const uint32_t i = fdiv(rem, g.div_w);
            const uint32_t j = rem - i * (uint32_t)g.w;
            // generation = halvings until the index-tree ancestor is a root (the tree is index based outside LL)
            uint32_t t = 0, ii = i, jj = j;
            while (!(ii < (uint32_t)g.ll_h && jj < (uint32_t)g.ll_w)) { ii >>= 1; jj >>= 1; t++; }
            int32_t lh, lw;
            if (t == 0) {  // :597-600: LL does not subtract the slice start
                lh = local_coord(i, 0, a.slices[1]);
                lw = local_coord(j, 0, a.slices[3]);
            } else {       // :601-609: other_slices[level-1-depth][filter-1]
                const int32_t *s = a.slices + 4 + ((t - 1) * 3 + (filter - 1)) * 4;
                lh = local_coord(i, s[0], s[1] - s[0]);
                lw = local_coord(j, s[2], s[3] - s[2]);
            }
            r0 = make_int4((int)(ac & 7u), lh, lw, (int)k);
            r1 = make_int4((int)filter, a.level - (int)t, (int)(ac >> 3), 0);
        }
        int4 *row = reinterpret_cast<int4 *>(a.meta + q * 8);
        row[0] = r0;
        row[1] = r1;
    }
}

// encoder_decoder.rs:14-29
__device__ __forceinline__ int32_t meta_set_bit(int32_t x, uint32_t n, uint32_t bit) {
    const uint32_t m = 1u << n;
    if (x >= 0) return bit ? (int32_t)((uint32_t)x | m) : (int32_t)((uint32_t)x & ~m);
    uint32_t v = (uint32_t)(-x);
    v = bit ? (v | m) : (v & ~m);
    return -(int32_t)v;
}

// one thread per node: walk the node's records in stream order, note the value before each record, apply the writes
// (action 1 / 4: sign bit -> +-1.5*2^n, :714-724 / :751-761; action 6: refinement bit, :825)
__global__ __launch_bounds__(256) void k_meta_fold(MetaArgs a) {
    const uint64_t nbits = a.rows - 1;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.rows; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = a.skey[s];
        if (key == META_KEY_NONE) continue;
        if (s > 0 && a.skey[s - 1] == key) continue;  // not the first record of its node
        int32_t x = 0;
        for (uint64_t r = s; r < a.rows && a.skey[r] == key; r++) {
            const uint64_t q = a.spos[r];
            a.meta[q * 8 + 7] = x;
            if (q >= nbits) continue;  // the waiting operation never got its bit
            const uint32_t ac = a.tr_act[q], act = ac & 7u, n = ac >> 3;
            const uint32_t bit = (a.data[q >> 3] >> (q & 7)) & 1u;
            if (act == 1u || act == 4u) {
                const int32_t base = n == 0 ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));
                x = bit ? base : -base;
            } else if (act == 6u) {
                x = meta_set_bit(x, n, bit);
            }
        }
    }
}

// Progressive decoding to K bit budgets from ONE walk of the stream (the pattern of the reference's make_gif.py:46-61:
// decode(bytes[:k]) for many k).  The decoder's state after b bits is the state after the operations
// of stream positions < b, and the sorted trace has every node's operations in stream order: one thread per node replays
// them once and leaves the node's value in out[kk] whenever it passes budgets[kk] (ascending).  out: [K][g.n], zeroed
// by the caller (a node without operations stays 0).
__global__ __launch_bounds__(256) void k_budget_fold(MetaArgs a, const uint64_t *__restrict__ budgets, int K, int32_t *__restrict__ out) {
    const uint64_t nbits = a.rows - 1;
    const size_t n_cells = a.g.n;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.rows; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = a.skey[s];
        if (key == META_KEY_NONE) continue;
        if (s > 0 && a.skey[s - 1] == key) continue;  // not the first record of its node
        int32_t x = 0;
        int kk = 0;
        for (uint64_t r = s; r < a.rows && a.skey[r] == key; r++) {
            const uint64_t q = a.spos[r];
            for (; kk < K && budgets[kk] <= q; kk++)  // budgets that end in front of this operation
                if (x) out[(size_t)kk * n_cells + key] = x;
            if (q >= nbits) continue;  // the waiting operation never got its bit
            const uint32_t ac = a.tr_act[q], act = ac & 7u, n = ac >> 3;
            const uint32_t bit = (a.data[q >> 3] >> (q & 7)) & 1u;
            if (act == 1u || act == 4u) {
                const int32_t base = n == 0 ? 1 : (int32_t)((1u << (n - 1)) + (1u << n));
                x = bit ? base : -base;
            } else if (act == 6u) {
                x = meta_set_bit(x, n, bit);
            }
        }
        for (; kk < K; kk++)
            if (x) out[(size_t)kk * n_cells + key] = x;
    }
}

extern "C" int spiht_meta_sort_temp_bytes(uint64_t rows, size_t *bytes) {
    size_t sz = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sz, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (size_t)rows, 0u, 29u, (hipStream_t)0);
    *bytes = sz;
    return (int)e;
}

// a->skey / a->spos must point to keys_out / vals_out
extern "C" int spiht_launch_metadata(const MetaArgs *a, uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out,
                                     uint32_t *vals_out, void *temp, size_t temp_bytes, hipStream_t st) {
    const uint64_t rows = a->rows;
    const int grid = (int)std::min<uint64_t>((rows + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(k_meta_keys, dim3(grid), dim3(256), 0, st, a->tr_ent, a->tr_act, rows, keys_in, vals_in);
    hipLaunchKernelGGL(k_meta_rows, dim3(grid), dim3(256), 0, st, *a);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)rows, 0u, 29u, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_meta_fold, dim3(grid), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}

// the trace sorted by (node, position) as above, then k_budget_fold.  a->skey / a->spos must point to keys_out / vals_out
extern "C" int spiht_launch_budget_fold(const MetaArgs *a, uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out,
                                        uint32_t *vals_out, void *temp, size_t temp_bytes, const uint64_t *d_budgets, int K,
                                        int32_t *d_out, hipStream_t st) {
    const uint64_t rows = a->rows;
    const int grid = (int)std::min<uint64_t>((rows + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(k_meta_keys, dim3(grid), dim3(256), 0, st, a->tr_ent, a->tr_act, rows, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)rows, 0u, 29u, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_budget_fold, dim3(grid), dim3(256), 0, st, *a, d_budgets, K, d_out);
    return (int)hipGetLastError();
}
