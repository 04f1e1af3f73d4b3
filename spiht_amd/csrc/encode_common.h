// Device helpers shared by the two list-encoder kernels (encode.hip: one workgroup per image; encode_wide.hip: one image on
// several CUs): tree geometry of /root/reference/src/encoder_decoder.rs:43-75 on the packed array, list-entry flags, the
// start plane.
#pragma once
#include "common.h"

__device__ __forceinline__ uint32_t iabs_u(int32_t x) { return (uint32_t)(x < 0 ? -x : x); }


__device__ __forceinline__ void decomp(const Geom &g, uint32_t idx, uint32_t &k, uint32_t &i, uint32_t &j) {
    k = fdiv(idx, g.div_hw);
    uint32_t r = idx - k * g.hw;
    i = fdiv(r, g.div_w);
    j = r - i * (uint32_t)g.w;
}

// offspring (0,0) of node (i,j): row r, column cc (the others are +1 in either direction); returns its linear
// index within the image  (encoder_decoder.rs:43-75)
__device__ __forceinline__ uint32_t child_base(const Geom &g, uint32_t k, uint32_t i, uint32_t j, uint32_t &r, uint32_t &cc) {
    if (i < (uint32_t)g.ll_h && j < (uint32_t)g.ll_w) {
        r = (i & 1u) * (uint32_t)g.ll_h + (i & ~1u);
        cc = (j & 1u) * (uint32_t)g.ll_w + (j & ~1u);
    } else {
        r = 2 * i;
        cc = 2 * j;
    }
    return k * g.hw + r * (uint32_t)g.w + cc;
}

// type-A entry for child (ci,cj) with linear index idx: flagged leaf when it has no offspring of its own
__device__ __forceinline__ uint32_t make_a_entry(uint32_t idx, uint32_t ci, uint32_t cj, uint32_t H, uint32_t W) {
    return idx | ENT_A | ((2 * ci + 1 < H && 2 * cj + 1 < W) ? 0u : ENT_LEAF);
}

// `(max as f32).log2() as u8`  (encoder_decoder.rs:166) with the host libm's rounding (table from the host)
__device__ __forceinline__ int start_plane(uint32_t maxabs, const float *thr) {
    if (maxabs == 0) return 0;
    float m = (float)(int32_t)maxabs;
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    if (e + 1 <= 31 && m >= thr[e + 1]) e += 1;
    return e;
}

