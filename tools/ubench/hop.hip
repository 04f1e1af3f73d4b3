// Cycles per hop of the decoder's scalar walk (decode.hip, LIS pass), in isolation: one sequencer wavefront, optionally
// with idle-spinning neighbours in the workgroup as in k_decode.  Variants of the loop body are compared.
//   hipcc --offload-arch=gfx950 -O3 -o hop hop.hip && ./hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__device__ __forceinline__ void walk(uint64_t L0, uint64_t T0, uint32_t LAv, uint64_t &fm_out, uint32_t &pos_out, uint32_t &rel_out) {
    uint32_t rel = 0, pos = 0, f, dd, len;
    uint64_t fm = 0, Lr = L0, Tr = T0, c64;
    if (V == 0) {  // the loop of decode.hip
        asm volatile(
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc0 s_hd%=\n"
            "s_hl%=:\n\t"
            "s_ff1_i32_b64 %[d], %[c]\n\t"
            "s_add_i32 %[f], %[pos], %[d]\n\t"
            "s_lshr_b64 %[T], %[T], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], %[d]\n\t"
            "s_bitset1_b64 %[fm], %[f]\n\t"
            "s_lshr_b64 %[T], %[T], 1\n\t"
            "v_readlane_b32 %[len], %[LAv], %[f]\n\t"
            "s_add_i32 %[rel], %[rel], %[d]\n\t"
            "s_add_i32 %[rel], %[rel], 1\n\t"
            "s_lshr_b64 %[L], %[L], %[len]\n\t"
            "s_add_i32 %[pos], %[f], %[len]\n\t"
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc1 s_hl%=\n"
            "s_hd%=:\n\t"
            : [rel] "+s"(rel), [pos] "+s"(pos), [fm] "+s"(fm), [L] "+s"(Lr), [T] "+s"(Tr), [c] "=&s"(c64), [f] "=&s"(f),
              [d] "=&s"(dd), [len] "=&s"(len)
            : [LAv] "v"(LAv)
            : "scc");
    } else if (V == 1) {  // two hops per trip round the loop
        asm volatile(
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc0 s_hd%=\n"
            "s_hl%=:\n\t"
            "s_ff1_i32_b64 %[d], %[c]\n\t"
            "s_add_i32 %[f], %[pos], %[d]\n\t"
            "s_lshr_b64 %[T], %[T], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], %[d]\n\t"
            "s_bitset1_b64 %[fm], %[f]\n\t"
            "s_lshr_b64 %[T], %[T], 1\n\t"
            "v_readlane_b32 %[len], %[LAv], %[f]\n\t"
            "s_add_i32 %[rel], %[rel], %[d]\n\t"
            "s_add_i32 %[rel], %[rel], 1\n\t"
            "s_lshr_b64 %[L], %[L], %[len]\n\t"
            "s_add_i32 %[pos], %[f], %[len]\n\t"
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc0 s_hd%=\n\t"
            "s_ff1_i32_b64 %[d], %[c]\n\t"
            "s_add_i32 %[f], %[pos], %[d]\n\t"
            "s_lshr_b64 %[T], %[T], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], %[d]\n\t"
            "s_bitset1_b64 %[fm], %[f]\n\t"
            "s_lshr_b64 %[T], %[T], 1\n\t"
            "v_readlane_b32 %[len], %[LAv], %[f]\n\t"
            "s_add_i32 %[rel], %[rel], %[d]\n\t"
            "s_add_i32 %[rel], %[rel], 1\n\t"
            "s_lshr_b64 %[L], %[L], %[len]\n\t"
            "s_add_i32 %[pos], %[f], %[len]\n\t"
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc1 s_hl%=\n"
            "s_hd%=:\n\t"
            : [rel] "+s"(rel), [pos] "+s"(pos), [fm] "+s"(fm), [L] "+s"(Lr), [T] "+s"(Tr), [c] "=&s"(c64), [f] "=&s"(f),
              [d] "=&s"(dd), [len] "=&s"(len)
            : [LAv] "v"(LAv)
            : "scc");
    } else if (V == 2) {  // lean: no rel, T shifted by d+1 in one go (d+1 <= 63 assumed by the data), pos = f+len only
        asm volatile(
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc0 s_hd%=\n"
            "s_hl%=:\n\t"
            "s_ff1_i32_b64 %[d], %[c]\n\t"
            "s_add_i32 %[f], %[pos], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], %[d]\n\t"
            "s_add_i32 %[d], %[d], 1\n\t"
            "s_bitset1_b64 %[fm], %[f]\n\t"
            "s_lshr_b64 %[T], %[T], %[d]\n\t"
            "v_readlane_b32 %[len], %[LAv], %[f]\n\t"
            "s_nop 0\n\t"
            "s_lshr_b64 %[L], %[L], %[len]\n\t"
            "s_add_i32 %[pos], %[f], %[len]\n\t"
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc1 s_hl%=\n"
            "s_hd%=:\n\t"
            : [rel] "+s"(rel), [pos] "+s"(pos), [fm] "+s"(fm), [L] "+s"(Lr), [T] "+s"(Tr), [c] "=&s"(c64), [f] "=&s"(f),
              [d] "=&s"(dd), [len] "=&s"(len)
            : [LAv] "v"(LAv)
            : "scc");
    } else {  // V == 3: token length from a scalar table lookup instead of v_readlane: len = 5 + popcount trick is not
              // possible in general; use s_bfe on a packed nibble table held in 4 x 64-bit SGPRs?  Here: measure the
              // chain WITHOUT the readlane (constant len) as a lower bound of what a scalar-only hop could cost
        asm volatile(
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc0 s_hd%=\n"
            "s_hl%=:\n\t"
            "s_ff1_i32_b64 %[d], %[c]\n\t"
            "s_add_i32 %[f], %[pos], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], %[d]\n\t"
            "s_add_i32 %[d], %[d], 1\n\t"
            "s_bitset1_b64 %[fm], %[f]\n\t"
            "s_lshr_b64 %[T], %[T], %[d]\n\t"
            "s_lshr_b64 %[L], %[L], 6\n\t"
            "s_add_i32 %[pos], %[f], 6\n\t"
            "s_and_b64 %[c], %[L], %[T]\n\t"
            "s_cbranch_scc1 s_hl%=\n"
            "s_hd%=:\n\t"
            : [rel] "+s"(rel), [pos] "+s"(pos), [fm] "+s"(fm), [L] "+s"(Lr), [T] "+s"(Tr), [c] "=&s"(c64), [f] "=&s"(f),
              [d] "=&s"(dd), [len] "=&s"(len)
            : [LAv] "v"(LAv)
            : "scc");
    }
    fm_out = fm; pos_out = pos; rel_out = rel;
}

template <int V>
__global__ void k_hop(const uint64_t *Lw, const uint64_t *Tw, int nwin, int reps, int prio, uint64_t *out) {
    __shared__ uint32_t flag;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    if (wave == 0) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        uint64_t hops = 0, acc = 0;
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; r++)
            for (int w = 0; w < nwin; w++) {
                const uint64_t L0 = Lw[w], T0 = Tw[w];
                // token length if a fired entry started at this lane's bit (as decode.hip)
                const uint64_t bb = lane ? (L0 >> lane) : L0;
                uint32_t pl = (uint32_t)(bb >> 1) & 0xFFu, ns = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) { uint32_t s = pl & 1u; pl >>= 1 + s; ns += s; }
                const uint32_t LAv = 5 + ns;
                uint64_t fm; uint32_t pos, rel;
                walk<V>(__builtin_amdgcn_readfirstlane((uint32_t)L0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(L0 >> 32)) << 32),
                        __builtin_amdgcn_readfirstlane((uint32_t)T0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(T0 >> 32)) << 32),
                        LAv, fm, pos, rel);
                hops += __popcll(fm);
                acc += pos + rel;
            }
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = hops; out[blockIdx.x * 4 + 2] = acc; }
        __hip_atomic_store(&flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        // neighbours: spin like idle decoder workers
        while (__hip_atomic_load(&flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    }
}

template <int V>
static void run(const char *name, const uint64_t *dL, const uint64_t *dT, int nwin, int threads, int blocks, int prio, uint64_t *dout) {
    const int reps = 200;
    hipLaunchKernelGGL(k_hop<V>, dim3(blocks), dim3(threads), 0, 0, dL, dT, nwin, reps, prio, dout);
    CHK(hipDeviceSynchronize());
    uint64_t h[4];
    CHK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-34s threads=%4d blocks=%4d prio=%d: %8.1f cycles/hop  (%.1f hops/window, %.0f cycles/window)\n", name, threads, blocks,
           prio, (double)h[0] / (double)h[1], (double)h[1] / reps / nwin, (double)h[0] / reps / nwin);
}

int main() {
    const int nwin = 256;
    uint64_t *dL, *dT, *dout;
    CHK(hipMalloc(&dL, nwin * 8)); CHK(hipMalloc(&dT, nwin * 8)); CHK(hipMalloc(&dout, 4096 * 32));
    // two densities of fired entries: the slope between them is the cost of one hop, the rest is per-window overhead
    for (int dens : {10, 35, 60, 90}) {
        uint64_t hL[nwin], hT[nwin];
        srand(1);
        for (int w = 0; w < nwin; w++) {
            uint64_t l = 0, t = 0;
            for (int b = 0; b < 64; b++) {
                if (rand() % 100 < 45) l |= 1ull << b;
                if (rand() % 100 < dens) t |= 1ull << b;
            }
            hL[w] = l; hT[w] = t;
        }
        CHK(hipMemcpy(dL, hL, sizeof(hL), hipMemcpyHostToDevice));
        CHK(hipMemcpy(dT, hT, sizeof(hT), hipMemcpyHostToDevice));
        printf("-- type-A density %d %%\n", dens);
        run<0>("V0 current", dL, dT, nwin, 512, 256, 1, dout);
        run<1>("V1 two hops per trip", dL, dT, nwin, 512, 256, 1, dout);
        run<2>("V2 lean (no rel, T>>d+1)", dL, dT, nwin, 512, 256, 1, dout);
        run<3>("V3 no readlane (lower bound)", dL, dT, nwin, 512, 256, 1, dout);
    }
    return 0;
}
