// Test infrastructure (tests/test_gpu_spiht.py): occupies the GPU with workgroups that do nothing for a given time, from a
// stream of its own -- the "other process's kernel" the several-CUs-per-image encoder cannot see coming.  Built with hipcc
// by the test that uses it; nothing of the product links it.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_fill(uint64_t ticks, uint32_t lds_words, uint32_t *sink) {
    extern __shared__ uint32_t dyn[];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t guard = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks && ++guard < (1u << 26)) __builtin_amdgcn_s_sleep(32);
    if (lds_words && dyn[threadIdx.x % lds_words] == 0x12345u && sink) *sink = 1;  // (keeps the LDS allocation alive)
}

static hipStream_t g_st = nullptr;

extern "C" int filler_launch(int blocks, int threads, uint32_t lds_bytes, uint64_t ticks) {
    if (!g_st && hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking) != hipSuccess) return -1;
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute((const void *)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(threads), lds_bytes, g_st, ticks, lds_bytes / 4, (uint32_t *)nullptr);
    return (int)hipGetLastError();
}
extern "C" int filler_wait(void) { return g_st ? (int)hipStreamSynchronize(g_st) : 0; }
extern "C" int filler_num_cu(void) {
    hipDeviceProp_t p;
    return hipGetDeviceProperties(&p, 0) == hipSuccess ? p.multiProcessorCount : -1;
}
