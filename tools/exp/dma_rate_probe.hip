// dma_rate_probe.hip — how many bytes per second does ONE CU take in from L2 / HBM, by path?
//   mode 0: LDS-DMA (global_load_lds_dwordx4), 1: global_load_dwordx4 -> VGPR -> ds_write_b128, 2: global_load_dwordx4 -> VGPR only
// 256 workgroups x 512 threads (one per CU); every wave keeps `depth` 1-KiB pieces in flight and moves `pieces` pieces per
// iteration; the source is a window of `window` bytes (L2-resident when small and shared by all workgroups, HBM when
// large and private).      hipcc -O3 --offload-arch=gfx950 -o dma_rate_probe dma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void k(const char *src, int64_t window, int64_t wg_stride, int iters, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 8 waves x DEPTH x 1 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *base = src + (int64_t)blockIdx.x * wg_stride;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH];
    int64_t off = (int64_t)wave * 1024 + lane * 16;
    // prologue: DEPTH pieces in flight
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const char *g = base + (off % window);
        if constexpr (MODE == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
        else r[d] = *reinterpret_cast<const u32x4 *>(g);
        off += 8192;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            // retire the oldest piece, issue a new one into its slot
            if constexpr (MODE == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
                acc[0] ^= *reinterpret_cast<const uint32_t *>(smem + (wave * DEPTH + d) * 1024 + lane * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (MODE == 1) {
                *reinterpret_cast<u32x4 *>(smem + (wave * DEPTH + d) * 1024 + lane * 16) = r[d];
            } else {
                acc ^= r[d];
            }
            const char *g = base + (off % window);
            if constexpr (MODE == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
            else r[d] = *reinterpret_cast<const u32x4 *>(g);
            off += 8192;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == 1) acc[0] ^= *reinterpret_cast<const uint32_t *>(smem + threadIdx.x * 4);
    if constexpr (MODE != 0) for (int d = 0; d < DEPTH; d++) acc ^= r[d];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH> double run(const char *src, int64_t window, int64_t wg_stride, int iters, uint32_t *sink) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, src, window, wg_stride, iters, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, src, window, wg_stride, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * 256 * 8 * (double)(iters + 1) * DEPTH * 1024;
    return bytes / (ms * 1e-3) / 1e9 / 256;   // GB/s per CU
}

int main() {
    char *src; uint32_t *sink;
    const int64_t total = 1ll << 30;
    hipMalloc(&src, total); hipMalloc(&sink, 4);
    hipMemset(src, 1, total);
    struct Cfg { const char *name; int64_t window, stride; } cfgs[] = {
        {"shared 2 MiB window (L2 hits, every CU the same lines)", 2ll << 20, 0},
        {"private 128 KiB window per CU (L2 hits, distinct lines)", 128ll << 10, 128ll << 10},
        {"private 4 MiB per CU = 1 GiB swept (HBM)", 4ll << 20, 4ll << 20}};
    for (auto &c : cfgs) {
        const int iters = 400;
        printf("%s\n", c.name);
        printf("  LDS-DMA         depth 2: %6.1f  depth 4: %6.1f  depth 8: %6.1f  depth 12: %6.1f GB/s per CU\n",
               run<0, 2>(src, c.window, c.stride, iters, sink), run<0, 4>(src, c.window, c.stride, iters, sink),
               run<0, 8>(src, c.window, c.stride, iters, sink), run<0, 12>(src, c.window, c.stride, iters, sink));
        printf("  load+ds_write   depth 2: %6.1f  depth 4: %6.1f  depth 8: %6.1f GB/s per CU\n",
               run<1, 2>(src, c.window, c.stride, iters, sink), run<1, 4>(src, c.window, c.stride, iters, sink),
               run<1, 8>(src, c.window, c.stride, iters, sink));
        printf("  load to VGPR    depth 2: %6.1f  depth 4: %6.1f  depth 8: %6.1f GB/s per CU\n",
               run<2, 2>(src, c.window, c.stride, iters, sink), run<2, 4>(src, c.window, c.stride, iters, sink),
               run<2, 8>(src, c.window, c.stride, iters, sink));
    }
    return 0;
}
