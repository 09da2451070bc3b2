// cores_exp.hip — probe: can a SMALL helper kernel (<= 32 VGPRs, one wave per SIMD) run on the same SIMDs as k_gemm_dense (476 of
// the 512 unified registers, 128 KiB of LDS) when the two are launched on two streams?  The helper stamps its start / end
// (s_memrealtime, 100 MHz), its hardware id and how much of its loop it got through; the harness (ab_cores.py) launches it beside
// the product GEMM and reads the stamps.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// `work`: 0 = dependent VALU chain; 1 = streams `bytes_per_wg` bytes from src to dst (16 B per lane and instruction, write-through)
template <int REGS>
__global__ __launch_bounds__(256) void k_helper(int iters, int work, const u32x4 *__restrict__ src, u32x4 *__restrict__ dst,
                                                int64_t vec_per_wg, uint64_t *__restrict__ stamps) {
    const uint64_t t0 = __builtin_readcyclecounter();
    uint64_t r0 = wall_clock64();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (REGS > 32) asm volatile("v_mov_b32 v60, 0" ::: "v60");
    float a = (float)threadIdx.x;
    if (work == 0) {
        for (int i = 0; i < iters; i++) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(a));
    } else {
        const u32x4 *s = src + (int64_t)blockIdx.x * vec_per_wg;
        u32x4 *d = dst + (int64_t)blockIdx.x * vec_per_wg;
        for (int64_t i = threadIdx.x; i < vec_per_wg; i += 256) {
            u32x4 v = s[i];
            v.x += 1;
            __builtin_nontemporal_store(v, d + i);
        }
    }
    uint64_t r1 = wall_clock64();
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint64_t *o = stamps + 6 * blockIdx.x;
        o[0] = r0;
        o[1] = r1;
        o[2] = hw;
        o[3] = xcc;
        o[4] = t1 - t0;
        o[5] = (uint64_t)__float_as_uint(a);
    }
}

extern "C" int cores_helper(int regs, int grid, int iters, int work, const void *src, void *dst, int64_t vec_per_wg, void *stamps, void *stream) {
    auto st = static_cast<hipStream_t>(stream);
    if (regs <= 32)
        hipLaunchKernelGGL((k_helper<32>), dim3(grid), dim3(256), 0, st, iters, work, (const u32x4 *)src, (u32x4 *)dst, vec_per_wg, (uint64_t *)stamps);
    else
        hipLaunchKernelGGL((k_helper<64>), dim3(grid), dim3(256), 0, st, iters, work, (const u32x4 *)src, (u32x4 *)dst, vec_per_wg, (uint64_t *)stamps);
    return (int)hipGetLastError();
}
