// dq4_exp.hip — variants of the large-matrix dequantize_4bit kernel (diagnostic): libdq4_exp.so exports exp_dq4(variant, ...).
// Flat form (cols == cols_padded, cols % blocksize == 0, 16-bit outputs): thread t of a workgroup handles dwords t + 256 u (u < UN) of a
// contiguous run of 256 UN dwords -- every load instruction of a wave reads 256 contiguous bytes, every store instruction writes 1 KiB
// contiguous --, all loads of the UN groups issued before the first decode.  NT: nontemporal stores.
#include "../../mps_bitsandbytes_amd/csrc/common.h"
using namespace mbnb;
template <typename T, int QT, bool NESTED, int UN, int NT, bool XMAP = false>   // XMAP: workgroups of one XCD (blockIdx % 8) take a contiguous eighth; NT: 0 plain stores, 1 nontemporal, 2 write-through "sc0 sc1", 3 "sc1", 4 "sc0"
__global__ __launch_bounds__(256) void k_dq4_flat(const uint8_t *__restrict__ packed, AbsmaxView am, int64_t ndw, int bs_shift, T *__restrict__ out) {
    __shared__ float lut[16];
    const int tid = threadIdx.x;
    const int64_t per = (gridDim.x + 7) / 8;
    const int64_t bid = XMAP ? (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t base = bid * (256 * UN) + tid;
    uint32_t w[UN];
    float a[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t g = base + 256 * u;
        const bool ok = g < ndw;
        w[u] = ok ? reinterpret_cast<const uint32_t *>(packed)[g] : 0u;
        a[u] = ok ? load_absmax<NESTED>(am, (g * 8) >> bs_shift) : 0.0f;
    }
    fill_code_lut<QT>(lut, tid);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t g = base + 256 * u;
        if (g >= ndw) continue;
        u32x4 p;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v0 = lut[(w[u] >> (8 * j)) & 15] * a[u], v1 = lut[(w[u] >> (8 * j + 4)) & 15] * a[u];
            p[j] = pack2<T>(v0, v1);
        }
        u32x4 *o = reinterpret_cast<u32x4 *>(out) + g;
        if constexpr (NT == 1) __builtin_nontemporal_store(p, o);
        else if constexpr (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(p) : "memory");
        else if constexpr (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(p) : "memory");
        else if constexpr (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(o), "v"(p) : "memory");
        else *o = p;
    }
}
extern "C" int exp_dq4(int variant, const uint8_t *packed, const float *absmax, void *out, int64_t rows, int64_t cols, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    AbsmaxView am{absmax, nullptr, nullptr, 0};
    const int64_t ndw = rows * cols / 8;
#define RUN(UN, NT) hipLaunchKernelGGL((k_dq4_flat<bf16_t, MBNB_NF4, false, UN, NT>), dim3((unsigned)((ndw + 256 * UN - 1) / (256 * UN))), dim3(256), 0, st, packed, am, ndw, 6, static_cast<bf16_t *>(out))
    switch (variant) {
        case 1: RUN(1, 0); break;
        case 21: RUN(1, 2); break;
        case 22: RUN(1, 3); break;
        case 23: RUN(1, 4); break;
        case 54: hipLaunchKernelGGL((k_dq4_flat<bf16_t, MBNB_NF4, false, 4, 3, true>), dim3((unsigned)(((ndw + 1023) / 1024 + 7) / 8 * 8)), dim3(256), 0, st, packed, am, ndw, 6, static_cast<bf16_t *>(out)); break;
        case 51: hipLaunchKernelGGL((k_dq4_flat<bf16_t, MBNB_NF4, false, 1, 3, true>), dim3((unsigned)(((ndw + 255) / 256 + 7) / 8 * 8)), dim3(256), 0, st, packed, am, ndw, 6, static_cast<bf16_t *>(out)); break;
        case 33: RUN(3, 3); break;
        case 35: RUN(5, 3); break;
        case 32: RUN(2, 3); break;
        case 34: RUN(4, 3); break;
        case 38: RUN(8, 3); break;
        case 36: RUN(16, 3); break;
        case 2: RUN(2, 0); break;
        case 4: RUN(4, 0); break;
        case 8: RUN(8, 0); break;
        case 14: RUN(4, 1); break;
        case 18: RUN(8, 1); break;
        case 12: RUN(2, 1); break;
        default: return -1;
    }
#undef RUN
    return (int)hipGetLastError();
}
