// small_stamps.hip — where does a step of k_gemm_small go?  Built with GS_STAMPS: workgroup (7, 0, 0)'s thread GS_STAMP_TID stamps every step:
// before its vmcnt(0), after it, after the barrier, after the 64 MFMAs.  exp_small_stamps runs k_gemm_small<bf16, plain, MF 8, NF 1, 16 steps> on one
// slice and copies the stamps out.
#define GS_STAMPS 1
#include "../../mps_bitsandbytes_amd/csrc/gemm_small.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
extern "C" int exp_small_stamps_nf2(const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K, unsigned long long *host_stamps, void *stream) {
    auto kern = k_gemm_small<bf16_t, false, 8, 2, 8>;
    constexpr int lds = gemm_small_lds_bytes<8>();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    AbsmaxView am{absmax, nullptr, nullptr, 0};
    const dim3 grid((unsigned)((N + 127) / 128), 1u, (unsigned)((M + 127) / 128));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), packed, am, static_cast<const bf16_t *>(nullptr), out,
                       (int)MBNB_BF16, static_cast<float *>(nullptr), M, N, K, K, K, (int)MBNB_NF4, 6);
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return -3;
    return (int)hipMemcpyFromSymbol(host_stamps, HIP_SYMBOL(g_gs_stamps), sizeof(unsigned long long) * 256);
}
extern "C" int exp_small_stamps(const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K, unsigned long long *host_stamps, void *stream) {
    auto kern = k_gemm_small<bf16_t, false, 8, 1, 16>;
    constexpr int lds = gemm_small_lds_bytes<8>();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    AbsmaxView am{absmax, nullptr, nullptr, 0};
    const dim3 grid((unsigned)((N + 63) / 64), 1u, (unsigned)((M + 127) / 128));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), packed, am, static_cast<const bf16_t *>(nullptr), out,
                       (int)MBNB_BF16, static_cast<float *>(nullptr), M, N, K, K, K, (int)MBNB_NF4, 6);
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return -3;
    return (int)hipMemcpyFromSymbol(host_stamps, HIP_SYMBOL(g_gs_stamps), sizeof(unsigned long long) * 256);
}
