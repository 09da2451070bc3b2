// m0_exp.hip — A/B harness for GD_M0_GROUP (common.h) in the kernels other than k_gemm_dense (dense_exp.hip): built twice into
// libm0_exp0.so / libm0_exp1.so.  exp_d128 (k_gemm_dense128, bf16), exp_i8 (k_gemm_i8_inplace, bf16 out), exp_small (k_gemm_small<bf16, plain,
// MF 8, NF 1, 16 steps>, one slice, no partials).
#include "../../mps_bitsandbytes_amd/csrc/gemm_dense128.h"
#include "../../mps_bitsandbytes_amd/csrc/gemm_i8_inplace.h"
#include "../../mps_bitsandbytes_amd/csrc/gemm_small.h"
namespace mbnb {
void set_error(const char *, ...) {}
void set_kernel_name(const char *) {}
int check_launch(const char *) { return (int)hipGetLastError(); }
int ensure_dyn_lds(const void *f, int bytes, const char *) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
}  // namespace mbnb
using namespace mbnb;
static int lds_once(const void *f, int bytes) { return (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
extern "C" int exp_d128(const void *X, const void *Wd, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_dense128<bf16_t>;
    if (lds_once(reinterpret_cast<const void *>(kern), G128_LDS)) return -2;
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), G128_LDS, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), static_cast<const bf16_t *>(Wd),
                       static_cast<const bf16_t *>(nullptr), out, (int)MBNB_BF16, M, N, K, K);
    return (int)hipGetLastError();
}
extern "C" int exp_i8(const int8_t *A, const int8_t *B, const float *sA, const float *sB, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_i8_inplace<bf16_t>;
    if (lds_once(reinterpret_cast<const void *>(kern), GD_LDS)) return -2;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, static_cast<hipStream_t>(stream), A, B, sA, sB, static_cast<bf16_t *>(out), M, N, K);
    return (int)hipGetLastError();
}
extern "C" int exp_small(const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_small<bf16_t, false, 8, 1, 16>;
    constexpr int lds = gemm_small_lds_bytes<8>();
    if (lds_once(reinterpret_cast<const void *>(kern), lds)) return -2;
    AbsmaxView am{absmax, nullptr, nullptr, 0};
    const dim3 grid((unsigned)((N + 63) / 64), 1u, (unsigned)((M + 127) / 128));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), packed, am, static_cast<const bf16_t *>(nullptr), out,
                       (int)MBNB_BF16, static_cast<float *>(nullptr), M, N, K, K, K, (int)MBNB_NF4, 6);
    return (int)hipGetLastError();
}

// exp_small_v: k_gemm_small<bf16, plain, MF, NF, 16> with one K slice (K <= 4096), variant 0..3 = (MF, NF) (8, 1), (4, 1), (2, 1), (8, 2)
template <int MF, int NF>
static int small_v(const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K, void *stream) {
    auto kern = k_gemm_small<bf16_t, false, MF, NF, 16>;
    constexpr int lds = gemm_small_lds_bytes<MF>();
    if (lds_once(reinterpret_cast<const void *>(kern), lds)) return -2;
    AbsmaxView am{absmax, nullptr, nullptr, 0};
    const dim3 grid((unsigned)((N + 64 * NF - 1) / (64 * NF)), 1u, (unsigned)((M + 16 * MF - 1) / (16 * MF)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(X), packed, am, static_cast<const bf16_t *>(nullptr), out,
                       (int)MBNB_BF16, static_cast<float *>(nullptr), M, N, K, K, K, (int)MBNB_NF4, 6);
    return (int)hipGetLastError();
}
extern "C" int exp_small_v(const void *X, const uint8_t *packed, const float *absmax, void *out, int64_t M, int64_t N, int64_t K, void *stream, int variant) {
    if (K > 4096 || K % 256) return -3;
    switch (variant) {
        case 0: return small_v<8, 1>(X, packed, absmax, out, M, N, K, stream);
        case 1: return small_v<4, 1>(X, packed, absmax, out, M, N, K, stream);
        case 2: return small_v<2, 1>(X, packed, absmax, out, M, N, K, stream);
        case 3: return small_v<8, 2>(X, packed, absmax, out, M, N, K, stream);
    }
    return -4;
}
