// gemm_fused4.hip — launch of k_gemm_fused4 (gemm_fused4.h): the fused 4-bit decode + MFMA GEMM on the four-wave pipeline, for
// large M at blocksize 64.  Own translation unit (the kernel compiles for minutes).
#include "gemm_fused4.h"

namespace mbnb {

bool gemm_fused4_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight, int blocksize) {
    if (blocksize != 64 || K % 64 != 0 || K < 128 || K_weight % 256 != 0) return false;
    if (256 * K * 2 >= ((int64_t)1 << 31) || 256 * (K_weight / 2) >= ((int64_t)1 << 31)) return false;
    return ((M + 255) / 256) * ((N + 255) / 256) >= 96;
}

template <typename T, bool NESTED>
static int launch_fused4(const T *x, const typename Q4ProducerRT<T, NESTED>::Params &wp, const T *bias, void *out, int out_dtype,
                         int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kern = k_gemm_fused4<T, NESTED>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GF_LDS, "matmul_4bit(fused4)")) return rc;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GF_LDS, st, x, wp, bias, out, out_dtype, M, N, K);
    return check_launch("matmul_4bit(fused4)");
}

// Returns MBNB_NOT_APPLICABLE when the kernel does not serve the call, otherwise the launch status.
int matmul_4bit_fused4_path(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                            int64_t K_weight, int blocksize, int qt, int w_dtype, const void *bias, int out_dtype, void *out,
                            hipStream_t st) {
    if (w_dtype != MBNB_F16 && w_dtype != MBNB_BF16) return MBNB_NOT_APPLICABLE;
    if (!gemm_fused4_shape(M, N, K, K_weight, blocksize)) return MBNB_NOT_APPLICABLE;
    const bool nested = am.i8 != nullptr;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15)) return MBNB_NOT_APPLICABLE;
    int bs2_shift = 0;
    if (nested) {
        if (am.bs2 < 4 || (am.bs2 & (am.bs2 - 1)) != 0 || (reinterpret_cast<uintptr_t>(am.i8) & 3) || (reinterpret_cast<uintptr_t>(am.am2) & 3))
            return MBNB_NOT_APPLICABLE;
        while ((1 << bs2_shift) < am.bs2) bs2_shift++;
    } else if (reinterpret_cast<uintptr_t>(am.f32) & 15) {
        return MBNB_NOT_APPLICABLE;
    }
    int rc;
#define MBNB_F4(T, NESTED)                                                                                                    \
    do {                                                                                                                      \
        typename Q4ProducerRT<T, NESTED>::Params wp{packed, am, N, K_weight, K_weight / 64, 6, qt, bs2_shift, 8, 6};          \
        rc = launch_fused4<T, NESTED>(static_cast<const T *>(A), wp, static_cast<const T *>(bias), out, out_dtype, M, N, K, st); \
    } while (0)
    if (w_dtype == MBNB_F16) {
        if (nested) MBNB_F4(f16_t, true);
        else MBNB_F4(f16_t, false);
    } else {
        if (nested) MBNB_F4(bf16_t, true);
        else MBNB_F4(bf16_t, false);
    }
#undef MBNB_F4
    set_kernel_name("mfma256f");
    return rc;
}

}  // namespace mbnb
