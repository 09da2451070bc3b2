// gemm_i8_dense.hip — launch of k_gemm_i8_dense (gemm_i8_dense.h): matmul_int8 for large aligned problems.  Own translation
// unit (the kernel exists in one copy of its 128-slot loop per wave and output type; it compiles for minutes).
#include "gemm_i8_dense.h"

namespace mbnb {

bool gemm_i8_dense_shape(int64_t M, int64_t N, int64_t K) {
    return (K % 128 == 0) && (N % 16 == 0) && 256 * K < ((int64_t)1 << 31) && K * N < ((int64_t)1 << 31) &&
           ((M + 255) / 256) * ((N + 255) / 256) >= 96;
}

int launch_gemm_i8_dense(const int8_t *A, const int8_t *B, const float *sA, const float *sB, int64_t M, int64_t N, int64_t K,
                         int out_dtype, void *out, hipStream_t st) {
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
#define MBNB_I8D(OT)                                                                                                   \
    do {                                                                                                               \
        auto kern = k_gemm_i8_dense<OT>;                                                                               \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_int8(dense)")) return rc;    \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, A, B, sA, sB, static_cast<OT *>(out), M, N, K); \
    } while (0)
    switch (out_dtype) {
        case MBNB_F16: MBNB_I8D(f16_t); break;
        case MBNB_BF16: MBNB_I8D(bf16_t); break;
        default: MBNB_I8D(float); break;
    }
#undef MBNB_I8D
    set_kernel_name("i8_dense");
    return check_launch("matmul_int8(dense)");
}

}  // namespace mbnb
