// gemm_i8_inplace.hip — launch of k_gemm_i8_inplace (gemm_i8_inplace.h): matmul_int8 for large aligned problems on the four-wave
// pipeline with B read where it lies ([K, N], transposing LDS reads): no transpose pass, no workspace.
#include "gemm_i8_inplace.h"

namespace mbnb {

bool gemm_i8_inplace_shape(const int8_t *A, const int8_t *B, int64_t M, int64_t N, int64_t K) {
    return (K % 128 == 0) && K >= 256 && (N % 16 == 0) && N >= 256 && ((M + 255) / 256) * ((N + 255) / 256) >= 96 && 256 * K < ((int64_t)1 << 31) &&
           K * N < ((int64_t)1 << 31) - (1 << 17) &&   /* the zero-fill offset of gemm_i8_inplace.h (0x7FFF0000 - 3072) lies beyond num_records */ ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
}

int launch_gemm_i8_inplace(const int8_t *A, const int8_t *B, const float *sA, const float *sB, int64_t M, int64_t N, int64_t K,
                           int out_dtype, void *out, hipStream_t st) {
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
#define MBNB_I8IP(OT)                                                                                                   \
    do {                                                                                                                \
        auto kern = k_gemm_i8_inplace<OT>;                                                                              \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_int8(in place)")) return rc;  \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, A, B, sA, sB, static_cast<OT *>(out), M, N, K); \
    } while (0)
    switch (out_dtype) {
        case MBNB_F16: MBNB_I8IP(f16_t); break;
        case MBNB_BF16: MBNB_I8IP(bf16_t); break;
        default: MBNB_I8IP(float); break;
    }
#undef MBNB_I8IP
    set_kernel_name("i8_inplace4");
    return check_launch("matmul_int8(in place)");
}

}  // namespace mbnb
