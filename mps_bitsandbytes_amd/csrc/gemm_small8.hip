// gemm_small8.hip — launch of k_gemm_small8 (gemm_small8.h): W8A16 (int8 / FP8 weights) for 16/32 < M <= 256 rows.
#include "gemm_small8.h"
#include "gemm_mid.h"

namespace mbnb {

bool gemm_small8_shape(int64_t M, int64_t N, int64_t K);
int64_t gemm_small8_slices(int64_t M, int64_t N, int64_t K);

// Returns MBNB_NOT_APPLICABLE when the kernel cannot serve the call.
template <typename T, int WF>
int launch_gemm_small8(const T *x, const uint8_t *W, const float *scales, const T *bias, T *out, int64_t M, int64_t N, int64_t K, float *ws,
                       int64_t ws_bytes, hipStream_t st) {
    if (!gemm_small8_shape(M, N, K)) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) return MBNB_NOT_APPLICABLE;
    int64_t slices = gemm_small8_slices(M, N, K);
    const int64_t steps = K / 256;
    if (slices > 1 && (ws == nullptr || ws_bytes < slices * M * N * 4 || (reinterpret_cast<uintptr_t>(ws) & 15))) {
        if (steps > 8) return MBNB_NOT_APPLICABLE;
        slices = 1;
    }
    const int64_t kps = ((steps + slices - 1) / slices) * 256;
    const int64_t used = (K + kps - 1) / kps;
#define MBNB_S8(MF)                                                                                                          \
    do {                                                                                                                     \
        auto kern = k_gemm_small8<T, WF, MF>;                                                                                \
        constexpr int lds = gemm_small_lds_bytes<MF>();                                                                      \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "linear_int8(small)")) return rc;             \
        const dim3 grid((unsigned)((N + 63) / 64), (unsigned)used, (unsigned)((M + 16 * MF - 1) / (16 * MF)));               \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, W, scales, bias, out, used > 1 ? ws : nullptr, M, N, K, kps);  \
    } while (0)
    if (M <= 64) MBNB_S8(4);
    else MBNB_S8(8);
#undef MBNB_S8
    if (used <= 1) {
        set_kernel_name(WF == W8_INT8 ? "w8a16_small" : "fp8a16_small");
        return check_launch("linear_int8(small)");
    }
    if (int rc = check_launch("linear_int8(small split-K)")) return rc;
    const int64_t groups = M * ((N + 3) / 4);
    hipLaunchKernelGGL((k_splitk_reduce_rm<T, T>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, ws, (int)used, bias, out, M, N);
    set_kernel_name(WF == W8_INT8 ? "w8a16_small_splitk" : "fp8a16_small_splitk");
    return check_launch("linear_int8(small split-K reduce)");
}

#define MBNB_INST(T, WF) \
    template int launch_gemm_small8<T, WF>(const T *, const uint8_t *, const float *, const T *, T *, int64_t, int64_t, int64_t, float *, int64_t, hipStream_t);
MBNB_INST(f16_t, W8_INT8) MBNB_INST(f16_t, W8_FP8) MBNB_INST(bf16_t, W8_INT8) MBNB_INST(bf16_t, W8_FP8)
#undef MBNB_INST

}  // namespace mbnb
