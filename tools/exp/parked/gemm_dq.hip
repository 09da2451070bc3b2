// gemm_dq.hip — launch of k_gemm_dq (gemm_dq.h): large-M matmul_4bit in one launch, the weight decoded once per launch by the
// GEMM's own workgroups (hand-off through agent-scope flags).  Own translation unit.
#include <mutex>
#include "gemm_dq.h"

namespace mbnb {

// compute units of the current device (every workgroup of k_gemm_dq must be resident at once): queried once per device
static int device_cus() {
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = -1;
        cus[dev] = v > 0 ? v : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

bool gemm_dq_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight, int blocksize) {
    if (blocksize != 64 || K_weight != K || K % (64 * GQ_SLAB) != 0 || K / (64 * GQ_SLAB) > GQ_MAX_SLABS || K / (64 * GQ_SLAB) < GQ_AHEAD + 1) return false;
    if ((M + 255) / 256 != 16) return false;                    // 16 producers per tile column, 16 weight rows each
    if (N * K * 2 >= ((int64_t)1 << 31) || 256 * K * 2 >= ((int64_t)1 << 31)) return false;
    return true;
}
int64_t gemm_dq_sync_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight, int blocksize) {
    if (!gemm_dq_shape(M, N, K, K_weight, blocksize)) return 0;
    return (gq_sync_bytes((N + 255) / 256) + 255) & ~(int64_t)255;
}

// Returns MBNB_NOT_APPLICABLE when the kernel does not serve the call, otherwise the launch status.  ws: the Wd scratch
// (N * K * 2 bytes, 256-byte aligned); sync: gemm_dq_sync_bytes() bytes, ZERO on entry (zero again when the launch has finished).
int matmul_4bit_dq_path(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N, int64_t K_weight,
                        int blocksize, int qt, int w_dtype, const void *bias, int out_dtype, void *out, void *ws, int64_t ws_bytes,
                        void *sync, int64_t sync_bytes, hipStream_t st) {
    if (w_dtype != MBNB_F16 && w_dtype != MBNB_BF16) return MBNB_NOT_APPLICABLE;
    if (am.i8 != nullptr || ws == nullptr || sync == nullptr) return MBNB_NOT_APPLICABLE;
    if (!gemm_dq_shape(M, N, K, K_weight, blocksize)) return MBNB_NOT_APPLICABLE;
    if (ws_bytes < N * K * 2 || sync_bytes < gemm_dq_sync_bytes(M, N, K, K_weight, blocksize)) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(ws) & 255) || (reinterpret_cast<uintptr_t>(packed) & 15) ||
        (reinterpret_cast<uintptr_t>(am.f32) & 3) || (reinterpret_cast<uintptr_t>(sync) & 3))
        return MBNB_NOT_APPLICABLE;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    if (tiles > device_cus()) return MBNB_NOT_APPLICABLE;       // every workgroup resident at once (one per CU)
    int rc;
    if (w_dtype == MBNB_F16) {
        auto kern = k_gemm_dq<f16_t>;
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GQ_LDS, "matmul_4bit(dq)"))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GQ_LDS, st, static_cast<const f16_t *>(A), packed, am.f32, qt,
                           static_cast<f16_t *>(ws), static_cast<uint32_t *>(sync), static_cast<const f16_t *>(bias), out, out_dtype, M, N, K);
    } else {
        auto kern = k_gemm_dq<bf16_t>;
        if ((rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GQ_LDS, "matmul_4bit(dq)"))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GQ_LDS, st, static_cast<const bf16_t *>(A), packed, am.f32, qt,
                           static_cast<bf16_t *>(ws), static_cast<uint32_t *>(sync), static_cast<const bf16_t *>(bias), out, out_dtype, M, N, K);
    }
    set_kernel_name("dq_inlaunch");
    return check_launch("matmul_4bit(dq)");
}

}  // namespace mbnb
