// gemm_mid.hip — launcher of k_gemm_mid (gemm_mid.h): the mid-sized-batch fused 4-bit GEMM, with its split-K policy.
#include "gemm_mid.h"

namespace mbnb {

// Which (M, N, K) the mid kernel serves (shape part of the policy; the dispatcher adds the layout conditions): more rows
// than the weight-streaming kernels take, fewer than 96 output tiles of 256 x 256, and at most 192 rows (384 for wide
// layers, N >= 8192) -- beyond that the 128 x 128 split-K path is as fast or faster (tools/exp/ab_mid.py,
// profiles/r02_mid_sweep.txt: 4096^2 M = 256: 29.8 vs 28.3 us; 11008 x 4096 M = 256: 60.5 vs 70.0 us).
bool gemm_mid_shape(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const int64_t m_max = N >= 8192 ? 384 : 192;
    return M > 32 && M <= m_max && tiles256 < 96 && K % 64 == 0 && K >= 256;
}
// K slices: the fewest that give every CU a workgroup, each a multiple of 256 k (absmax-by-4 blocks); the f32 partials
// are written and read once each, so never more than 8 (measured: 16 slices of 4 k-steps lose to 4 of 16 at M = 128).
int64_t gemm_mid_slices(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 63) / 64);
    if (tiles >= 256) return 1;
    int64_t s = (256 + tiles - 1) / tiles;
    const int64_t smax = K / 256;
    if (s > smax) s = smax;
    if (s > 8) s = 8;
    return s < 2 ? 1 : s;
}
int64_t gemm_mid_k_per_slice(int64_t K, int64_t slices) {
    const int64_t blocks = (K + 255) / 256;
    return ((blocks + slices - 1) / slices) * 256;
}
int64_t gemm_mid_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (!gemm_mid_shape(M, N, K)) return 0;
    const int64_t s = gemm_mid_slices(M, N, K);
    return s > 1 ? s * M * N * 4 : 0;
}

template <typename T, typename OutT, bool NESTED, int ABL = 0>
int launch_gemm_mid(const T *x, const typename Q4ProducerRT<T, NESTED>::Params &wp, const T *bias, OutT *out, int64_t M,
                    int64_t N, int64_t K, float *ws, int64_t ws_bytes, int force_slices, hipStream_t st) {
    auto kern = k_gemm_mid<T, NESTED, ABL>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), MID_LDS, "matmul_4bit(mid)")) return rc;
    int64_t slices = force_slices > 0 ? force_slices : gemm_mid_slices(M, N, K);
    if (slices > 1 && (ws == nullptr || ws_bytes < slices * M * N * 4 || (reinterpret_cast<uintptr_t>(ws) & 15))) slices = 1;
    int64_t kps = gemm_mid_k_per_slice(K, slices);
    slices = (K + kps - 1) / kps;    // no empty slice
    const int64_t tiles = ((M + 127) / 128) * ((N + 63) / 64);
    const int od = sizeof(OutT) == 4 ? MBNB_F32 : (std::is_same<OutT, f16_t>::value ? MBNB_F16 : MBNB_BF16);
    if (slices <= 1) {
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), MID_LDS, st, x, wp, bias, static_cast<void *>(out), od,
                           static_cast<float *>(nullptr), M, N, K, 1, K);
        set_kernel_name("mfma_mid");
        return check_launch("matmul_4bit(mid)");
    }
    set_kernel_name("mfma_mid_splitk");
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * slices)), dim3(512), MID_LDS, st, x, wp, bias, static_cast<void *>(out), od, ws,
                       M, N, K, (int)slices, kps);
    if (int rc = check_launch("matmul_4bit(mid split-K)")) return rc;
    const int64_t groups = M * ((N + 3) / 4);
    hipLaunchKernelGGL((k_splitk_reduce_rm<T, OutT>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, ws, (int)slices, bias,
                       out, M, N);
    return check_launch("matmul_4bit(mid reduce)");
}

#define MBNB_INST(T, OutT, NESTED)                                                                                         \
    template int launch_gemm_mid<T, OutT, NESTED>(const T *, const typename Q4ProducerRT<T, NESTED>::Params &, const T *, \
                                                  OutT *, int64_t, int64_t, int64_t, float *, int64_t, int, hipStream_t);
#define MBNB_INST3(T, NESTED) MBNB_INST(T, f16_t, NESTED) MBNB_INST(T, bf16_t, NESTED) MBNB_INST(T, float, NESTED)
MBNB_INST3(f16_t, false)
MBNB_INST3(f16_t, true)
MBNB_INST3(bf16_t, false)
MBNB_INST3(bf16_t, true)
#undef MBNB_INST3
#undef MBNB_INST

}  // namespace mbnb
