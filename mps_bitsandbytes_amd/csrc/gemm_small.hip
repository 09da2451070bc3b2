// gemm_small.hip — launch of k_gemm_small (gemm_small.h): fused 4-bit GEMM for 64 < M <= 256 rows, blocksize 64.
#include "gemm_small.h"
#include "gemm_mid.h"

namespace mbnb {

bool gemm_small_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    // from 33 rows (17 for layers of more than 16 Mi weights) the weight-streaming skinny kernel loses: 64 x 4096 x 4096 12.8 us
    // here, 18.5 there; 32 x 4096 x 4096 11.8 vs 11.7; 32 x 11008 x 4096 21.7 vs 31.0 (tools/exp/small_check.py)
    return (M > 32 || (M > 16 && N * K > ((int64_t)1 << 24))) && M <= 256 && K % 256 == 0 && K >= 512 && K <= 16 * 2048 && K_weight % 256 == 0 && N >= 64 &&
           256 * K * 2 < ((int64_t)1 << 31);
}
// K slices: the count that minimises  rounds on 256 CUs x k-steps per slice  (+ half a k-step per extra slice for its partials)
int64_t gemm_small_slices(int64_t M, int64_t N, int64_t K) {
    const int64_t mf = M > 64 ? 8 : 4;
    const int64_t wgs = ((N + 63) / 64) * ((M + 16 * mf - 1) / (16 * mf));
    const int64_t steps = K / 256;
    int64_t best = 1;
    double best_t = 1e30;
    for (int64_t s = 1; s <= 16 && s <= steps; s++) {
        const int64_t per = (steps + s - 1) / s;
        if (per > 8) continue;            // a slice's weights live in registers: at most 8 steps of 256 k
        if (per < 2 && s > 1) break;
        const double t = (double)((wgs * s + 255) / 256) * (double)per + 0.5 * (double)(s - 1);
        if (t < best_t - 1e-9) {
            best_t = t;
            best = s;
        }
    }
    return best;
}
int64_t gemm_small_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (!gemm_small_shape(M, N, K, K_weight)) return 0;
    const int64_t s = gemm_small_slices(M, N, K);
    return s > 1 ? s * M * N * 4 : 0;
}

// Returns 1 when the kernel cannot serve the call (K longer than one slice of 8 steps and no workspace for the partials).
template <typename T, typename OutT, bool NESTED, int MF>
static int launch_gemm_small_mf(const T *x, const uint8_t *packed, const AbsmaxView &am, const T *bias, OutT *out, int64_t M, int64_t N,
                                int64_t K, int64_t K_weight, int qt, float *ws, int64_t ws_bytes, hipStream_t st) {
    auto kern = k_gemm_small<T, NESTED, MF>;
    constexpr int lds = gemm_small_lds_bytes<MF>();
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(small)")) return rc;
    int64_t slices = gemm_small_slices(M, N, K);
    const int64_t steps = K / 256;
    if (slices > 1 && (ws == nullptr || ws_bytes < slices * M * N * 4 || (reinterpret_cast<uintptr_t>(ws) & 15))) {
        if (steps > 8) return 1;      // no room for the partials and too long a K for one slice: the caller falls through
        slices = 1;
    }
    const int64_t kps = ((steps + slices - 1) / slices) * 256;
    const int64_t used = (K + kps - 1) / kps;
    const int od = sizeof(OutT) == 4 ? MBNB_F32 : (std::is_same<OutT, f16_t>::value ? MBNB_F16 : MBNB_BF16);
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)used, (unsigned)((M + 16 * MF - 1) / (16 * MF)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, packed, am, bias, static_cast<void *>(out), od, used > 1 ? ws : nullptr, M, N, K,
                       K_weight, kps, qt);
    if (used <= 1) {
        set_kernel_name("mfma_small");
        return check_launch("matmul_4bit(small)");
    }
    if (int rc = check_launch("matmul_4bit(small split-K)")) return rc;
    const int64_t groups = M * ((N + 3) / 4);
    hipLaunchKernelGGL((k_splitk_reduce_rm<T, OutT>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, ws, (int)used, bias, out, M, N);
    set_kernel_name("mfma_small_splitk");
    return check_launch("matmul_4bit(small split-K reduce)");
}

template <typename T, typename OutT, bool NESTED>
int launch_gemm_small(const T *x, const uint8_t *packed, const AbsmaxView &am, const T *bias, OutT *out, int64_t M, int64_t N, int64_t K,
                      int64_t K_weight, int qt, float *ws, int64_t ws_bytes, hipStream_t st) {
    if (M <= 64) return launch_gemm_small_mf<T, OutT, NESTED, 4>(x, packed, am, bias, out, M, N, K, K_weight, qt, ws, ws_bytes, st);
    return launch_gemm_small_mf<T, OutT, NESTED, 8>(x, packed, am, bias, out, M, N, K, K_weight, qt, ws, ws_bytes, st);
}

#define MBNB_INST(T, OutT, NESTED)                                                                                          \
    template int launch_gemm_small<T, OutT, NESTED>(const T *, const uint8_t *, const AbsmaxView &, const T *, OutT *, int64_t, \
                                                    int64_t, int64_t, int64_t, int, float *, int64_t, hipStream_t);
MBNB_INST(f16_t, f16_t, false) MBNB_INST(f16_t, f16_t, true) MBNB_INST(f16_t, bf16_t, false) MBNB_INST(f16_t, bf16_t, true)
MBNB_INST(f16_t, float, false) MBNB_INST(f16_t, float, true)
MBNB_INST(bf16_t, f16_t, false) MBNB_INST(bf16_t, f16_t, true) MBNB_INST(bf16_t, bf16_t, false) MBNB_INST(bf16_t, bf16_t, true)
MBNB_INST(bf16_t, float, false) MBNB_INST(bf16_t, float, true)
#undef MBNB_INST

}  // namespace mbnb
