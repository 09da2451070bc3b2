// gemm_small.hip — launch of k_gemm_small (gemm_small.h): fused 4-bit GEMM for 16/32 < M <= 256 rows, blocksize >= 32.
#include "gemm_small.h"
#include "gemm_mid.h"

namespace mbnb {

static bool small_shape_up_to(int64_t M, int64_t N, int64_t K, int64_t K_weight, int64_t max_m, int64_t min_m) {
    // from min_m + 1 rows (17 for layers of more than 16 Mi weights) the weight-streaming skinny kernel loses.  4-bit form, round 3 (streamed weights): 32 x 4096 x 4096
    // 10.9 us here, 11.5 there; 24 rows 10.5 vs 10.4; 17 rows 9.9 vs 9.1; 32 x 11008 x 4096 21.0 vs 31.0 (tools/exp/small_check.py) -> from 29 rows; the W8A16 form keeps 33
    return (M > min_m || (M > 16 && N * K > ((int64_t)1 << 24))) && M <= max_m && K % 256 == 0 && K >= 512 && K <= 16 * 2048 && K_weight % 256 == 0 && N >= 64 &&
           256 * K * 2 < ((int64_t)1 << 31);
}
// 4-bit weights: up to 512 rows (round 3: k_gemm_small<.., MAXS_ = 16> holds 16 steps of weights in registers -- K = 4096 in one slice,
// 384 < M <= 512 on a 4096-wide layer = 256 workgroups, one round, no partials); the W8A16 form (gemm_small8.h) keeps 384 rows / 8 steps.
bool gemm_small_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight) { return small_shape_up_to(M, N, K, K_weight, 512, 28); }
bool gemm_small8_shape(int64_t M, int64_t N, int64_t K) { return small_shape_up_to(M, N, K, K, 384, 32); }
// Plan = (NF: n-fragments per wave -> 64 NF weight rows per workgroup; K slices).  The kernel is bound by what a workgroup takes
// in (activation tile 16 MF rows x 256 k per step against 32 NF bytes of weights per lane), so per step a workgroup costs
// about  0.4 us + its activation KiB / 55 GB/s  (small_check.py: 64 KiB -> 1.6 us, 32 KiB -> 1.0 us), a slice pays a prologue
// of ~2.5 us, and every extra slice adds M x N x 4 bytes of partials written and read (6 TB/s).
struct SmallPlan { int nf; int64_t slices; double us; int mf; };
// Round 3 (after the weights streamed in and the pipeline crossed the step boundary; tools/exp/small_stamps.py, small_nf2.py): a step of the 64-row
// form (NF = 1, 64 MFMAs per wave) takes ~3400 cycles = 1.5-1.6 us whatever the slice length, a step of the 128-row form (NF = 2, 128 MFMAs) ~4760 =
// 2.1 us with half the CUs busy and ~3.0 us with all of them (512 x 8192 x 4096: 50.8 us) -- 10-30 % less per weight row, because the activation tile (the
// CU's inflow limit) is taken in once for twice the rows.  The 128-row form
// halves the workgroups, so it pays where the 64-row form would need a second round (512 x 8192 x 4096: 256 workgroups instead of 512).
SmallPlan gemm_small_plan(int64_t M, int64_t N, int64_t K, int maxs_mf8 = 16) {
    const int64_t steps = K / 256;
    SmallPlan best{1, 1, 1e30, M > 64 ? 8 : 4};
    double best_t = 1e30;
    // Tile heights.  Up to 64 rows: 64 (MF = 4).  Above 256 rows: 128 (MF = 8; NF = 2 there, k_gemm_small<.., 8, 2, 16>).  In between both
    // are tried (the 4-bit kernel only: maxs_mf8 == 16): 64-row tiles double the workgroups, which often saves the K split and its reduction
    // launch -- 256 x 4096^2: 256 workgroups x 16 steps in ONE slice 18.0 us against 128 x 2 slices + reduction 22.4
    // (tools/exp/ab_small_pk.py, ab_small_mf.py).
    const int mf_lo = M > 64 && (M > 256 || maxs_mf8 != 16) ? 8 : 4, mf_hi = M > 64 ? 8 : 4;
    for (int mf = mf_lo; mf <= mf_hi; mf += 4) {
        const int64_t mt = (M + 16 * mf - 1) / (16 * mf);
        const int nf_max = (M > 256 && maxs_mf8 == 16) ? 2 : 1;
        for (int nf = 1; nf <= nf_max; nf++) {
            const int64_t wgs = ((N + 64 * nf - 1) / (64 * nf)) * mt, maxs = mf == 8 ? maxs_mf8 : (M > 64 ? 16 : 8);
            const double step_us = nf == 2 ? 3.0 : (M > 256 ? 1.6 : 0.4 + (double)(mf * 8) / 55.0);     // the plans up to 256 rows keep their tuned constants
            for (int64_t s = 1; s <= 16 && s <= steps; s++) {
                const int64_t per = (steps + s - 1) / s;
                if (per > maxs) continue;
                // the reduction launch: 3-4 us + the boundary (288 x 4096 x 4096 in two slices: 38.1 us); the plans up to 256 rows were tuned
                // with 2.0 and keep it where only one tile height is tried
                const double split_fixed = M > 256 ? 5.0 : (mf_lo != mf_hi ? 4.5 : 2.0);
                const double t = (double)((wgs * s + 255) / 256) * (2.5 + (double)per * step_us) + (s > 1 ? 8.0 * (double)s * (double)M * (double)N / 6.0e6 + split_fixed : 0.0);
                if (t < best_t - 1e-9) {
                    best_t = t;
                    best = SmallPlan{nf, s, t, mf};
                }
            }
        }
    }
    return best;
}
int64_t gemm_small_slices(int64_t M, int64_t N, int64_t K) { return gemm_small_plan(M, N, K).slices; }
// 256 < M <= 512 where the planned workgroups fit the chip in ONE round: k_gemm_small beats dequantise + dense there (512 x 4096^2
// 35.6 vs 40.6 us, 512 x 4096 x 2048 19.5 vs 27.5, 512 x 2048 x 4096 24.9 vs 34.8; two rounds lose: 400 x 5120 x 4096 56 vs 42,
// tools/exp/ab_small16.py, profiles/r03_small16_ab.txt), so matmul_4bit_dispatch tries it BEFORE the decode-once path.  A row's bits
// then depend on M below 513 rows (another summation order than the dense tiles'), as they always did below 257.
bool gemm_small_one_round(int64_t M, int64_t N, int64_t K, int64_t K_weight, int64_t ws_bytes) {
    if (M <= 256 || !gemm_small_shape(M, N, K, K_weight)) return false;
    const SmallPlan plan = gemm_small_plan(M, N, K);
    const int64_t wgs = ((N + 64 * plan.nf - 1) / (64 * plan.nf)) * ((M + 127) / 128) * plan.slices;
    if (wgs > 256 || (plan.slices > 1 && ws_bytes < plan.slices * M * N * 4)) return false;
    // against what the decode-once path would take: the dequantise pass (2.53 bytes per weight at 4.4 TB/s) + its boundary + 128 x 128 tiles
    const int64_t tiles2 = ((M + 127) / 128) * ((N + 127) / 128);
    // (weights of up to 32 Mi elements: the pass runs four dwords per thread with write-through stores, quant_kernels.hip, and the step is
    // ~3 us shorter: 400 x 5120 x 4096 43.6 -> 39.3 us)
    const double boundary = N * K_weight <= (int64_t(1) << 25) ? 0.6 : 3.6;
    const double dense_us = (double)N * (double)K_weight * 2.53 / 4.4e6 + boundary + (double)((tiles2 + 255) / 256) * (double)(K / 64) * (tiles2 <= 128 ? 0.36 : 0.47) + 2.0;
    return plan.us < dense_us * 1.05;
}
int64_t gemm_small8_slices(int64_t M, int64_t N, int64_t K) { return gemm_small_plan(M, N, K, 8).slices; }
int64_t gemm_small8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (!gemm_small8_shape(M, N, K)) return 0;
    const int64_t s = gemm_small8_slices(M, N, K);
    return s > 1 ? s * M * N * 4 : 0;
}
int64_t gemm_small_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (!gemm_small_shape(M, N, K, K_weight)) return 0;
    const int64_t s = gemm_small_slices(M, N, K);
    return s > 1 ? s * M * N * 4 : 0;
}

template <typename T, typename OutT, bool NESTED, int MF, int NF, int MAXS = 0>
static int launch_gemm_small_mf(const T *x, const uint8_t *packed, const AbsmaxView &am, const T *bias, OutT *out, int64_t M, int64_t N,
                                int64_t K, int64_t K_weight, int qt, int bs_shift, float *ws, int64_t ws_bytes, int64_t slices, hipStream_t st) {
    auto kern = k_gemm_small<T, NESTED, MF, NF, MAXS>;
    constexpr int lds = gemm_small_lds_bytes<MF>();
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(small)")) return rc;
    const int64_t steps = K / 256;
    const int64_t kps = ((steps + slices - 1) / slices) * 256;
    const int64_t used = (K + kps - 1) / kps;
    const int od = sizeof(OutT) == 4 ? MBNB_F32 : (std::is_same<OutT, f16_t>::value ? MBNB_F16 : MBNB_BF16);
    const dim3 grid((unsigned)((N + 64 * NF - 1) / (64 * NF)), (unsigned)used, (unsigned)((M + 16 * MF - 1) / (16 * MF)));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, packed, am, bias, static_cast<void *>(out), od, used > 1 ? ws : nullptr, M, N, K,
                       K_weight, kps, qt, bs_shift);
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

// Returns MBNB_NOT_APPLICABLE when the kernel cannot serve the call (no workspace for the partials and K too long for one slice).
template <typename T, typename OutT, bool NESTED>
int launch_gemm_small(const T *x, const uint8_t *packed, const AbsmaxView &am, const T *bias, OutT *out, int64_t M, int64_t N, int64_t K,
                      int64_t K_weight, int qt, int bs_shift, float *ws, int64_t ws_bytes, hipStream_t st) {
    SmallPlan plan = gemm_small_plan(M, N, K);
    if (plan.slices > 1 && (ws == nullptr || ws_bytes < plan.slices * M * N * 4 || (reinterpret_cast<uintptr_t>(ws) & 15))) {
        if (K / 256 > (M > 64 ? 16 : 8)) return MBNB_NOT_APPLICABLE;
        plan = SmallPlan{1, 1, 0.0, M > 64 ? 8 : 4};
    }
#define MBNB_SMALL(MF, NF) return launch_gemm_small_mf<T, OutT, NESTED, MF, NF>(x, packed, am, bias, out, M, N, K, K_weight, qt, bs_shift, ws, ws_bytes, plan.slices, st)
    if (M <= 64) {
        MBNB_SMALL(4, 1);
    }
    if (plan.mf == 4)
        return launch_gemm_small_mf<T, OutT, NESTED, 4, 1, 16>(x, packed, am, bias, out, M, N, K, K_weight, qt, bs_shift, ws, ws_bytes, plan.slices, st);
    if (plan.nf == 2)
        return launch_gemm_small_mf<T, OutT, NESTED, 8, 2, 16>(x, packed, am, bias, out, M, N, K, K_weight, qt, bs_shift, ws, ws_bytes, plan.slices, st);
    if ((K / 256 + plan.slices - 1) / plan.slices > 8)
        return launch_gemm_small_mf<T, OutT, NESTED, 8, 1, 16>(x, packed, am, bias, out, M, N, K, K_weight, qt, bs_shift, ws, ws_bytes, plan.slices, st);
    MBNB_SMALL(8, 1);
#undef MBNB_SMALL
}

#define MBNB_INST(T, OutT, NESTED)                                                                                          \
    template int launch_gemm_small<T, OutT, NESTED>(const T *, const uint8_t *, const AbsmaxView &, const T *, OutT *, int64_t, \
                                                    int64_t, int64_t, int64_t, int, int, float *, int64_t, hipStream_t);
MBNB_INST(f16_t, f16_t, false) MBNB_INST(f16_t, f16_t, true) MBNB_INST(f16_t, bf16_t, false) MBNB_INST(f16_t, bf16_t, true)
MBNB_INST(f16_t, float, false) MBNB_INST(f16_t, float, true)
MBNB_INST(bf16_t, f16_t, false) MBNB_INST(bf16_t, f16_t, true) MBNB_INST(bf16_t, bf16_t, false) MBNB_INST(bf16_t, bf16_t, true)
MBNB_INST(bf16_t, float, false) MBNB_INST(bf16_t, float, true)
#undef MBNB_INST

}  // namespace mbnb
