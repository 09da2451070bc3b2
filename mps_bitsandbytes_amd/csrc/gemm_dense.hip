// gemm_dense.hip — "decode once" path of matmul_4bit for large M: dequantize_4bit writes the weight [N, K_weight] in the
// compute dtype into the caller's workspace, k_gemm_dense (gemm_dense.h) multiplies.  Any blocksize / quant type / nested
// absmax (the decode is dequantize_4bit's), f16 and bf16 compute.  Own translation unit (the kernel compiles for minutes).
//
// Workspace layout (mbnb_matmul_4bit_workspace_bytes reports the sum):
//   [0, wd_bytes)                    Wd [N, K_weight] in the compute dtype, wd_bytes = N * K_weight * 2 rounded up to 256
//   [wd_bytes, + slices * M * N * 4) f32 partials of the split-K slices (none when slices == 1)
#include <cstdio>
#include "gemm_dense.h"
#include "gemm_dense128.h"
#include "gemm_mid.h"

namespace mbnb {

int dequantize_4bit_dispatch(const uint8_t *, const AbsmaxView &, int64_t, int64_t, int64_t, int, int, int, void *, hipStream_t, int store_policy = 0);

// The policy, from tools/exp/sweep_dense.py (profiles/r02_dense_sweep.txt, r02_dense_sweep2.txt, r02_dense_sweep3.txt).  The
// path is taken from 256 rows and 1.5 M outputs up (gemm_dense_shape; below that the fused split-K kernels win).  A plan = (wave-tile m fragments FM: 8 -> 256 x 256 tiles, 4 -> 256 n x 128 m tiles; K slices s), the
// pair minimising     rounds(FM, s) * k_steps(s) * step(FM)  +  (s > 1) * 8 s M N bytes / 6 TB/s
// -- workgroup rounds on 256 CUs times the measured k-step (1.3 us / 0.75 us), plus the f32 partials written once and read
// once (what makes split-K expensive here: two slices of a 4096 x 4096 output move 256 MB, 36 us measured).
// From 96 tiles of 256 x 256 up never split, although two slices measure up to 13 % faster at 96-128 tiles: an unsplit product
// has the same bits for a row whatever M it is computed in (the tile shape does not change a row's summation order, the
// slice count does), which is what lets row shards and row chunks of a large batch (sharding.py, bench.py --gpus N
// --verify) be compared bit for bit with the unsharded result -- for shards and chunks of MORE THAN 512 ROWS: since round 3
// matmul_4bit_dispatch offers 257-512 rows to k_gemm_small first (gemm_small_one_round), whose summation order is its own.
// Round 3: a third tile shape, 128 x 128 on three LDS stages (fm = 2, gemm_dense128.h; never split): 0.47 us per k-step with every
// CU busy, 0.36 us at <= 128 tiles (tools/exp/ab_dense128.py, profiles/r03_dense128_ab.txt).  It replaces the split plans wherever
// its tiles fit the chip in one round: 1024 x 4096 x 4096 35.9 us against 45.9 (256 x 128 tiles x 2 slices + the reduction pass),
// 512 x 4096^2 28.3 against 34.1, 768 x 4096^2 31.0 against 40.0, 1000 x 2600 x 1024 10.3 against 15.5; two rounds of it lose to one
// round of 256 x 128 tiles (1536 x 4096^2: 59.6 against 52.5).  A split plan is charged 6 us for its second launch.  Results equal
// the unsplit 256-wide tiles' bit for bit (same summation order per row).
struct DensePlan { int fm; int64_t slices; };
DensePlan gemm_dense_plan(int64_t M, int64_t N, int64_t K) {
    const int64_t tn = (N + 255) / 256, tiles8 = ((M + 255) / 256) * tn, tiles4 = ((M + 127) / 128) * tn;
    const int64_t tiles2 = ((M + 127) / 128) * ((N + 127) / 128);
    const int64_t steps = K / 64;
    DensePlan best{8, 1};
    double best_t = 1e30;
    if (K >= 192 && tiles8 < 96) {     // from 96 tiles of 256 x 256 up the big tiles stay (and nothing is ever split there)
        const int64_t rounds = (tiles2 + 255) / 256;
        best_t = (double)rounds * (double)steps * (tiles2 <= 128 ? 0.36 : 0.47);
        best = DensePlan{2, 1};
    }
    for (int fm = 8; fm >= 4; fm -= 4) {
        const int64_t tiles = fm == 8 ? tiles8 : tiles4;
        const double step = fm == 8 ? 1.3 : 0.75;
        for (int64_t s = 1; s <= 8; s++) {
            const int64_t per = (steps + s - 1) / s;
            if (s > 1 && (per < 8 || tiles8 >= 96)) break;
            const int64_t rounds = (tiles * s + 255) / 256;
            const double t = (double)rounds * (double)per * step + (s > 1 ? 6.0 + 8.0 * (double)s * (double)M * (double)N / 6.0e6 : 0.0);
            if (t < best_t * (fm == 8 ? 1.0 : 0.97)) {   // the smaller tile has to win by a margin
                best_t = t;
                best = DensePlan{fm, s};
            }
        }
    }
    return best;
}
int64_t gemm_dense_slices(int64_t M, int64_t N, int64_t K) { return gemm_dense_plan(M, N, K).slices; }

// Round 4: column-balanced grids for the unsplit 256-row tile (k_gemm_dense_nb, gemm_dense.h).  Uniform 256-wide columns give
// tiles_m x ceil(N / 256) tiles, and what is left after the last whole round of 256 CUs runs as a partial round: 4096 x 11008 = 688 tiles =
// 2.69 rounds, 4096 x 13824 = 3.375, 5000 x 5120 = 1.56.  VERDICT r3 priced that partial round as a whole one ("10 % wave quantisation").
// Measured (tools/exp/nb_rounds.py, profiles/r04_nb_rounds.txt; M = K = 4096, us per round of 256 tiles): a round of 256-wide tiles 90-92,
// of 224-wide 81-84 (0.90-0.92, not 0.875), of 192-wide 70-72 (0.78, not 0.75): narrower tiles pay the same A pieces, barriers, prologue and
// epilogue for fewer MFMAs.  And a PARTIAL round of f x 256 tiles costs max(0.66, 0.3 + 0.7 f) of a whole one, not 1: with fewer CUs busy the
// chip holds a higher clock (DVFS give-back), so uniform columns already sit within 2-6 % of perfect packing -- 4096 x 11008 measures
// 255.5 us uniform against 259-261 for 8 x 256 + 40 x 224 or 15 x 256 + 32 x 224 (profiles/r04_nb_forced.txt).  What balancing can still
// win is a small f at few rounds: 5000 x 5120 x 5120 207 -> 193 us (10 x 224 + 15 x 192), 4096 x 13824 x 5120 419 -> 404, 4096 x 14336 x 4096
// 342 -> 328 (all 224).  The plan therefore prices both forms with the measured costs and takes the balanced grid only where it is
// predicted to win by 4 %: `cols_a` columns of 32 fna and `cols_b` of 32 (fna - 1), the pair and the column count whose list schedule on
// 256 CUs -- wide tiles first, the order the grid is walked in -- ends earliest.  Any cut gives every output row the same summation order,
// so the choice is free of consequences for the bits.
struct NbPlan { int fna; int cols_a, cols_b; double t_us; };
// tile cost in us (multi-round rate): prologue + epilogue + k-steps, fitted on tools/exp/nb_rounds.py (K = 4096) and the K = 8192 rows of ab_dense_nb.py
static double nb_tile_cost(int fn, int64_t nk) {
    static const double fixed[4] = {6.5, 7.0, 7.5, 7.9}, step[4] = {0.85, 1.008, 1.18, 1.30};      // fn = 5, 6, 7, 8
    return fixed[fn - 5] + (double)nk * step[fn - 5];
}
static double nb_makespan(int64_t na, double ca, int64_t nb, double cb) {
    const int64_t P = 256, qa = na / P, ra = na % P;
    double lo = (double)qa * ca, hi = lo + ca;      // after the wide tiles: P - ra CUs are free at lo, ra CUs at hi
    const int64_t n_lo = P - ra, n_hi = ra;
    double end = na == 0 ? 0.0 : (ra ? hi : lo);
    while (nb > 0) {                                  // narrow tiles go to whichever group is free first, a group's worth at a time
        const bool use_lo = n_hi == 0 || lo <= hi;
        const int64_t cap = use_lo ? n_lo : n_hi;
        double &t = use_lo ? lo : hi;
        t += cb;
        if (t > end) end = t;
        nb -= nb < cap ? nb : cap;
    }
    return end;
}
// fm: the tile the uniform plan (gemm_dense_plan) chose, 8 = 256 x 256 or 4 = 256 (n) x 128 (m) -- the time to beat
NbPlan gemm_dense_nb_plan(int64_t M, int64_t N, int64_t K, int fm = 8) {
    const int64_t tiles_m = (M + 255) / 256, U = (N + 31) / 32, nk = K / 64;
    const int64_t c8 = (U + 7) / 8, tiles8 = tiles_m * c8;
    // uniform columns: whole rounds + the partial round at its measured discount
    const double f = (double)(tiles8 % 256) / 256.0, cost8 = nb_tile_cost(8, nk);
    double uniform_t = cost8 * ((double)(tiles8 / 256) + (f > 0.0 ? (0.3 + 0.7 * f > 0.66 ? 0.3 + 0.7 * f : 0.66) : 0.0));
    if (fm == 4) {      // 256 x 128 tiles: 0.75 us per k-step, ~5 us of prologue + epilogue, whole rounds (profiles/r02_dense_sweep.txt, r04_ab_dense_nb.txt)
        const int64_t tiles4 = ((M + 127) / 128) * c8;
        uniform_t = (double)((tiles4 + 255) / 256) * (5.0 + 0.75 * (double)nk);
    }
    NbPlan best{8, (int)c8, 0, uniform_t};
    if (tiles8 <= 256 || f == 0.0 || tiles_m > 256 || c8 > (1 << 20)) return best;     // less than a round (smaller tiles serve those), or whole rounds already
    const int64_t span = 256 / tiles_m + 1;          // more than one extra round of columns never helps
    for (int fna = 8; fna >= 6; fna--) {
        const int fnb = fna - 1;
        const int64_t cmin = (U + fna - 1) / fna, cmax = (U + fnb - 1) / fnb;
        for (int64_t C = cmin; C <= cmax && C <= cmin + span; C++) {
            int64_t a = U - fnb * C;
            if (a < 0) a = 0;
            if (a > C) continue;
            // mixed widths measure 1-3 % over the sum of their rounds (nb_rounds.py, last columns)
            const double t = nb_makespan(tiles_m * a, nb_tile_cost(fna, nk), tiles_m * (C - a), nb_tile_cost(fnb, nk)) * ((a > 0 && a < C) ? 1.02 : 1.0);
            if (t < best.t_us) best = NbPlan{fna, (int)a, (int)(C - a), t};
        }
    }
    if (best.t_us > 0.96 * uniform_t) return NbPlan{8, (int)c8, 0, uniform_t};   // the narrow tiles have to win by a margin
    return best;
}
int64_t gemm_dense_k_per_slice(int64_t K, int64_t slices) {
    const int64_t steps = K / 64;
    return ((steps + slices - 1) / slices) * 64;
}
bool gemm_dense_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (K % 64 != 0 || K < 128 || K_weight % 8 != 0) return false;
    if (256 * (K > K_weight ? K : K_weight) * 2 >= ((int64_t)1 << 31)) return false;
    if (M * N * 4 >= ((int64_t)1 << 40)) return false;
    // from 256 rows and 1.5 M outputs up (sweep3: 384 x 4096 x 4096 35.5 us here, 43.2 fused; 256 x 4096 x 4096 stays fused, 28.4 vs 33.8)
    // (280-360 x 4096 x 4096: 34.4-34.7 us here against 36-39.5 for k_gemm_small with three m-tiles)
    return (M >= 256 && M * N >= 1500000) || (M > 256 && M * N >= 1000000);
}
int64_t gemm_dense_wd_bytes(int64_t N, int64_t K_weight) { return (N * K_weight * 2 + 255) & ~(int64_t)255; }
int64_t gemm_dense_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (!gemm_dense_shape(M, N, K, K_weight)) return 0;
    const int64_t s = gemm_dense_slices(M, N, K);
    return gemm_dense_wd_bytes(N, K_weight) + (s > 1 ? s * M * N * 4 : 0);
}

template <typename T, int FM>
static int launch_gemm_dense_fm(const T *x, const T *wd, const T *bias, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                                int64_t ldw, float *partial, int64_t slices, hipStream_t st) {
    const int64_t tiles = ((M + 32 * FM - 1) / (32 * FM)) * ((N + 255) / 256);
    if (slices <= 1) {
        auto kern = k_gemm_dense<T, false, FM>;
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_4bit(dense)")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, wd, bias, out, out_dtype, static_cast<float *>(nullptr),
                           M, N, K, ldw, K, static_cast<const float *>(nullptr), static_cast<const float *>(nullptr), OutlierEpilogue{});
        set_kernel_name(FM == 8 ? "dense 256x256" : "dense 256x128");
        return check_launch("matmul_4bit(dense)");
    }
    auto kern = k_gemm_dense<T, true, FM>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_4bit(dense split-K)")) return rc;
    const int64_t kps = gemm_dense_k_per_slice(K, slices);
    const int64_t used = (K + kps - 1) / kps;    // no empty slice
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * used)), dim3(256), GD_LDS, st, x, wd, bias, out, out_dtype, partial, M, N, K, ldw,
                       kps, static_cast<const float *>(nullptr), static_cast<const float *>(nullptr), OutlierEpilogue{});
    if (int rc = check_launch("matmul_4bit(dense split-K)")) return rc;
    const int64_t groups = M * ((N + 3) / 4);
    const unsigned blocks = (unsigned)((groups + 255) / 256);
    if (out_dtype == MBNB_F32)
        hipLaunchKernelGGL((k_splitk_reduce_rm<T, float>), dim3(blocks), dim3(256), 0, st, partial, (int)used, bias, static_cast<float *>(out), M, N);
    else if (out_dtype == MBNB_F16)
        hipLaunchKernelGGL((k_splitk_reduce_rm<T, f16_t>), dim3(blocks), dim3(256), 0, st, partial, (int)used, bias, static_cast<f16_t *>(out), M, N);
    else
        hipLaunchKernelGGL((k_splitk_reduce_rm<T, bf16_t>), dim3(blocks), dim3(256), 0, st, partial, (int)used, bias, static_cast<bf16_t *>(out), M, N);
    return check_launch("matmul_4bit(dense split-K reduce)");
}

// column-balanced grid of 256-row tiles (k_gemm_dense_nb)
template <typename T>
static int launch_gemm_dense_nb(const T *x, const T *wd, const T *bias, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                                int64_t ldw, const NbPlan &pl, hipStream_t st) {
    const int tiles_n = pl.cols_a + pl.cols_b;
    const int64_t tiles = ((M + 255) / 256) * tiles_n;
#define MBNB_NB(FNA, FNB)                                                                                                        \
    do {                                                                                                                         \
        auto kern = k_gemm_dense_nb<T, FNA, FNB>;                                                                                \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_4bit(dense nb)")) return rc;           \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, x, wd, bias, out, out_dtype, M, N, K, ldw, tiles_n, \
                           pl.cols_a);                                                                                           \
    } while (0)
    if (pl.fna == 8) MBNB_NB(8, 7);
    else if (pl.fna == 7) MBNB_NB(7, 6);
    else MBNB_NB(6, 5);
#undef MBNB_NB
    // the plan in the kernel name (mbnb_last_kernel after mbnb_gemm_dense; the matmul paths overwrite it with their own)
    static thread_local char name[64];
    snprintf(name, sizeof(name), "dense_nb %dx%d+%dx%d", pl.cols_a, 32 * pl.fna, pl.cols_b, 32 * (pl.fna - 1));
    set_kernel_name(name);
    return check_launch("matmul_4bit(dense nb)");
}

// 128 x 128 tiles (gemm_dense128.h), never split
template <typename T>
static int launch_gemm_dense128(const T *x, const T *wd, const T *bias, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                                int64_t ldw, hipStream_t st) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    auto kern = k_gemm_dense128<T>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), G128_LDS, "matmul_4bit(dense128)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), G128_LDS, st, x, wd, bias, out, out_dtype, M, N, K, ldw);
    return check_launch("matmul_4bit(dense128)");
}

// fm: 8 / 4 as planned by gemm_dense_plan (the column-balanced grid where its plan beats them), 2 = 128 x 128 tiles; 9 / 10 = 256 x 256 / 256 x 128
// tiles in uniform 256-wide columns, no balancing (diagnostic: what fm = 8 / 4 ran before round 4)
template <typename T>
static int launch_gemm_dense(const T *x, const T *wd, const T *bias, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                             int64_t ldw, float *partial, int64_t slices, int fm, hipStream_t st) {
    if (fm == 2) return launch_gemm_dense128<T>(x, wd, bias, out, out_dtype, M, N, K, ldw, st);
    if ((fm == 8 || fm == 4) && slices <= 1) {
        const NbPlan pl = gemm_dense_nb_plan(M, N, K, fm);
        if (pl.cols_b > 0 || pl.fna != 8) return launch_gemm_dense_nb<T>(x, wd, bias, out, out_dtype, M, N, K, ldw, pl, st);
    }
    if (fm == 4 || fm == 10) return launch_gemm_dense_fm<T, 4>(x, wd, bias, out, out_dtype, M, N, K, ldw, partial, slices, st);
    return launch_gemm_dense_fm<T, 8>(x, wd, bias, out, out_dtype, M, N, K, ldw, partial, slices, st);
}

// Returns MBNB_NOT_APPLICABLE when the path does not apply (caller falls through to the fused kernels), otherwise the launch status.
int matmul_4bit_dense_path(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                           int64_t K_weight, int blocksize, int qt, int w_dtype, const void *bias, int out_dtype, void *out,
                           void *ws, int64_t ws_bytes, hipStream_t st) {
    if (w_dtype != MBNB_F16 && w_dtype != MBNB_BF16) return MBNB_NOT_APPLICABLE;
    if (!gemm_dense_shape(M, N, K, K_weight) || ws == nullptr) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(ws) & 255) || (reinterpret_cast<uintptr_t>(packed) & 3)) return MBNB_NOT_APPLICABLE;
    const int64_t wd_bytes = gemm_dense_wd_bytes(N, K_weight);
    if (ws_bytes < wd_bytes) return MBNB_NOT_APPLICABLE;
    DensePlan plan = gemm_dense_plan(M, N, K);
    if (plan.slices > 1 && ws_bytes < wd_bytes + plan.slices * M * N * 4) plan.slices = 1;   // a short workspace costs the split, not the path
    const int64_t slices = plan.slices;
    char *wsb = static_cast<char *>(ws);
    if (int rc = dequantize_4bit_dispatch(packed, am, N, K_weight, K_weight, blocksize, qt, w_dtype, wsb, st, N * K_weight <= (int64_t(1) << 25) ? 2 : (M >= 2048 ? 1 : 0))) return rc;
    float *partial = reinterpret_cast<float *>(wsb + wd_bytes);
    int rc;
    if (w_dtype == MBNB_F16)
        rc = launch_gemm_dense<f16_t>(static_cast<const f16_t *>(A), reinterpret_cast<const f16_t *>(wsb), static_cast<const f16_t *>(bias),
                                      out, out_dtype, M, N, K, K_weight, partial, slices, plan.fm, st);
    else
        rc = launch_gemm_dense<bf16_t>(static_cast<const bf16_t *>(A), reinterpret_cast<const bf16_t *>(wsb),
                                       static_cast<const bf16_t *>(bias), out, out_dtype, M, N, K, K_weight, partial, slices, plan.fm, st);
    set_kernel_name(slices > 1 ? "dequant+dense_splitk" : "dequant+dense");   // tile shape: gemm_dense_plan (not part of the name)
    return rc;
}

int dequantize_rowwise_dispatch(const int8_t *, const float *, int64_t, int64_t, int, void *, hipStream_t, int store_policy = 0);
int dequantize_fp8_dispatch(const uint8_t *, const float *, int64_t, int64_t, int, void *, hipStream_t, int store_policy = 0);

// Linear8bit.forward / LinearFP8.forward at large M (nn/linear8bit.py:70-102, functional.py:796-807): the reference's own two
// steps -- dequantize_rowwise / dequantize_fp8_e4m3 into the compute dtype, then F.linear -- on the workspace.  Same policy
// and workspace layout as the 4-bit path.  Returns MBNB_NOT_APPLICABLE when it does not apply.
int linear8_dense_path(const void *X, int dtype, int64_t M, int64_t K, const void *W, const float *scales, int64_t N, bool fp8,
                       const void *bias, void *out, void *ws, int64_t ws_bytes, hipStream_t st) {
    if (dtype != MBNB_F16 && dtype != MBNB_BF16) return MBNB_NOT_APPLICABLE;
    if (!gemm_dense_shape(M, N, K, K) || ws == nullptr) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(ws) & 255)) return MBNB_NOT_APPLICABLE;
    const int64_t wd_bytes = gemm_dense_wd_bytes(N, K);
    if (ws_bytes < wd_bytes) return MBNB_NOT_APPLICABLE;
    DensePlan plan = gemm_dense_plan(M, N, K);
    if (plan.slices > 1 && ws_bytes < wd_bytes + plan.slices * M * N * 4) plan.slices = 1;
    const int64_t slices = plan.slices;
    char *wsb = static_cast<char *>(ws);
    // the pass's in-step form, as for the 4-bit weights (matmul_4bit_dense_path): four rows per thread + write-through stores on weights of up to
    // 32 Mi elements, write-through alone from 2048 rows up
    const int policy = N * K <= (int64_t(1) << 25) ? 2 : (M >= 2048 ? 1 : 0);
    const int rcq = fp8 ? dequantize_fp8_dispatch(static_cast<const uint8_t *>(W), scales, N, K, dtype, wsb, st, policy)
                        : dequantize_rowwise_dispatch(static_cast<const int8_t *>(W), scales, N, K, dtype, wsb, st, policy);
    if (rcq) return rcq;
    float *partial = reinterpret_cast<float *>(wsb + wd_bytes);
    int rc;
    if (dtype == MBNB_F16)
        rc = launch_gemm_dense<f16_t>(static_cast<const f16_t *>(X), reinterpret_cast<const f16_t *>(wsb), static_cast<const f16_t *>(bias),
                                      out, dtype, M, N, K, K, partial, slices, plan.fm, st);
    else
        rc = launch_gemm_dense<bf16_t>(static_cast<const bf16_t *>(X), reinterpret_cast<const bf16_t *>(wsb),
                                       static_cast<const bf16_t *>(bias), out, dtype, M, N, K, K, partial, slices, plan.fm, st);
    set_kernel_name(fp8 ? (slices > 1 ? "fp8a16_dequant+dense_splitk" : "fp8a16_dequant+dense")
                        : (slices > 1 ? "w8a16_dequant+dense_splitk" : "w8a16_dequant+dense"));
    return rc;
}

// matmul_int8 on the same kernel (k_gemm_dense<.., I8 = true>): A int8 [M, K], Bt = B^T int8 [N, K] (the caller's transpose),
// K % 128 == 0.  The 16-bit container type only moves bytes; out_dtype any of the three.
bool gemm_i8_dense_shape(int64_t M, int64_t N, int64_t K) {
    return (K % 128 == 0) && K >= 256 && 256 * K < ((int64_t)1 << 31) && ((M + 255) / 256) * ((N + 255) / 256) >= 96;
}
// ep != nullptr (OutlierAwareLinear): outlier term and bias in the epilogue; the caller checks gemm_i8_dense_outlier_ok first.
bool gemm_i8_dense_outlier_ok(const OutlierEpilogue &ep, int out_dtype) {
    return out_dtype != MBNB_F32 && (ep.x == nullptr || ep.n_out == 0 || (ep.ldx <= 64 && ep.ldx % 16 == 0 && (reinterpret_cast<uintptr_t>(ep.x) & 15) == 0));
}
int launch_gemm_i8_dense(const int8_t *A, const int8_t *Bt, const float *sA, const float *sB, int64_t M, int64_t N, int64_t K,
                         int out_dtype, void *out, hipStream_t st, const OutlierEpilogue *ep) {
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    if (ep != nullptr) {
        const bool two = ep->x != nullptr && ep->n_out > 0 && ep->ldx > 32;   // chunks of 32 outlier columns
        auto kern = out_dtype == MBNB_F16 ? (two ? k_gemm_dense<bf16_t, false, 8, true, 1, 2> : k_gemm_dense<bf16_t, false, 8, true, 1, 1>)
                                          : (two ? k_gemm_dense<bf16_t, false, 8, true, 2, 2> : k_gemm_dense<bf16_t, false, 8, true, 2, 1>);
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_int8(dense+outliers)")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, reinterpret_cast<const bf16_t *>(A),
                           reinterpret_cast<const bf16_t *>(Bt), static_cast<const bf16_t *>(nullptr), out, out_dtype,
                           static_cast<float *>(nullptr), M, N, K / 2, K / 2, K / 2, sA, sB, *ep);
        return check_launch("matmul_int8(dense+outliers)");
    }
    auto kern = k_gemm_dense<bf16_t, false, 8, true>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), GD_LDS, "matmul_int8(dense)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), GD_LDS, st, reinterpret_cast<const bf16_t *>(A),
                       reinterpret_cast<const bf16_t *>(Bt), static_cast<const bf16_t *>(nullptr), out, out_dtype,
                       static_cast<float *>(nullptr), M, N, K / 2, K / 2, K / 2, sA, sB, OutlierEpilogue{});
    return check_launch("matmul_int8(dense)");
}

// diagnostic entry for tools/exp (the dense kernel alone on a caller-made Wd)
// tile codes (x 128): 128 -> 256 x 128 tiles, 256 -> 256 x 256 tiles with UNIFORM columns (no column balancing), 384 -> 128 x 128 tiles, 0 -> the plan;
// 640 / 768 / 896 (codes 5 / 6 / 7) -> a FORCED column-balanced grid: `forced_cols_a` columns of 32 (code + 1), the rest of N in columns of 32 code
int gemm_dense_direct(const void *A, const void *Wd, int dtype, const void *bias, int out_dtype, void *out, int64_t M, int64_t N,
                      int64_t K, int64_t ldw, float *partial, int64_t slices, int tile_m, int forced_cols_a, hipStream_t st) {
    if (tile_m >= 640) {
        const int fnb = tile_m / 128, fna = fnb + 1;
        const int64_t U = (N + 31) / 32;
        int64_t a = forced_cols_a;
        if (a * fna > U) a = U / fna;
        const int64_t b = (U - a * fna + fnb - 1) / fnb;
        const NbPlan pl{fna, (int)a, (int)b, 0.0};
        if (dtype == MBNB_F16)
            return launch_gemm_dense_nb<f16_t>(static_cast<const f16_t *>(A), static_cast<const f16_t *>(Wd), static_cast<const f16_t *>(bias), out, out_dtype, M, N, K, ldw, pl, st);
        return launch_gemm_dense_nb<bf16_t>(static_cast<const bf16_t *>(A), static_cast<const bf16_t *>(Wd), static_cast<const bf16_t *>(bias), out, out_dtype, M, N, K, ldw, pl, st);
    }
    const int fm = tile_m == 128 ? 10 : (tile_m == 256 ? 9 : (tile_m == 384 ? 2 : gemm_dense_plan(M, N, K).fm));   // 9 / 10: that tile in uniform columns, no balancing
    if (dtype == MBNB_F16)
        return launch_gemm_dense<f16_t>(static_cast<const f16_t *>(A), static_cast<const f16_t *>(Wd), static_cast<const f16_t *>(bias), out,
                                        out_dtype, M, N, K, ldw, partial, slices, fm, st);
    return launch_gemm_dense<bf16_t>(static_cast<const bf16_t *>(A), static_cast<const bf16_t *>(Wd), static_cast<const bf16_t *>(bias), out,
                                     out_dtype, M, N, K, ldw, partial, slices, fm, st);
}

}  // namespace mbnb
