// gemm_f32.hip — matmul_4bit for an f32 QuantState.dtype (the weight of a default nn.Linear, functional.py:756-773:
// `A.to(weight.dtype) @ dequantize_4bit(B).T` in f32) at more than a handful of rows.
//
// The 16-bit kernels do not apply (their operands are 16-bit), and the generic one-wave-per-column kernel re-reads the
// packed weight per 8 rows: 1.2 TFLOP/s at 4096^2.  Same two steps as the 16-bit decode-once path (gemm_dense.hip):
// dequantize_4bit into the caller's workspace as f32 [N, K_weight] — exactly the values the reference multiplies —
// then a dense f32 GEMM on v_mfma_f32_32x32x2_f32 (f32 products, f32 accumulation: no precision is given up).
//
// k_gemm_f32<BT>: workgroup tile BT x BT (128 or 64), four waves as 2 x 2, each (BT/2)^2 in 32 x 32 fragments; k in
// steps of 32 through a double-buffered LDS image (rows of 32 floats at a 144-byte pitch: the lanes' 16-byte fragment
// reads fall on distinct bank groups), the next step's global loads in flight under this step's products.  The f32 MFMA
// takes 64 cycles per 32 x 32 x 2 product, so the loads and the LDS port idle most of the time: the kernel is bound by
// the f32 matrix rate (157 TFLOP/s nominal).  A lane's four consecutive k of a fragment row are one ds_read_b128; the
// MFMA's two k positions are the lane halves, so operand j of half h multiplies k = 8 q + 4 h + j on both sides.
#include "gemm_mid.h"

namespace mbnb {

int dequantize_4bit_dispatch(const uint8_t *, const AbsmaxView &, int64_t, int64_t, int64_t, int, int, int, void *, hipStream_t, int store_policy = 0);

constexpr int GF_BK = 32;
constexpr int GF_PITCH = 36;   // floats per LDS row (144 bytes)

template <int BT> constexpr int gemm_f32_lds_bytes() { return 2 * 2 * BT * GF_PITCH * 4; }

template <int BT, bool SPLITK>
__global__ __launch_bounds__(256) void k_gemm_f32(const float *__restrict__ X, const float *__restrict__ Wd,
                                                  const float *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                  float *__restrict__ partial, int64_t M, int64_t N, int64_t K, int64_t ldw,
                                                  int64_t k_per_slice) {
    constexpr int WT = BT / 2, F = WT / 32, LD = BT / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [stage][A / B][BT][GF_PITCH]
    constexpr int OPER = BT * GF_PITCH, STAGE = 2 * OPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t tiles_m = (M + BT - 1) / BT, tiles = tiles_m * ((N + BT - 1) / BT);
    const int64_t tile = SPLITK ? (int64_t)blockIdx.x % tiles : (int64_t)blockIdx.x;
    const int64_t slice = SPLITK ? (int64_t)blockIdx.x / tiles : 0;
    // m fastest: the workgroups in flight together share few weight rows (each [BT, K] weight panel is read from HBM once)
    const int64_t m0 = (tile % tiles_m) * BT, n0 = (tile / tiles_m) * BT;
    const int64_t k_begin = SPLITK ? slice * k_per_slice : 0;
    const int64_t k_end = SPLITK ? (k_begin + k_per_slice < K ? k_begin + k_per_slice : K) : K;

    const int lr = tid >> 3, lc = (tid & 7) * 4;
    f32x4 ga[LD], gb[LD];
    auto gload = [&](int64_t k0) {
        const int64_t k = k0 + lc;
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int64_t m = m0 + lr + 32 * i, n = n0 + lr + 32 * i;
            ga[i] = (m < M && k < k_end) ? *reinterpret_cast<const f32x4 *>(X + m * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            gb[i] = (n < N && k < k_end) ? *reinterpret_cast<const f32x4 *>(Wd + n * ldw + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int s) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            *reinterpret_cast<f32x4 *>(&lds[s * STAGE + (lr + 32 * i) * GF_PITCH + lc]) = ga[i];
            *reinterpret_cast<f32x4 *>(&lds[s * STAGE + OPER + (lr + 32 * i) * GF_PITCH + lc]) = gb[i];
        }
    };

    f32x16 acc[F][F];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int64_t nk = (k_end - k_begin + GF_BK - 1) / GF_BK;
    gload(k_begin);
    lstore(0);
    __syncthreads();
    const int fo_a = (wm * WT + (lane & 31)) * GF_PITCH + (lane >> 5) * 4;
    const int fo_b = OPER + (wn * WT + (lane & 31)) * GF_PITCH + (lane >> 5) * 4;
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int s = (int)(kt & 1);
        if (kt + 1 < nk) gload(k_begin + (kt + 1) * GF_BK);
#pragma unroll
        for (int h = 0; h < GF_BK / 8; ++h) {
            f32x4 a[F], b[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                a[f] = *reinterpret_cast<const f32x4 *>(&lds[s * STAGE + fo_a + f * 32 * GF_PITCH + h * 8]);
                b[f] = *reinterpret_cast<const f32x4 *>(&lds[s * STAGE + fo_b + f * 32 * GF_PITCH + h * 8]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int fm = 0; fm < F; ++fm)
#pragma unroll
                    for (int fn = 0; fn < F; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][j], b[fn][j], acc[fm][fn], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(s ^ 1);   // the other stage: last read in step kt - 1, before the barrier that ended it
        __syncthreads();
    }

    // C fragment: register r of a lane = row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32
    float *part = SPLITK ? partial + slice * M * N : nullptr;
#pragma unroll
    for (int fn = 0; fn < F; ++fn) {
        const int64_t n = n0 + wn * WT + fn * 32 + (lane & 31);
        if (n >= N) continue;
        const float bv = (!SPLITK && bias) ? bias[n] : 0.f;
#pragma unroll
        for (int fm = 0; fm < F; ++fm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WT + fm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (m >= M) continue;
                float v = acc[fm][fn][r];
                const int64_t o = m * N + n;
                if (SPLITK) {
                    part[o] = v;
                    continue;
                }
                if (bias) v += bv;   // functional.py:765-766: f32 sum, then the bias in f32
                if (out_dtype == MBNB_F32) static_cast<float *>(out_v)[o] = v;
                else if (out_dtype == MBNB_F16) static_cast<f16_t *>(out_v)[o] = from_f32<f16_t>(v);
                else static_cast<bf16_t *>(out_v)[o] = from_f32<bf16_t>(v);
            }
        }
    }
}

// Where two (three with split-K) launches beat the generic kernel, which streams the packed weight once per 8 rows
// (tools/exp/f32_crossover.py, r = N K / 4096^2: generic ~ 4 + ceil(M / 8) (0.15 + 17 r) us, this path ~ 18 + 31 r us at small
// M): from 17 rows at 4096^2, from ~100 rows at 1024^2.  K % 4 == 0: rows of X are read 16 bytes at a time.
bool gemm_f32_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (M < 5 || N < 32 || K < 16 || K % 4 != 0 || K_weight % 4 != 0) return false;
    const double r = (double)N * (double)K / 16777216.0;
    return (double)((M + 7) / 8) * (0.15 + 17.0 * r) + 4.0 > 18.0 + 31.0 * r;
}

// 128 x 128 tiles from one per CU up (two fit a CU: 72 KiB of LDS each; 1024 x 4096^2: 336 us against 352 us on 64 x 64
// tiles); otherwise 64 x 64 tiles (up to four
// per CU), and K cut into slices of at least 256 until about 1024 workgroups exist -- one wave per SIMD cannot cover the
// load latency with a one-step prefetch, four can.  Partials [slice][M][N] f32 behind the weight, added in slice order.
struct F32Plan {
    int bt;
    int64_t slices, kps;
};
static F32Plan gemm_f32_plan(int64_t M, int64_t N, int64_t K) {
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    if (t128 >= 256) return {128, 1, K};
    const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
    int64_t s = (1024 + t64 - 1) / t64;
    if (s > K / 256) s = K / 256;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    const int64_t kps = (((K + s - 1) / s + GF_BK - 1) / GF_BK) * GF_BK;
    return {64, (K + kps - 1) / kps, kps};
}

static int64_t gemm_f32_wd_bytes(int64_t N, int64_t K_weight) { return ((N * K_weight * 4 + 255) / 256) * 256; }

int64_t gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t K_weight) {
    if (!gemm_f32_shape(M, N, K, K_weight)) return 0;
    const F32Plan p = gemm_f32_plan(M, N, K);
    return gemm_f32_wd_bytes(N, K_weight) + (p.slices > 1 ? p.slices * M * N * 4 : 0);
}

template <int BT>
static int launch_gemm_f32(const float *x, const float *wd, const float *b, void *out, int out_dtype, float *partial, int64_t M,
                           int64_t N, int64_t K, int64_t ldw, int64_t slices, int64_t kps, hipStream_t st) {
    const int64_t tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
    constexpr int lds = gemm_f32_lds_bytes<BT>();
    if (slices <= 1) {
        auto kern = k_gemm_f32<BT, false>;
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(dense_f32)")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wd, b, out, out_dtype, partial, M, N, K, ldw, K);
        return check_launch("matmul_4bit(dense_f32)");
    }
    auto kern = k_gemm_f32<BT, true>;
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(dense_f32 split-K)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * slices)), dim3(256), lds, st, x, wd, b, out, out_dtype, partial, M, N, K, ldw, kps);
    if (int rc = check_launch("matmul_4bit(dense_f32 split-K)")) return rc;
    const int64_t groups = M * ((N + 3) / 4);
    const unsigned blocks = (unsigned)((groups + 255) / 256);
    if (out_dtype == MBNB_F32)
        hipLaunchKernelGGL((k_splitk_reduce_rm<float, float>), dim3(blocks), dim3(256), 0, st, partial, (int)slices, b, static_cast<float *>(out), M, N);
    else if (out_dtype == MBNB_F16)
        hipLaunchKernelGGL((k_splitk_reduce_rm<float, f16_t>), dim3(blocks), dim3(256), 0, st, partial, (int)slices, b, static_cast<f16_t *>(out), M, N);
    else
        hipLaunchKernelGGL((k_splitk_reduce_rm<float, bf16_t>), dim3(blocks), dim3(256), 0, st, partial, (int)slices, b, static_cast<bf16_t *>(out), M, N);
    return check_launch("matmul_4bit(dense_f32 split-K reduce)");
}

// Returns MBNB_NOT_APPLICABLE when the path does not apply (the caller continues to the generic kernel), otherwise the launch status.
int matmul_4bit_f32_path(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                         int64_t K_weight, int blocksize, int qt, const void *bias, int out_dtype, void *out, void *ws,
                         int64_t ws_bytes, hipStream_t st) {
    if (ws == nullptr || !gemm_f32_shape(M, N, K, K_weight)) return MBNB_NOT_APPLICABLE;
    const int64_t wd_bytes = gemm_f32_wd_bytes(N, K_weight);
    if (ws_bytes < wd_bytes) return MBNB_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(ws) & 15)) return MBNB_NOT_APPLICABLE;
    F32Plan p = gemm_f32_plan(M, N, K);
    if (p.slices > 1 && ws_bytes < wd_bytes + p.slices * M * N * 4) p = {p.bt, 1, K};   // a short workspace costs the split, not the path
    float *wd = static_cast<float *>(ws);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(ws) + wd_bytes);
    if (int rc = dequantize_4bit_dispatch(packed, am, N, K_weight, K_weight, blocksize, qt, MBNB_F32, wd, st)) return rc;
    const float *x = static_cast<const float *>(A), *b = static_cast<const float *>(bias);
    const int rc = p.bt == 128 ? launch_gemm_f32<128>(x, wd, b, out, out_dtype, partial, M, N, K, K_weight, p.slices, p.kps, st)
                               : launch_gemm_f32<64>(x, wd, b, out, out_dtype, partial, M, N, K, K_weight, p.slices, p.kps, st);
    set_kernel_name(p.slices > 1 ? "dequant+dense_f32_splitk" : "dequant+dense_f32");
    return rc;
}

}  // namespace mbnb
