// matmul4_kernels.hip — fused 4-bit (NF4/FP4) dequant + matmul for gfx950.
//
// Replaces the reference's nf4/fp4 matmul kernels (mm:393-771, :859-1004, selected at
// mm:1987-1993) with three gfx950 kernels:
//   gemv     M <= 16   one wave per weight row pair, 16-byte packed loads straight to VGPRs,
//                      v_dot2 f32 accumulation, HBM-bound               (reference: nf4_matmul_simd)
//   mfma     M  > 16   LDS-tiled MFMA GEMM with the dequant in the B-tile producer (gemm_tile.h)
//                                                                      (reference: nf4_matmul_large/_fused)
//   generic  any shape / blocksize / f32: one wave per output row, scalar unpack
//                                                                      (reference: nf4_linear_simple)
// Numerics follow the reference CPU branch (functional.py:752-773): the decoded weight is
// rounded to the weight dtype before the contraction, accumulation is f32, one rounding of the
// result to the weight dtype, then a cast to the requested output dtype.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "gemm256.h"
#include "gemv4.h"
#include "gemv4_lean.h"

namespace mbnb {

// mid-sized batches (gemm_mid.h), compiled in gemm_mid.hip
template <typename T, typename OutT, bool NESTED, int ABL = 0>
int launch_gemm_mid(const T *x, const typename Q4ProducerRT<T, NESTED>::Params &wp, const T *bias, OutT *out, int64_t M,
                    int64_t N, int64_t K, float *ws, int64_t ws_bytes, int force_slices, hipStream_t st);
bool gemm_mid_shape(int64_t M, int64_t N, int64_t K);
bool gemm_small_shape(int64_t M, int64_t N, int64_t K, int64_t K_weight);
bool gemm_small_one_round(int64_t M, int64_t N, int64_t K, int64_t K_weight, int64_t ws_bytes);
template <typename T, typename OutT, bool NESTED>
int launch_gemm_small(const T *, const uint8_t *, const AbsmaxView &, const T *, OutT *, int64_t, int64_t, int64_t, int64_t, int, int, float *,
                      int64_t, hipStream_t);
int matmul_4bit_fused4_path(const void *, int64_t, int64_t, const uint8_t *, const AbsmaxView &, int64_t, int64_t, int, int, int, const void *, int, void *,
                            hipStream_t);
int matmul_4bit_f32_path(const void *, int64_t, int64_t, const uint8_t *, const AbsmaxView &, int64_t, int64_t, int, int, const void *, int, void *, void *, int64_t, hipStream_t);
int matmul_4bit_dense_path(const void *, int64_t, int64_t, const uint8_t *, const AbsmaxView &, int64_t, int64_t, int, int, int, const void *,
                           int, void *, void *, int64_t, hipStream_t);

// =====================================================================================
// generic kernel: wave per (n, m-chunk of MT rows); lanes stride over k in steps of 8
// =====================================================================================
// flags: bit 0 = a lane's 8 weights are one aligned dword of `packed` and share one absmax (K_weight % 8 == 0, blocksize >= 8,
// packed 4-byte aligned); bit 1 = a lane's 8 activations of a row are 16-byte loads (X 16-byte aligned, row pitch a multiple
// of 16 bytes).  Same values, same fma order per lane with and without the flags: the results do not depend on them.
template <typename T, typename OutT, int QT, bool NESTED, int MT>
__global__ __launch_bounds__(256) void k_matmul4_generic(const T *__restrict__ X, const uint8_t *__restrict__ packed,
                                                        AbsmaxView am, const T *__restrict__ bias,
                                                        OutT *__restrict__ out, int64_t M, int64_t N, int64_t K,
                                                        int64_t K_weight, int bs_shift, int flags) {
    __shared__ float lut[16];
    fill_code_lut<QT>(lut, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.y * MT;
    if (n >= N) return;
    const int64_t nblk = K_weight >> bs_shift;
    const bool wfast = flags & 1, xvec = flags & 2;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) acc[i] = 0.0f;
    for (int64_t k0 = (int64_t)lane * 8; k0 < K; k0 += 512) {
        float w[8];
        if (wfast) {
            const int64_t flat = n * K_weight + k0;
            const uint32_t pk = *reinterpret_cast<const uint32_t *>(packed + (flat >> 1));
            const float a = load_absmax<NESTED>(am, n * nblk + (k0 >> bs_shift));
#pragma unroll
            for (int j = 0; j < 8; j++)
                w[j] = (k0 + j < K) ? to_f32(from_f32<T>(lut[(pk >> (4 * j)) & 15] * a)) : 0.0f;   // weight rounded to its dtype (functional.py:382)
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t k = k0 + j;
                if (k < K) {
                    const int64_t flat = n * K_weight + k;
                    const uint8_t b = packed[flat >> 1];
                    const int idx = (flat & 1) ? (b >> 4) : (b & 15);
                    const float v = lut[idx] * load_absmax<NESTED>(am, n * nblk + (k >> bs_shift));
                    w[j] = to_f32(from_f32<T>(v));
                } else w[j] = 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int64_t m = m0 + i;
            if (m < M) {
                if (xvec && k0 + 8 <= K) {
                    __attribute__((aligned(16))) T xv[8];
                    constexpr int NV = (int)sizeof(T) * 8 / 16;
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        reinterpret_cast<u32x4 *>(xv)[v] = reinterpret_cast<const u32x4 *>(X + m * K + k0)[v];
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[i] = fmaf(to_f32(xv[j]), w[j], acc[i]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (k0 + j < K) acc[i] = fmaf(to_f32(X[m * K + k0 + j]), w[j], acc[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const float s = wave_sum(acc[i]);
        const int64_t m = m0 + i;
        if (lane == 0 && m < M) {
            float v = s + (bias ? to_f32(bias[n]) : 0.0f);
            out[m * N + n] = from_f32<OutT>(to_f32(from_f32<T>(v)));
        }
    }
}

// =====================================================================================
// Skinny GEMM (2 <= M <= 64): the weight-streaming regime with more than a handful of activation rows.
// One workgroup = 16 waves x 32 weight rows; wave w contracts the 128-k blocks w, w+16, ... of those rows with all
// M activation rows on v_mfma_f32_16x16x32, so every packed weight is fetched and decoded exactly once (as in the
// GEMV) while the M rows ride along in the MFMA's free dimension.  k is laid out so that lane quarter q of a
// 128-k block owns k in [32q, 32q + 32): its 16 packed bytes are one load, the four dwords are the A fragments of
// the block's four MFMAs, and the activation fragment of MFMA g is the 16 bytes at k = 32q + 8g.  The 16 partial
// accumulators of a workgroup are added in wave order through LDS (deterministic), then bias and one rounding.
// Needs K % 128 == 0, blocksize >= 32, 16-bit types.
// =====================================================================================
template <typename T, typename OutT, int QT, bool NESTED, int MT, int NR>
__global__ __launch_bounds__(1024) void k_skinny4(const T *__restrict__ X, const uint8_t *__restrict__ packed, AbsmaxView am,
                                                 const T *__restrict__ bias, OutT *__restrict__ out, int64_t M, int64_t N,
                                                 int64_t K, int64_t K_weight, int bs_shift) {
    constexpr int WV = 16;
    __shared__ float lut[16];
    extern __shared__ __attribute__((aligned(16))) char red_raw[];   // [WV][NR * MT][256] f32
    float *red = reinterpret_cast<float *>(red_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * (16 * NR);
    const int64_t nblk = K_weight >> bs_shift, row_bytes = K_weight >> 1;
    fill_code_lut<QT>(lut, threadIdx.x);

    const uint8_t *wrow[NR];
    int64_t arow[NR];
#pragma unroll
    for (int rg = 0; rg < NR; rg++) {
        int64_t n = n0 + 16 * rg + r16;
        n = n < N ? n : N - 1;
        wrow[rg] = packed + n * row_bytes;
        arow[rg] = n * nblk;
    }
    const T *xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int64_t m = 16 * mt + r16;
        m = m < M ? m : M - 1;
        xrow[mt] = X + m * K;
    }
    f32x4 acc[NR][MT];
#pragma unroll
    for (int rg = 0; rg < NR; rg++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[rg][mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t nb128 = K >> 7;
    // packed weights + absmax of block b + WV are requested before block b is decoded (HBM latency); the activation
    // fragments (L2-resident) are requested at the top of each trip
    u32x4 w[NR], w_n[NR];
    float a[NR], a_n[NR];
    auto request_w = [&](int64_t b) {
        const int64_t k_lane = (b << 7) + 32 * q;
#pragma unroll
        for (int rg = 0; rg < NR; rg++) {
            w_n[rg] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow[rg] + (k_lane >> 1)));
            a_n[rg] = load_absmax<NESTED>(am, arow[rg] + (k_lane >> bs_shift));
        }
    };
    if (wave < nb128) request_w(wave);
    // raw barrier for the code table (__syncthreads() would drain the requests just issued)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int64_t b = wave; b < nb128; b += WV) {
        const int64_t k_lane = (b << 7) + 32 * q;
        typename Mfma16<T>::frag xf[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                xf[mt][g] = *reinterpret_cast<const typename Mfma16<T>::frag *>(xrow[mt] + k_lane + 8 * g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rg = 0; rg < NR; rg++) {
            w[rg] = w_n[rg];
            a[rg] = a_n[rg];
        }
        if (b + WV < nb128) request_w(b + WV);
        __builtin_amdgcn_sched_barrier(0);   // every request of the block is in flight before the first use
#pragma unroll
        for (int rg = 0; rg < NR; rg++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // the reference's dequantize arithmetic: code * absmax in f32 -> RNE 16 bit (functional.py:388-416)
                const uint32_t wd = w[rg][g];
                const uint32_t wo = wd & 0xF0F0F0F0u;
                const uint32_t we = (wd << 2) & 0x3C3C3C3Cu;
                const char *lutb = reinterpret_cast<const char *>(lut);
                u32x4 fr;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float lo = *reinterpret_cast<const float *>(lutb + bfe_u32(we, 8 * j, 8)) * a[rg];
                    const float hi = *reinterpret_cast<const float *>(lutb + bfe_u32(wo, 8 * j + 2, 6)) * a[rg];
                    fr[j] = pack2<T>(lo, hi);
                }
                const auto af = __builtin_bit_cast(typename Mfma16<T>::frag, fr);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[rg][mt] = Mfma16<T>::run(af, xf[mt][g], acc[rg][mt]);
            }
    }
    // ---- reduce the WV partial 16 x 16 tiles in wave order
#pragma unroll
    for (int rg = 0; rg < NR; rg++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
            *reinterpret_cast<f32x4 *>(red + ((wave * (NR * MT) + rg * MT + mt) * 256 + lane * 4)) = acc[rg][mt];
    __syncthreads();
    for (int t = threadIdx.x; t < NR * MT * 256; t += 1024) {
        const int tile = t >> 8, e = t & 255;
        float s = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WV; wv++) s += red[(wv * (NR * MT) + tile) * 256 + e];
        const int ln = e >> 2, r = e & 3;
        const int rg = tile / MT, mt = tile % MT;
        const int64_t n = n0 + 16 * rg + 4 * (ln >> 4) + r;   // D[a][b]: a = 4 * (lane >> 4) + r (weight row), b = lane & 15
        const int64_t m = 16 * mt + (ln & 15);
        if (n < N && m < M) {
            const float v = s + (bias ? to_f32(bias[n]) : 0.0f);
            out[m * N + n] = from_f32<OutT>(to_f32(from_f32<T>(v)));
        }
    }
}

// =====================================================================================
// dispatch
// =====================================================================================
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) s++;
    return s;
}

// Number of K slices for the 128 x 128 kernel: enough workgroups for two per CU, at least four k-steps (256 k) per
// slice.  1 = no split.  Shared by the dispatcher and mbnb_matmul_4bit_workspace_bytes.
int64_t matmul4_splitk_slices(int64_t M, int64_t N, int64_t K) {
    if (M <= 4 || K % 64 != 0) return 1;
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    if (tiles256 >= 96) return 1;                      // served by the 256 x 256 kernel
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int64_t s = (512 + tiles - 1) / tiles;   // 512 workgroups: best of {256, 512, 768, 1024} at M = 128 ... 1024 (4096^2)
    const int64_t smax = K / 256;
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

template <typename T, typename OutT, int QT, bool NESTED>
static int launch_matmul4(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                          int64_t K_weight, int blocksize, const void *bias, void *out, float *ws, int64_t ws_bytes,
                          hipStream_t st) {
    const T *x = static_cast<const T *>(A);
    const T *b = static_cast<const T *>(bias);
    OutT *o = static_cast<OutT *>(out);
    constexpr bool is16 = sizeof(T) == 2;
    const bool fast_layout = is16 && blocksize >= 32 && (K_weight % 32 == 0) && (K % 8 == 0) && aligned16(A) &&
                             aligned16(packed);
    if constexpr (is16) {
        const int64_t slices = matmul4_splitk_slices(M, N, K);
        // split-K needs the caller's workspace (mbnb_matmul_4bit_ws): full 128 x 128 f32 tiles per slice
        const bool splitk = fast_layout && slices > 1 && ws != nullptr && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0) &&
                            ws_bytes >= slices * ((M + 127) / 128) * ((N + 127) / 128) * 65536;
#ifdef MBNB_ABLATION
        static const bool no_skinny = getenv("MBNB_NO_SKINNY") != nullptr;   // diagnostic builds only: A/B switch
#else
        constexpr bool no_skinny = false;
#endif
        // weight-streaming regime with a few activation rows: 2 <= M <= 32, and up to 64 for layers of <= 16 Mi weights
        // (beyond that the activation re-reads of the skinny kernel cost more than split-K's second pass)
        const bool small_ok = blocksize >= 32 && gemm_small_shape(M, N, K, K_weight);   // 32 < M <= 256: gemm_small.h
        const bool skinny = fast_layout && !no_skinny && M >= 2 && !small_ok && (M <= 32 || (M <= 64 && N * K <= ((int64_t)1 << 24))) &&
                            (K % 128 == 0);
        if (fast_layout && M <= 16 && (K % 32 == 0) && !(splitk && M > 4) && !skinny) {
            const int sh = ilog2(blocksize);
            const int64_t Kp = (K + 2047) & ~(int64_t)2047;
            const bool xlds = (int64_t)8 * Kp * 2 <= 65536;  // largest MT rows fit the default dynamic-LDS limit
#define MBNB_GEMV(MT, NR, KU)                                                                                       \
    do {                                                                                                            \
        dim3 grid((unsigned)((N + 4 * NR - 1) / (4 * NR)), (unsigned)((M + MT - 1) / MT));                          \
        if (xlds)                                                                                                   \
            hipLaunchKernelGGL((k_gemv4<T, OutT, QT, NESTED, MT, NR, KU, true>), grid, dim3(256),                   \
                               (size_t)MT * Kp * 2, st, x, packed, am, b, o, M, N, K, K_weight, sh);                \
        else                                                                                                        \
            hipLaunchKernelGGL((k_gemv4<T, OutT, QT, NESTED, MT, NR, KU, false>), grid, dim3(256), 0, st, x, packed, \
                               am, b, o, M, N, K, K_weight, sh);                                                    \
    } while (0)
            // M = 1, blocksize 64, K % 64 == 0 up to 16384 (round 3): k_gemv4_lean -- the same arithmetic with the per-wave fixed cost
            // cut (480 instead of 627 instructions per wave: buffer descriptors instead of 64-bit address VALU, table from
            // constant memory, DPP reduction, no next-trip ring); bit-equal outputs, 5.25 -> 5.08 us per rotating 4096^2 layer,
            // 4.67 -> 4.24 us on a cache-resident one (profiles/r03_gemv_lean_ab.txt)
            if constexpr (std::is_same<OutT, T>::value) {
                const int64_t ku = (K + 2047) / 2048;
                if (M == 1 && blocksize == 64 && K_weight == K && K % 64 == 0 && K >= 1024 && ku <= 8 && N * (K / 2) < ((int64_t)1 << 40) &&
                    (!NESTED || (am.bs2 > 0 && (am.bs2 & (am.bs2 - 1)) == 0 && (reinterpret_cast<uintptr_t>(am.i8) & 3) == 0 && (K / 64) % 4 == 0))) {
                    const dim3 grid((unsigned)((N + 3) / 4));
#define MBNB_LEAN(KU) hipLaunchKernelGGL((k_gemv4_lean<T, OutT, QT, NESTED, KU>), grid, dim3(256), (size_t)KU * 4096, st, x, packed, am, b, o, N, K)
                    if (ku == 1) MBNB_LEAN(1);
                    else if (ku == 2) MBNB_LEAN(2);
                    else if (ku == 3) MBNB_LEAN(3);
                    else if (ku == 4) MBNB_LEAN(4);
                    else if (ku <= 6) MBNB_LEAN(6);
                    else MBNB_LEAN(8);
#undef MBNB_LEAN
                    set_kernel_name("gemv");
                    return check_launch("matmul_4bit(gemv lean)");
                }
            }
            // M = 1: two rows per wave once there are enough rows to fill the chip twice over (4 KiB of packed
            // weights in flight per wave: +12 % streaming rate at N >= 8192, tools/gemv_sweep.py)
            if (M == 1 && N >= 8192) MBNB_GEMV(1, 2, 2);
            else if (M == 1) MBNB_GEMV(1, 1, 2);
            else if (!xlds) {           // activations in registers: keep the register-light shapes
                if (M == 2) MBNB_GEMV(2, 1, 2);
                else if (M <= 4) MBNB_GEMV(4, 2, 1);
                else MBNB_GEMV(8, 1, 1);
            } else if (M == 2) MBNB_GEMV(2, 1, 2);
            else if (M <= 4) MBNB_GEMV(4, 1, 2);
            else MBNB_GEMV(8, 1, 2);
#undef MBNB_GEMV
            set_kernel_name("gemv");
            return check_launch("matmul_4bit(gemv)");
        }
        if (skinny) {
            const int sh = ilog2(blocksize);
#define MBNB_SKINNY(MT, NR)                                                                                          \
    do {                                                                                                             \
        constexpr int lds = 16 * NR * MT * 1024;                                                                     \
        auto kern = k_skinny4<T, OutT, QT, NESTED, MT, NR>;                                                          \
        if (lds > 65536) {                                                                                           \
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(skinny)")) return rc; \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)((N + 16 * NR - 1) / (16 * NR))), dim3(1024), lds, st, x, packed, am, b, o, M, N, \
                           K, K_weight, sh);                                                                         \
    } while (0)
            // one 16-row group per workgroup (N / 16 workgroups): measured faster than two at every shape tried
            // (tools/m_sweep2.py), the extra activation traffic notwithstanding
            if (M <= 16) MBNB_SKINNY(1, 1);
            else if (M <= 32) MBNB_SKINNY(2, 1);
            else MBNB_SKINNY(4, 1);
#undef MBNB_SKINNY
            set_kernel_name("skinny_mfma16");
            return check_launch("matmul_4bit(skinny)");
        }
        if (fast_layout && small_ok) {
            // 64 < M <= 256: weights decoded from registers to registers, activations by LDS-DMA (gemm_small.h)
            const int rc = launch_gemm_small<T, OutT, NESTED>(x, packed, am, b, o, M, N, K, K_weight, QT, ilog2(blocksize), ws, ws_bytes, st);
            if (rc != MBNB_NOT_APPLICABLE) return rc;
        }
        if (fast_layout && blocksize == 64 && (K_weight % 256 == 0) && gemm_mid_shape(M, N, K)) {
            // mid-sized batches: 128 x 64 tiles on the LDS-DMA pipeline, K split over the caller's workspace when given
            bool ok = true;
            if constexpr (NESTED) ok = am.bs2 >= 4 && (am.bs2 & (am.bs2 - 1)) == 0 && (reinterpret_cast<uintptr_t>(am.i8) & 3) == 0;
            if (ok) {
                using P = Q4ProducerRT<T, NESTED>;
                typename P::Params wp{packed, am, N, K_weight, K_weight / blocksize, 6, QT, NESTED ? ilog2(am.bs2) : 0, 8, 6};
                return launch_gemm_mid<T, OutT, NESTED>(x, wp, b, o, M, N, K, ws, ws_bytes, 0, st);
            }
        }
        if (fast_layout && (K % 64 == 0) && ((M + 255) / 256) * ((N + 255) / 256) >= 96) {
            // large problems: 256 x 256 tiles, one workgroup per CU
            using P = Q4ProducerRT<T, NESTED>;
            const bool bs2_pow2 = !NESTED || (am.bs2 > 0 && (am.bs2 & (am.bs2 - 1)) == 0);
            typename P::Params wp{packed, am, N, K_weight, K_weight / blocksize, ilog2(blocksize), QT,
                                  NESTED ? ilog2(am.bs2) : 0, 8, 6};
            const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
            int od = sizeof(OutT) == 4 ? MBNB_F32 : (std::is_same<OutT, f16_t>::value ? MBNB_F16 : MBNB_BF16);
            if (bs2_pow2) {
                using KernT = void (*)(const T *, typename P::Params, const T *, void *, int, int64_t, int64_t, int64_t);
                // k_gemm256p with the byte-table decode; at blocksize 64 the absmax (plain f32, or int8 codes + absmax2)
                // is fetched once per four k-steps (AM4).  Diagnostic builds (-DMBNB_ABLATION) read A/B switches for
                // those two and for the schedule variants from the environment; the product reads no environment.
#ifdef MBNB_ABLATION
                static const bool no_blut = getenv("MBNB_NO_BLUT") != nullptr;
                static const bool no_am4 = getenv("MBNB_NO_AM4") != nullptr;
#else
                constexpr bool no_blut = false, no_am4 = false;
#endif
                KernT kern = no_blut ? k_gemm256p<T, NESTED> : k_gemm256p<T, NESTED, 0, false, true>;
                bool am4 = !no_am4 && blocksize == 64 && (K_weight % 256 == 0);
                if constexpr (NESTED) am4 = am4 && am.bs2 >= 4 && (reinterpret_cast<uintptr_t>(am.i8) & 3) == 0;
                if (am4) kern = no_blut ? k_gemm256p<T, NESTED, 0, true> : k_gemm256p<T, NESTED, 0, true, true>;
#ifdef MBNB_ABLATION
                static const bool use_p = getenv("MBNB_256P") != nullptr;   // diagnostic builds: the round-1 kernel for every shape
#else
                constexpr bool use_p = false;
#endif
                if (am4 && !use_p) {
                    // blocksize 64: k_gemm_fused4 (gemm_fused4.h: the decode inside the four-wave MFMA pipeline; the no-scratch kernel since
                    // round 4 -- 117.5 us against 120.3 for the eight-wave k_gemm256s at 4096^3, tools/exp/parked/); what it does not take
                    // (unaligned packed bytes / absmax) stays on k_gemm256p below
                    const int rc4 = matmul_4bit_fused4_path(A, M, K, packed, am, N, K_weight, blocksize, QT, std::is_same<T, f16_t>::value ? MBNB_F16 : MBNB_BF16,
                                                            bias, od, out, st);
                    if (rc4 != MBNB_NOT_APPLICABLE) return rc4;
                }
#ifdef MBNB_ABLATION
                // diagnostic builds: the measured schedule alternatives
                static const bool use_pp = getenv("MBNB_PINGPONG") != nullptr;    // ping-pong schedule
                static const bool use_valu = getenv("MBNB_VALUDEC") != nullptr;   // slot-pinned + VALU decode
                if (use_pp) kern = k_gemm256pp<T, NESTED>;
                if (use_valu) kern = (!NESTED && am4) ? k_gemm256v<T, NESTED, 0, !NESTED> : k_gemm256v<T, NESTED>;
#else
                constexpr bool use_pp = false;
#endif
#ifdef MBNB_ABLATION
                if constexpr (std::is_same<T, bf16_t>::value && !NESTED) {
                    static const int abl = getenv("MBNB_ABLATE") ? atoi(getenv("MBNB_ABLATE")) : 0;
                    switch (abl) {
#define MBNB_ABL(v) case v: kern = use_pp ? k_gemm256pp<T, NESTED, v> : k_gemm256p<T, NESTED, v>; break;
                        MBNB_ABL(1) MBNB_ABL(2) MBNB_ABL(3) MBNB_ABL(4) MBNB_ABL(8) MBNB_ABL(16) MBNB_ABL(12) MBNB_ABL(20)
                        MBNB_ABL(24) MBNB_ABL(28) MBNB_ABL(31) MBNB_ABL(7) MBNB_ABL(23) MBNB_ABL(32) MBNB_ABL(64) MBNB_ABL(128) MBNB_ABL(256) MBNB_ABL(512) MBNB_ABL(520) MBNB_ABL(535) MBNB_ABL(1024) MBNB_ABL(2048) MBNB_ABL(2056) MBNB_ABL(4096) MBNB_ABL(4608) MBNB_ABL(516) MBNB_ABL(515) MBNB_ABL(528) MBNB_ABL(532) MBNB_ABL(519) MBNB_ABL(32768) MBNB_ABL(65536) MBNB_ABL(98304)
#undef MBNB_ABL
                        default: break;
                    }
                }
#endif
                constexpr int lds = gemm256p_lds_bytes<NESTED>();
                if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(mfma256)")) return rc;
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, b, static_cast<void *>(o), od, M, N, K);
                set_kernel_name("mfma256");
                return check_launch("matmul_4bit(mfma256)");
            }
            auto kern = k_gemm256<T, P>;
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), G256_LDS, "matmul_4bit(mfma256)")) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), G256_LDS, st, x, wp, b, static_cast<void *>(o), od, M, N, K);
            set_kernel_name("mfma256");
            return check_launch("matmul_4bit(mfma256)");
        }
        if (fast_layout) {
            using P = Q4Producer<T, QT, NESTED>;
            typename P::Params wp{packed, am, N, K_weight, K_weight / blocksize, ilog2(blocksize)};
            constexpr int BM = 128, BN = 128;
            const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
            constexpr int lds = gemm_decode_lds_bytes<BM, BN>();
            auto kern = k_gemm_decode<T, OutT, P, BM, BN>;
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_4bit(mfma128)")) return rc;
            if (splitk) {
                // k slices of whole 64-k steps, the last one takes the remainder
                int64_t kps = (((K / 64) + slices - 1) / slices) * 64;
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)slices), dim3(256), lds, st, x, wp, b, o, M, N, K, ws, kps);
                int rc = check_launch("matmul_4bit(mfma128 split-K)");
                if (rc) return rc;
                hipLaunchKernelGGL((k_splitk_reduce<T, OutT>), dim3((unsigned)(tiles * 16)), dim3(256), 0, st, ws, (int)slices,
                                   b, o, M, N, (M + BM - 1) / BM, tiles);
                set_kernel_name("mfma128_splitk");
                return check_launch("matmul_4bit(split-K reduce)");
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wp, b, o, M, N, K, static_cast<float *>(nullptr), (int64_t)0);
            set_kernel_name("mfma128");
            return check_launch("matmul_4bit(mfma128)");
        }
    }
    {
        const unsigned gx = (unsigned)((N + 3) / 4);
        const int bs_shift = ilog2(blocksize);
        const int flags = ((K_weight % 8 == 0 && blocksize >= 8 && (reinterpret_cast<uintptr_t>(packed) & 3) == 0) ? 1 : 0) |
                          (((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (K * (int64_t)sizeof(T)) % 16 == 0) ? 2 : 0);
        if (M == 1)
            hipLaunchKernelGGL((k_matmul4_generic<T, OutT, QT, NESTED, 1>), dim3(gx, 1), dim3(256), 0, st, x,
                               packed, am, b, o, M, N, K, K_weight, bs_shift, flags);
        else if (M <= 4)   // the packed weight once for all rows
            hipLaunchKernelGGL((k_matmul4_generic<T, OutT, QT, NESTED, 4>), dim3(gx, 1), dim3(256), 0, st, x,
                               packed, am, b, o, M, N, K, K_weight, bs_shift, flags);
        else
            hipLaunchKernelGGL((k_matmul4_generic<T, OutT, QT, NESTED, 8>), dim3(gx, (unsigned)((M + 7) / 8)),
                               dim3(256), 0, st, x, packed, am, b, o, M, N, K, K_weight, bs_shift, flags);
        set_kernel_name("generic");
        return check_launch("matmul_4bit(generic)");
    }
}

template <typename T, typename OutT>
static int matmul4_qt(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                      int64_t K_weight, int blocksize, int qt, const void *bias, void *out, float *ws, int64_t ws_bytes,
                      hipStream_t st) {
    const bool nested = am.i8 != nullptr;
    if (qt == MBNB_NF4)
        return nested ? launch_matmul4<T, OutT, MBNB_NF4, true>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, ws, ws_bytes, st)
                      : launch_matmul4<T, OutT, MBNB_NF4, false>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, ws, ws_bytes, st);
    return nested ? launch_matmul4<T, OutT, MBNB_FP4, true>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, ws, ws_bytes, st)
                  : launch_matmul4<T, OutT, MBNB_FP4, false>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, ws, ws_bytes, st);
}

template <typename T>
static int matmul4_out(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                       int64_t K_weight, int blocksize, int qt, const void *bias, int out_dtype, void *out, float *ws,
                       int64_t ws_bytes, hipStream_t st) {
    switch (out_dtype) {
        case MBNB_F16: return matmul4_qt<T, f16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, ws, ws_bytes, st);
        case MBNB_BF16: return matmul4_qt<T, bf16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, ws, ws_bytes, st);
        default: return matmul4_qt<T, float>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, ws, ws_bytes, st);
    }
}

#ifdef MBNB_ABLATION
extern "C" int mbnb_debug_read_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg_stamps), sizeof(unsigned long long) * 2 * 1024);
}
#endif

int matmul_4bit_dispatch(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                         int64_t K_weight, int blocksize, int qt, int w_dtype, const void *bias, int out_dtype,
                         void *out, void *workspace, int64_t ws_bytes, int flags, hipStream_t st) {
    const bool fused_only = (flags & MBNB_MATMUL_FUSED_ONLY) != 0;
    // 256 < M <= 512 rows that k_gemm_small serves in one round of workgroups stay fused (gemm_small.hip: gemm_small_one_round); the
    // conditions of that branch of launch_matmul4 are repeated here so that nothing else is kept away from the decode-once path
    const bool small_first = (w_dtype == MBNB_F16 || w_dtype == MBNB_BF16) && blocksize >= 32 && (K_weight % 32 == 0) && (K % 8 == 0) &&
                             aligned16(A) && aligned16(packed) && (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) &&
                             gemm_small_one_round(M, N, K, K_weight, workspace ? ws_bytes : 0);
    // large M: decode the weight once into the workspace, then the dense MFMA GEMM (gemm_dense.hip)
    if (!fused_only && !small_first) {
        const int rc = matmul_4bit_dense_path(A, M, K, packed, am, N, K_weight, blocksize, qt, w_dtype, bias, out_dtype, out, workspace,
                                              ws_bytes, st);
        if (rc != MBNB_NOT_APPLICABLE) return rc;
        if (w_dtype == MBNB_F32) {   // f32 weights: decode once + f32 MFMA GEMM (gemm_f32.hip)
            const int rf = matmul_4bit_f32_path(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, workspace, ws_bytes, st);
            if (rf != MBNB_NOT_APPLICABLE) return rf;
        }
    }
    float *ws = static_cast<float *>(workspace);
    switch (w_dtype) {
        case MBNB_F16: return matmul4_out<f16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, ws, ws_bytes, st);
        case MBNB_BF16: return matmul4_out<bf16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, ws, ws_bytes, st);
        default: return matmul4_out<float>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, ws, ws_bytes, st);
    }
}

}  // namespace mbnb
