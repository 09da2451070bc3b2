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

namespace mbnb {

// =====================================================================================
// generic kernel: wave per (n, m-chunk of MT rows); lanes stride over k in steps of 8
// =====================================================================================
template <typename T, typename OutT, int QT, bool NESTED, int MT>
__global__ __launch_bounds__(256) void k_matmul4_generic(const T *__restrict__ X, const uint8_t *__restrict__ packed,
                                                        AbsmaxView am, const T *__restrict__ bias,
                                                        OutT *__restrict__ out, int64_t M, int64_t N, int64_t K,
                                                        int64_t K_weight, int blocksize) {
    __shared__ float lut[16];
    fill_code_lut<QT>(lut, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.y * MT;
    if (n >= N) return;
    const int64_t nblk = K_weight / blocksize;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) acc[i] = 0.0f;
    for (int64_t k0 = (int64_t)lane * 8; k0 < K; k0 += 512) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int64_t k = k0 + j;
            if (k < K) {
                const int64_t flat = n * K_weight + k;
                const uint8_t b = packed[flat >> 1];
                const int idx = (flat & 1) ? (b >> 4) : (b & 15);
                const float v = lut[idx] * load_absmax<NESTED>(am, n * nblk + k / blocksize);
                w[j] = to_f32(from_f32<T>(v));  // weight rounded to its dtype (functional.py:382)
            } else w[j] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int64_t m = m0 + i;
            if (m < M) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (k0 + j < K) acc[i] = fmaf(to_f32(X[m * K + k0 + j]), w[j], acc[i]);
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
// GEMV (M <= 16): HBM-bound.  A wave owns NR consecutive weight rows; per k-step each lane
// loads 16 B of packed nibbles (32 k) per row straight to VGPRs -- all NR loads of a step are
// issued before the first is consumed -- plus the matching 64 B of each activation row (L1/L2
// resident, shared by the NR rows).  Decode = LDS table lookup * absmax -> 16-bit (the exact
// reference weight bits), contraction = v_dot2 into f32.
// =====================================================================================
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}
template <typename T> struct Dot2;
template <> struct Dot2<f16_t> {
    static __device__ __forceinline__ float run(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    }
};
template <> struct Dot2<bf16_t> {
    static __device__ __forceinline__ float run(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
    }
};

// KU k-steps of 2048 (64 lanes x 32 k) are processed per loop trip with ALL their loads issued
// before the first use; out-of-range chunks load from a clamped address with a zeroed absmax
// (no branch or select between a load and its use: that makes hipcc wait vmcnt(0) per load).
template <typename T, typename OutT, int QT, bool NESTED, int MT, int NR, int KU, bool XLDS>
__global__ __launch_bounds__(256) void k_gemv4(const T *__restrict__ X, const uint8_t *__restrict__ packed, AbsmaxView am,
                                              const T *__restrict__ bias, OutT *__restrict__ out, int64_t M, int64_t N,
                                              int64_t K, int64_t K_weight, int bs_shift) {
    __shared__ float lut[16];
    // XLDS: the MT activation rows are staged once per workgroup in LDS (MT*K*2 bytes) and shared by
    // its 4 waves; otherwise every wave re-reads them from L2, which at M = 1 doubles the bytes moved
    // through the CU's vector-memory path (the real bound of this kernel, not HBM)
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int lane = threadIdx.x & 63;
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
    const int64_t m0 = (int64_t)blockIdx.y * MT;
    const int64_t nblk = K_weight >> bs_shift;
    const int64_t row_bytes = K_weight >> 1;

    float acc[NR][MT];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int i = 0; i < MT; i++) acc[r][i] = 0.0f;

    const uint8_t *wrow[NR];
    int64_t arow[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int64_t n = (n0 + r < N) ? n0 + r : N - 1;
        wrow[r] = packed + n * row_bytes;
        arow[r] = n * nblk;
    }
    const T *xrow[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) xrow[i] = X + ((m0 + i < M) ? m0 + i : M - 1) * K;

    // The packed weights and absmax of trip t+1 are requested before trip t is decoded, and those of trip 0
    // before the prologue (code table, activation staging, barrier): HBM latency overlaps the prologue.
    // K % 32 == 0 here, so a lane's 32-k chunk is entirely inside or outside [0, K).  Outside: load from
    // k = 0 (in bounds) and zero the absmax, so the products vanish -- no branch, no select between a load
    // and its use.
    u32x4 wq[KU][NR], wq_n[KU][NR];
    float a[KU][NR], a_n[KU][NR];
    float vf[KU], vf_n[KU];
    auto request_w = [&](int64_t kbase) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int64_t k0 = kbase + u * 2048 + lane * 32;
            const bool wvalid = k0 < K;
            const int64_t kc = wvalid ? k0 : 0;
            vf_n[u] = wvalid ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                wq_n[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow[r] + (kc >> 1)));
                a_n[u][r] = load_absmax<NESTED>(am, arow[r] + (kc >> bs_shift));
            }
        }
    };
    // activations -> LDS by LDS-DMA (no registers, no wait before the weight requests); rows are padded to
    // Kp = K rounded up to 2048 so that whole-wave 1 KiB pieces never run past the allocation
    const int64_t Kp = (K + 2047) & ~(int64_t)2047;
    if constexpr (XLDS) {
#pragma unroll
        for (int i = 0; i < MT; i++)
            for (int64_t kb = 0; kb < K; kb += 2048) {
                const int64_t k = kb + (int64_t)threadIdx.x * 8;
                auto g = (const __attribute__((address_space(1))) void *)(xrow[i] + (k < K ? k : 0));
                auto l = (__attribute__((address_space(3))) void *)(xs + (i * Kp + kb) * 2 + (threadIdx.x >> 6) * 1024);
                __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    request_w(0);
    __builtin_amdgcn_sched_barrier(0);
    fill_code_lut<QT>(lut, threadIdx.x);
    if constexpr (XLDS) {
        // vmcnt is in order: everything older than the KU*NR weight (+ absmax) requests has landed
        constexpr int NW = KU * NR * (NESTED ? 3 : 2);
        static_assert(NW <= 63, "vmcnt range");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
    }
    // raw barrier: __syncthreads() would also wait for the weight requests (its fence drains vmcnt)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int64_t kbase = 0; kbase < K; kbase += 2048 * KU) {
        u32x4 xv[KU][MT][4];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            vf[u] = vf_n[u];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                wq[u][r] = wq_n[u][r];
                a[u][r] = a_n[u][r];
            }
        }
        if (kbase + 2048 * KU < K) request_w(kbase + 2048 * KU);
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int64_t k0 = kbase + u * 2048 + lane * 32;
            const int64_t kc = k0 < K ? k0 : 0;
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    if constexpr (XLDS) xv[u][i][c] = *reinterpret_cast<const u32x4 *>(xs + (i * Kp + kc + 8 * c) * 2);
                    else xv[u][i][c] = *reinterpret_cast<const u32x4 *>(xrow[i] + kc + 8 * c);
                }
        }
        // keep every load above issued before anything waits on one of them
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KU; u++)
#pragma unroll
            for (int r = 0; r < NR; r++) a[u][r] *= vf[u];
        // Decode, software-pipelined over the KU*NR*4 quarters (8 k each): the table lookups of quarter q+1
        // are issued before quarter q is multiplied, so their LDS latency is not waited for in place.
        constexpr int NQ = KU * NR * 4;
        float L[2][8];
        auto lookup = [&](auto qq, float (&Lq)[8]) {
            constexpr int q = decltype(qq)::value;
            constexpr int u = q / (NR * 4), r = (q / 4) % NR, c = q % 4;
            const uint32_t w = wq[u][r][c];
            {
                // byte offsets 4*idx into the code table with one v_bfe_u32 per nibble (see gemm256.h)
                const uint32_t wo = w & 0xF0F0F0F0u;
                const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
                const char *lutb = reinterpret_cast<const char *>(lut);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    Lq[2 * j] = *reinterpret_cast<const float *>(lutb + bfe_u32(we, 8 * j, 8));
                    Lq[2 * j + 1] = *reinterpret_cast<const float *>(lutb + bfe_u32(wo, 8 * j + 2, 6));
                }
            }
        };
        auto consume = [&](auto qq, const float (&Lq)[8]) {
            constexpr int q = decltype(qq)::value;
            constexpr int u = q / (NR * 4), r = (q / 4) % NR, c = q % 4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 pr = f32x2{Lq[2 * j], Lq[2 * j + 1]} * f32x2{a[u][r], a[u][r]};  // two IEEE f32 products
                const uint32_t wp = pack2<T>(pr[0], pr[1]);
#pragma unroll
                for (int i = 0; i < MT; i++) acc[r][i] = Dot2<T>::run(wp, xv[u][i][c][j], acc[r][i]);
            }
        };
        lookup(std::integral_constant<int, 0>{}, L[0]);
        static_for<NQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            if constexpr (q + 1 < NQ) lookup(std::integral_constant<int, q + 1>{}, L[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            consume(qq, L[q & 1]);
        });
    }
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const float s = wave_sum(acc[r][i]);
            const int64_t n = n0 + r, m = m0 + i;
            if (lane == 0 && n < N && m < M) {
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

template <typename T, typename OutT, int QT, bool NESTED>
static int launch_matmul4(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                          int64_t K_weight, int blocksize, const void *bias, void *out, hipStream_t st) {
    const T *x = static_cast<const T *>(A);
    const T *b = static_cast<const T *>(bias);
    OutT *o = static_cast<OutT *>(out);
    constexpr bool is16 = sizeof(T) == 2;
    const bool fast_layout = is16 && blocksize >= 32 && (K_weight % 32 == 0) && (K % 8 == 0) && aligned16(A) &&
                             aligned16(packed);
    if constexpr (is16) {
        if (fast_layout && M <= 16 && (K % 32 == 0)) {
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
            // M = 1: two rows per wave once there are enough rows to fill the chip twice over (4 KiB of packed
            // weights in flight per wave: +12 % streaming rate at N >= 8192, tools/gemv_sweep.py)
            if (M == 1 && N >= 8192) MBNB_GEMV(1, 2, 2);
            else if (M == 1) MBNB_GEMV(1, 1, 2);
            else if (M == 2) MBNB_GEMV(2, 1, 2);
            else if (M <= 4) MBNB_GEMV(4, 2, 1);
            else MBNB_GEMV(8, 1, 1);
#undef MBNB_GEMV
            set_kernel_name("gemv");
            return check_launch("matmul_4bit(gemv)");
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
                static const bool use_pp = getenv("MBNB_PINGPONG") != nullptr;  // debug A/B switch (default: lockstep schedule, faster as measured)
                using KernT = void (*)(const T *, typename P::Params, const T *, void *, int, int64_t, int64_t, int64_t);
                                static const bool use_valu = getenv("MBNB_VALUDEC") != nullptr;  // debug A/B switch: slot-pinned + VALU decode
                // production: k_gemm256p with the byte-table decode; plain f32 absmax at blocksize 64 additionally
                // fetches absmax once per four k-steps (AM4).  The other variants are kept as measured alternatives.
                static const bool no_blut = getenv("MBNB_NO_BLUT") != nullptr;  // debug A/B switch: 16-entry table decode
                KernT kern = use_valu ? k_gemm256v<T, NESTED>
                                      : (use_pp ? k_gemm256pp<T, NESTED>
                                                : (no_blut ? k_gemm256p<T, NESTED> : k_gemm256p<T, NESTED, 0, false, true>));
                static const bool no_am4 = getenv("MBNB_NO_AM4") != nullptr;  // debug A/B switch
                if constexpr (!NESTED) {
                    if (!no_am4 && !use_pp && blocksize == 64 && (K_weight % 256 == 0)) {
                        if (use_valu) kern = k_gemm256v<T, false, 0, true>;
                        else kern = no_blut ? k_gemm256p<T, false, 0, true> : k_gemm256p<T, false, 0, true, true>;
                    }
                } else {
                    // double-quantised absmax: four int8 codes (one aligned dword) + their shared absmax2 per four k-steps
                    if (!no_am4 && !use_pp && !use_valu && blocksize == 64 && (K_weight % 256 == 0) && am.bs2 >= 4 &&
                        (reinterpret_cast<uintptr_t>(am.i8) & 3) == 0)
                        kern = no_blut ? k_gemm256p<T, true, 0, true> : k_gemm256p<T, true, 0, true, true>;
                }
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
                {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    if (e != hipSuccess) {
                        set_error("matmul_4bit: hipFuncSetAttribute(256p) failed: %s", hipGetErrorString(e));
                        return (int)e;
                    }
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, x, wp, b, static_cast<void *>(o), od, M, N, K);
                set_kernel_name("mfma256");
                return check_launch("matmul_4bit(mfma256)");
            }
            auto kern = k_gemm256<T, P>;
            static bool attr_done256 = false;
            if (!attr_done256) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS);
                if (e != hipSuccess) {
                    set_error("matmul_4bit: hipFuncSetAttribute(256) failed: %s", hipGetErrorString(e));
                    return (int)e;
                }
                attr_done256 = true;
            }
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
            static bool attr_done = false;  // benign race: idempotent
            if (!attr_done) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) {
                    set_error("matmul_4bit: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                    return (int)e;
                }
                attr_done = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wp, b, o, M, N, K);
            set_kernel_name("mfma128");
            return check_launch("matmul_4bit(mfma128)");
        }
    }
    {
        const unsigned gx = (unsigned)((N + 3) / 4);
        if (M <= 4)
            hipLaunchKernelGGL((k_matmul4_generic<T, OutT, QT, NESTED, 1>), dim3(gx, (unsigned)M), dim3(256), 0, st, x,
                               packed, am, b, o, M, N, K, K_weight, blocksize);
        else
            hipLaunchKernelGGL((k_matmul4_generic<T, OutT, QT, NESTED, 8>), dim3(gx, (unsigned)((M + 7) / 8)),
                               dim3(256), 0, st, x, packed, am, b, o, M, N, K, K_weight, blocksize);
        set_kernel_name("generic");
        return check_launch("matmul_4bit(generic)");
    }
}

template <typename T, typename OutT>
static int matmul4_qt(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                      int64_t K_weight, int blocksize, int qt, const void *bias, void *out, hipStream_t st) {
    const bool nested = am.i8 != nullptr;
    if (qt == MBNB_NF4)
        return nested ? launch_matmul4<T, OutT, MBNB_NF4, true>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, st)
                      : launch_matmul4<T, OutT, MBNB_NF4, false>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, st);
    return nested ? launch_matmul4<T, OutT, MBNB_FP4, true>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, st)
                  : launch_matmul4<T, OutT, MBNB_FP4, false>(A, M, K, packed, am, N, K_weight, blocksize, bias, out, st);
}

template <typename T>
static int matmul4_out(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                       int64_t K_weight, int blocksize, int qt, const void *bias, int out_dtype, void *out,
                       hipStream_t st) {
    switch (out_dtype) {
        case MBNB_F16: return matmul4_qt<T, f16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, st);
        case MBNB_BF16: return matmul4_qt<T, bf16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, st);
        default: return matmul4_qt<T, float>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out, st);
    }
}

#ifdef MBNB_ABLATION
extern "C" int mbnb_debug_read_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg_stamps), sizeof(unsigned long long) * 2 * 1024);
}
#endif

int matmul_4bit_dispatch(const void *A, int64_t M, int64_t K, const uint8_t *packed, const AbsmaxView &am, int64_t N,
                         int64_t K_weight, int blocksize, int qt, int w_dtype, const void *bias, int out_dtype,
                         void *out, hipStream_t st) {
    switch (w_dtype) {
        case MBNB_F16: return matmul4_out<f16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, st);
        case MBNB_BF16: return matmul4_out<bf16_t>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, st);
        default: return matmul4_out<float>(A, M, K, packed, am, N, K_weight, blocksize, qt, bias, out_dtype, out, st);
    }
}

}  // namespace mbnb
