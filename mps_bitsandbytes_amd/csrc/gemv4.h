// gemv4.h — k_gemv4: fused 4-bit dequant + GEMV for M = 1 (and M <= 16 when the skinny MFMA kernel does not apply).
#pragma once

#include <type_traits>
#include <utility>

#include "gemm256.h"

namespace mbnb {

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
// DEC (decode flavour): 0 = 16-entry f32 table, one ds_read_b32 per nibble (v_bfe offsets), v_pk_mul_f32;
// 1 = 256-entry byte table of (code[b & 15], code[b >> 4]) pairs, one ds_read_b64 per packed byte, v_pk_mul_f32;
// 2 = the byte table with two scalar v_mul_f32.  The table is copied from constant memory (no 16-way select chain).
__device__ const float g_nf4_tab[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                        -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                        0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                                        0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                                        0.7229568362236023f, 1.0f};
__device__ const float g_fp4_tab[16] = {0.0f, 0.0625f, 0.125f, 0.25f, 0.375f, 0.5f, 0.75f, 1.0f,
                                        -0.0f, -0.0625f, -0.125f, -0.25f, -0.375f, -0.5f, -0.75f, -1.0f};

template <typename T, typename OutT, int QT, bool NESTED, int MT, int NR, int KU, bool XLDS, int DEC = 0>
__global__ __launch_bounds__(256) void k_gemv4(const T *__restrict__ X, const uint8_t *__restrict__ packed, AbsmaxView am,
                                              const T *__restrict__ bias, OutT *__restrict__ out, int64_t M, int64_t N,
                                              int64_t K, int64_t K_weight, int bs_shift) {
    __shared__ __attribute__((aligned(2048))) float lut[DEC == 0 ? 16 : 512];
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
    if constexpr (DEC == 0) {
        fill_code_lut<QT>(lut, threadIdx.x);
    } else {
        const float *tab = QT == MBNB_NF4 ? g_nf4_tab : g_fp4_tab;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = threadIdx.x + 256 * h, b = e >> 1;
            lut[e] = tab[(e & 1) ? (b >> 4) : (b & 15)];
        }
    }
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
        // activations: from LDS they are fetched quarter by quarter beside the table lookups (xq, two quarters in
        // flight) so that MT rows x KU chunks never sit in registers at once; without LDS staging (K too large)
        // all of a trip's activations are requested up front (xv)
        u32x4 xv[XLDS ? 1 : KU][XLDS ? 1 : MT][4];
        u32x4 xq[2][MT];
        int64_t kcs[KU];
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
            kcs[u] = kc;
            if constexpr (!XLDS) {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int i = 0; i < MT; i++) xv[u][i][c] = *reinterpret_cast<const u32x4 *>(xrow[i] + kc + 8 * c);
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
            if constexpr (XLDS) {
#pragma unroll
                for (int i = 0; i < MT; i++)
                    xq[q & 1][i] = *reinterpret_cast<const u32x4 *>(xs + (i * Kp + kcs[u] + 8 * c) * 2);
            }
            if constexpr (DEC == 0) {
                // byte offsets 4*idx into the code table with one v_bfe_u32 per nibble (see gemm256.h)
                const uint32_t wo = w & 0xF0F0F0F0u;
                const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
                const char *lutb = reinterpret_cast<const char *>(lut);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    Lq[2 * j] = *reinterpret_cast<const float *>(lutb + bfe_u32(we, 8 * j, 8));
                    Lq[2 * j + 1] = *reinterpret_cast<const float *>(lutb + bfe_u32(wo, 8 * j + 2, 6));
                }
            } else {
                const char *lutb = reinterpret_cast<const char *>(lut);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(lutb + (((w >> (8 * j)) & 0xFFu) << 3));
                    Lq[2 * j] = v[0];
                    Lq[2 * j + 1] = v[1];
                }
            }
        };
        auto consume = [&](auto qq, const float (&Lq)[8]) {
            constexpr int q = decltype(qq)::value;
            constexpr int u = q / (NR * 4), r = (q / 4) % NR, c = q % 4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t wp;
                if constexpr (DEC == 2) {
                    float p0, p1;
                    asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(Lq[2 * j]), "v"(a[u][r]));
                    asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(Lq[2 * j + 1]), "v"(a[u][r]));
                    wp = pack2<T>(p0, p1);
                } else {
                    const f32x2 pr = f32x2{Lq[2 * j], Lq[2 * j + 1]} * f32x2{a[u][r], a[u][r]};  // two IEEE f32 products
                    wp = pack2<T>(pr[0], pr[1]);
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    if constexpr (XLDS) acc[r][i] = Dot2<T>::run(wp, xq[q & 1][i][j], acc[r][i]);
                    else acc[r][i] = Dot2<T>::run(wp, xv[u][i][c][j], acc[r][i]);
                }
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

}  // namespace mbnb
