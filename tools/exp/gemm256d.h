// gemm256d.h — k_gemm256d: 256 x 256 x 64 MFMA GEMM on an ALREADY DEQUANTISED weight (experiment: "decode once, then a
// bare GEMM" against the fused k_gemm256s, whose every workgroup re-decodes its weight tile for its own 256 rows).
// Same tile -> workgroup map, LDS images, swizzle, fragment order and epilogue as k_gemm256s; both operands arrive by LDS-DMA
// (8 pieces of 8 rows x 128 B per wave per k-step), no decode, no raw / absmax slots.  out = X [M, K] * Wd [N, K]^T.
// MFMA order per accumulator is that of k_gemm256s, and Wd holds the bits dequantize_4bit produces, so the output is bit-equal.
#pragma once

#include "../../mps_bitsandbytes_amd/csrc/gemm256.h"

namespace mbnb {

constexpr int gemm256d_lds_bytes() { return 8 * 64 * 264; }  // 4 image stages (128 KiB) < the epilogue's store staging

template <typename T, int VAR>
__global__ __launch_bounds__(512, 2) void k_gemm256d(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                     void *__restrict__ out_v, int out_dtype, int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

    const int64_t tiles_m = (M + 255) >> 8, tiles_n = (N + 255) >> 8;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    int64_t tm, tn;
    if ((tiles_m % 4 == 0) && (tiles_n % 8 == 0)) {
        const int64_t patch = bid >> 5, within = bid & 31;
        const int64_t patches_m = tiles_m >> 2;
        tm = (patch % patches_m) * 4 + (within & 3);
        tn = (patch / patches_m) * 8 + (within >> 2);
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
    const int64_t m0 = tm << 8, n0 = tn << 8;

    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    auto dma16 = [&](const void *g, int off) {
        lds_dma<16>(g, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off)));
    };
    const T *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row, n = n0 + row;
        m = m < M ? m : M - 1;
        n = n < N ? n : N - 1;
        a_src[i] = X + m * K + 8 * c;
        b_src[i] = Wd + n * K + 8 * c;
    }
    // BUF form: buffer_load ... offen lds -- per-lane 32-bit offsets that never change, the k position in one SGPR
    // (no VALU per piece); rows past M / N read as zeros through the descriptor's range check instead of a clamp
    constexpr bool BUF = (VAR & 32) != 0;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
    int a_voff[4], b_voff[4];
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K), pb = reinterpret_cast<uint64_t>(Wd + n0 * K);
        const int64_t ra = (M - m0 < 256 ? M - m0 : 256) * K * 2, rb = (N - n0 < 256 ? N - n0 : 256) * K * 2;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)ra, 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)rb, 0x00020000};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            rs_a[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            rs_b[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = 8 * (wave * 4 + i) + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            a_voff[i] = b_voff[i] = (int)(row * K * 2 + 16 * c);
        }
    }
    auto dma16b = [&](int voff, const i32x4_t &rs, int soff, int off) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     ::"s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off))), "v"(voff), "s"(rs),
                       "s"(soff)
                     : "memory", "m0");
    };
    auto issue_a = [&](int stage, int64_t k0, int first, int count) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            if constexpr (BUF) dma16b(a_voff[i], rs_a, (int)(k0 * 2), P_A + stage * P_IMG + (wave * 4 + i) * 1024);
            else dma16(a_src[i] + k0, P_A + stage * P_IMG + (wave * 4 + i) * 1024);
        }
    };
    auto issue_b = [&](int stage, int64_t k0, int first, int count) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            if constexpr (BUF) dma16b(b_voff[i], rs_b, (int)(k0 * 2), P_B + stage * P_IMG + (wave * 4 + i) * 1024);
            else dma16(b_src[i] + k0, P_B + stage * P_IMG + (wave * 4 + i) * 1024);
        }
    };

    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 64 * ROW_BYTES + f;
    }
    auto read_frags = [&](int stage, int s, Frag (&wf)[4], Frag (&xf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const Frag *>(smem + fw[s] + stage * P_IMG + i * 32 * ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 2; j++) xf[j] = *reinterpret_cast<const Frag *>(smem + fx[s] + stage * P_IMG + j * 32 * ROW_BYTES);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    auto mfma_group = [&](const Frag (&wf)[4], const Frag (&xf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[i], xf[j], acc[i][j]);
    };

    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    // prologue: tile 0 -> stage 0, tile 1 -> stage 1
    issue_a(0, 0, 0, 4);
    issue_b(0, 0, 0, 4);
    issue_a(1, kclamp(1), 0, 4);
    issue_b(1, kclamp(1), 0, 4);
    MBNB_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    // k-step j (stage C holds tile j): G0 G1 G2 | barrier j | G3.  Tile j+2 goes to stage C after barrier j; its 8 DMA
    // pieces are issued over G3(j) .. G1(j+1) and must have landed by barrier j+1.
    //   VAR 0: G3 a0 a1 b0 b1 | G0 a2 b2 | G1 a3 b3 | G2 -    -> vmcnt(0) before barrier
    //   VAR 1: G3 all eight
    constexpr bool ALL3 = (VAR & 1) != 0, NODMA = (VAR & 2) != 0, NOBAR = (VAR & 4) != 0, HALFRD = (VAR & 8) != 0, HALFMM = (VAR & 16) != 0;
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        // ---- group 0
        read_frags(C, 1, wfB, xfB);
        mfma_group(wfA, xfA);
        if constexpr (!ALL3 && !NODMA) {
            if (j > 0) {
                issue_a(Nn, kclamp(j + 1), 2, 1);
                issue_b(Nn, kclamp(j + 1), 2, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1
        if constexpr (!HALFRD) read_frags(C, 2, wfA, xfA);
        if constexpr (!HALFMM) mfma_group(wfB, xfB);
        if constexpr (!ALL3 && !NODMA) {
            if (j > 0) {
                issue_a(Nn, kclamp(j + 1), 3, 1);
                issue_b(Nn, kclamp(j + 1), 3, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 2
        read_frags(C, 3, wfB, xfB);
        mfma_group(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        MBNB_VMCNT(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 3
        if constexpr (!HALFRD) read_frags(Nn, 0, wfA, xfA);
        if constexpr (!HALFMM) mfma_group(wfB, xfB);
        if constexpr (NODMA) {
        } else if constexpr (!ALL3) {
            issue_a(C, kclamp(j + 2), 0, 2);
            issue_b(C, kclamp(j + 2), 0, 2);
        } else {
            issue_a(C, kclamp(j + 2), 0, 4);
            issue_b(C, kclamp(j + 2), 0, 4);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (out_dtype != MBNB_F32) {
        char *wave_lds = smem + wave * 64 * 264;
        if (out_dtype == MBNB_F16)
            epilogue_staged<T, f16_t>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 64, n0 + wn * 128, lane);
        else
            epilogue_staged<T, bf16_t>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 64, n0 + wn * 128, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t nn = n0 + wn * 128 + i * 32 + 8 * g + 4 * fh;
                if (m >= M || nn >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[i][j][4 * g + e];
                    if (bias != nullptr && nn + e < N) s += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(s));
                }
                store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
            }
        }
}

}  // namespace mbnb
