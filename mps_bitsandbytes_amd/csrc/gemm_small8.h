// gemm_small8.h — k_gemm_small8: Linear8bit.forward / LinearFP8.forward (W8A16) for FEW activation rows (16/32 < M <= 256): the
// structure of k_gemm_small (gemm_small.h) with 8-bit weights.  A workgroup = 64 weight rows x one K slice x up to 128
// activation rows; wave w owns 16 weight rows and all activation rows; a lane decodes exactly the v_mfma_f32_16x16x32 operand
// it needs, registers to registers: sign-extend (or E4M3 decode) -> * scale / 127 (or * scale) in f32 -> RNE 16 bit, the bits
// dequantize_rowwise / dequantize_fp8_e4m3 produce.  k order inside a 256-k step: lane (row r, kc) loads the 16 bytes at
// k = 64 j + 16 kc .. + 15 (j = 0..3) and uses bytes 8 h .. 8 h + 7 as the operand of slice 2 j + h; the activation operand of that
// slice is chunk 8 j + 2 kc + h of the LDS row (swizzle: chunk ^ row on the low four bits -- the two k-chunk halves of a
// ds_read_b128 lane group differ in chunk bit 1 here, and r -> r maps rows {0-3, 12-15} and {4-11} onto complementary sets).
// One scale per weight row, no table: the decode is VALU only.  Everything else (LDS-DMA of the activations, whole-slice
// weights in registers, split-K partials + k_splitk_reduce_rm) as gemm_small.h.
#pragma once
#include "gemm_small.h"

namespace mbnb {

template <typename T, int WF, int MF>
__global__ __launch_bounds__(256, 1) void k_gemm_small8(const T *__restrict__ X, const uint8_t *__restrict__ W, const float *__restrict__ scales,
                                                        const T *__restrict__ bias, T *__restrict__ out, float *__restrict__ partial,
                                                        int64_t M, int64_t N, int64_t K, int64_t k_per_slice) {
    using Frag = typename Mfma16<T>::frag;
    constexpr int ROWS = 16 * MF, STAGE = ROWS * 512, NPW = ROWS / 8, MAXS = 8;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kc = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 64, m0 = (int64_t)blockIdx.z * ROWS;
    const int slice = blockIdx.y;
    const int64_t k_begin = (int64_t)slice * k_per_slice;
    const int64_t k_len = K - k_begin < k_per_slice ? K - k_begin : k_per_slice;
    const int nsteps = (int)(k_len >> 8);

    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K + k_begin);
        const int64_t rows_a = M - m0 < ROWS ? M - m0 : ROWS;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
#pragma unroll
        for (int e = 0; e < 4; e++) rs_a[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
    }
    int voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int row = 2 * (NPW * wave + i) + (lane >> 5), pos = lane & 31;
        voff[i] = (int)(row * K * 2) + 16 * ((pos & 16) | ((pos & 15) ^ (row & 15))) - GD_M0_GROUP * (i & 3) * 1024;
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    const uint32_t lds_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(wave * NPW * 1024)));
    auto issue_piece = [&](auto ii, int stage, int soff) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value;
        const uint32_t dst = lds_wave + (uint32_t)(stage * STAGE + i * 1024);
        const int vo = voff[i];
        const i32x4_t rs = rs_a;
        // four pieces share ONE M0 write: the instruction offset (added to the LDS address and to the global address alike) carries the
        // piece inside the group, the per-lane offsets are that much smaller (gemm_dense.h, GD_M0_GROUP; K >= 512 here)
        if constexpr (GD_M0_GROUP && (i & 3) != 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rs), "s"(soff), "n"((i & 3) * 1024) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "m0");
    };

    // ---- weights: lane (row r16 of the wave's 16, k chunk kc) -> 16 bytes at k = 64 j + 16 kc of the step, j = 0..3
    int64_t nrow = n0 + 16 * wave + r16;
    nrow = nrow < N ? nrow : N - 1;
    const uint8_t *wrow = W + nrow * K + k_begin + 16 * kc;
    const float sc = w8_row_scale<WF>(scales[nrow]);
    struct WRegs { u32x4 w[4]; };
    auto load_w = [&](int step, WRegs &r) __attribute__((always_inline)) {
        gs_static_for<4>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const uint8_t *pw = wrow + (int64_t)step * 256 + 64 * j;
            u32x4 &dw = r.w[j];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dw) : "v"(pw) : "memory");
        });
    };
    // the registers of a step's weights travel through the wait (prologue) / past it (steps 2 ..): no use can move above the point where
    // they have landed
    auto landed = [&](WRegs &r, bool wait) __attribute__((always_inline)) {
        u32x4 &w0 = r.w[0], &w1 = r.w[1], &w2 = r.w[2], &w3 = r.w[3];
        if (wait) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)::"memory");
        else asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)::"memory");
    };

    // ---- activation fragment addresses: row 16 g + r16, chunk 8 j + 2 kc + h -> one register per (j & 1, h); g, j >> 1, stage immediates
    int fa[2][2];
#pragma unroll
    for (int jl = 0; jl < 2; jl++)
#pragma unroll
        for (int h = 0; h < 2; h++) fa[jl][h] = r16 * 512 + 16 * ((8 * jl + 2 * kc + h) ^ r16);

    f32x4 acc[MF];
#pragma unroll
    for (int g = 0; g < MF; g++) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    WRegs wr[MAXS];
    {
        const int soff0 = 0;
        gs_static_for<NPW>([&](auto ii) { issue_piece(ii, 0, soff0); });
    }
    // the weights stream in step by step as in k_gemm_small (gemm_small.h, round 3): two steps up front, the request of step t + 2 at the END
    // of step t, left in flight by the vmcnt(NW) at the start of step t + 1
    constexpr int NW = 4;     // vector-memory instructions of one step's weight request
    gs_static_for<2>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if (i < nsteps) {
            load_w(i, wr[i]);
        } else {
            gs_static_for<4>([&](auto jj) { wr[i].w[decltype(jj)::value] = u32x4{0, 0, 0, 0}; });
        }
    });
    landed(wr[0], true);
    landed(wr[1], true);

    auto compute = [&](int stage, const WRegs &w, int next) __attribute__((always_inline)) {
        const int nsoff = __builtin_amdgcn_readfirstlane((next < 0 ? 0 : next) << 9);
        Frag xf[2][MF];
        auto issue_x = [&](auto ii, auto pp) __attribute__((always_inline)) {
            constexpr int i = decltype(ii)::value, P = decltype(pp)::value, j = i >> 1, h = i & 1;
#pragma unroll
            for (int g = 0; g < MF; g++)
                xf[P][g] = *reinterpret_cast<const Frag *>(smem + stage * STAGE + fa[j & 1][h] + g * 16 * 512 + (j >> 1) * 256);
        };
        auto decode = [&](auto ii) __attribute__((always_inline)) {
            constexpr int i = decltype(ii)::value, j = i >> 1, h = i & 1;
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const uint32_t word = w.w[j][2 * h + d];
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const float q0 = w8_decode_sel<WF>(word, 2 * b), q1 = w8_decode_sel<WF>(word, 2 * b + 1);
                    float p0, p1;
                    asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(q0), "v"(sc));
                    asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(q1), "v"(sc));
                    o[2 * d + b] = pack2<T>(p0, p1);
                }
            }
            return __builtin_bit_cast(Frag, o);
        };
        issue_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        Frag wf = decode(std::integral_constant<int, 0>{});
        gs_static_for<8>([&](auto ii) {
            constexpr int i = decltype(ii)::value, P = i & 1;
            if constexpr (i < 7) issue_x(std::integral_constant<int, (i + 1) & 7>{}, std::integral_constant<int, P ^ 1>{});
            Frag wn = wf;
            if constexpr (i < 7) wn = decode(std::integral_constant<int, (i + 1) & 7>{});
#pragma unroll
            for (int g = 0; g < MF; g++) acc[g] = Mfma16<T>::run(wf, xf[P][g], acc[g]);
            if (next >= 0) {
                gs_static_for<NPW / 8>([&](auto pp) { issue_piece(std::integral_constant<int, i * (NPW / 8) + decltype(pp)::value>{}, stage ^ 1, nsoff); });
            }
            wf = wn;
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    auto step = [&](auto tt) __attribute__((always_inline)) {
        constexpr int TT = decltype(tt)::value;
        if constexpr (TT == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int more = __builtin_amdgcn_readfirstlane(TT + 1 < nsteps ? 1 : 0);
            asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lg8_w0%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lg8_w1%=\n.Lg8_w0%=:\n\ts_waitcnt vmcnt(0)\n.Lg8_w1%=:"
                         ::"s"(more), "n"(NW) : "scc", "memory");
        }
        __syncthreads();
        if constexpr (TT >= 2) landed(wr[TT], false);
        compute(TT & 1, wr[TT], TT + 1 < nsteps ? TT + 1 : -1);
        if constexpr (TT + 2 < MAXS) {
            if (TT + 2 < nsteps) load_w(TT + 2, wr[TT + 2]);
        }
    };
    gs_static_for<MAXS>([&](auto tt) {
        if (decltype(tt)::value < nsteps) step(tt);
    });

    // ---- epilogue: acc[g][r] = out[m0 + 16 g + (lane & 15)][n0 + 16 wave + 4 (lane >> 4) + r]
    const int64_t nn = n0 + 16 * wave + 4 * kc;
    if (partial != nullptr) {
        float *o = partial + (int64_t)slice * M * N;
#pragma unroll
        for (int g = 0; g < MF; g++) {
            const int64_t m = m0 + 16 * g + r16;
            float v[4] = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
            if (m < M && nn < N) store4_partial(o + m * N + nn, v, nn, N);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < MF; g++) {
        const int64_t m = m0 + 16 * g + r16;
        if (m >= M || nn >= N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float sv = acc[g][e];
            if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
            v[e] = to_f32(from_f32<T>(sv));
        }
        store4(out + m * N + nn, v, nn, N);
    }
}

}  // namespace mbnb
