// gemm256v.h — k_gemm256v: 256 x 256 x 64 bf16/f16 MFMA GEMM on an already dequantised weight, FOUR waves (one per SIMD),
// 128 (n) x 128 (m) per wave, software pipeline after the schedule the vendor's hand-written gfx950 kernel uses for this
// shape (read from its disassembly; `DESIGN.md` 5.3b):
//   * two LDS stages of (A 32 KiB + B 32 KiB); fragments of a WHOLE k-step live in registers (32 ds_read_b128 = 128
//     VGPRs), so a stage is free as soon as its second half has been read -- early in the k-step, not at its end;
//   * the 16 LDS-DMA pieces a wave moves per k-step go out right behind that point and have a FULL k-step to land
//     (k_gemm256s / k_gemm256d: under one k-step, and `vmcnt(0)` in front of every barrier);
//   * `buffer_load_dwordx4 ... offen lds`: per-lane offsets that never change, the row block and k position in one SGPR ->
//     no VALU per piece; rows past M / N come back as zeros through the descriptor's range check (no clamps);
//   * every filler sits in a fenced slot behind one MFMA (64 slots per k-step).
// out = X [M, K] * Wd [N, K]^T (+ bias).  Same LDS images / swizzle / MFMA operand order as k_gemm256s: bit-equal outputs.
// Requirements: K % 64 == 0, K >= 128, 256 * K * 2 < 2^31, 16-byte aligned X / Wd rows.
#pragma once
#include "../../mps_bitsandbytes_amd/csrc/gemm256.h"
#include <utility>

namespace mbnb {

template <int... I, class F> __device__ __forceinline__ void g256v_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void g256v_static_for(F &&f) {
    g256v_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

constexpr int gemm256v_lds_bytes() { return 4 * P_IMG; }   // 4 image stages; the epilogue's store staging fits inside

// slot positions (template VAR picks a plan): RPS = fragment reads per slot (2: a pair; 1: one), B1 = slot of barrier 1
// (stage C free), piece i is issued in slot D0 + i * DNUM / DDEN, B2 = slot of vmcnt + barrier 2 (next stage visible; the
// wait leaves the VM pieces of this k-step issued before it in flight), R0 = first slot of the next tile's half-0 reads
template <int VAR> struct G256VPlan {
    static constexpr bool SPREAD = (VAR & 64) != 0;
    static constexpr int RPS = SPREAD ? 1 : 2;
    static constexpr int B1 = SPREAD ? 20 : ((VAR & 1) ? 10 : 12);
    static constexpr int D0 = B1;
    static constexpr int DNUM = SPREAD ? ((VAR & 1) ? 10 : 11) : ((VAR & 4) ? 2 : 1), DDEN = SPREAD ? 4 : 1;
    static constexpr int B2 = (VAR & 2) ? 40 : 44;
    static constexpr int R0 = B2;
    static constexpr int dma_slot(int i) { return D0 + i * DNUM / DDEN; }
    static constexpr int piece_at(int t) {
        for (int i = 0; i < 16; i++)
            if (dma_slot(i) == t) return i;
        return -1;
    }
    static constexpr int vm_at_b2() {
        int n = 0;
        for (int i = 0; i < 16; i++) n += dma_slot(i) < B2 ? 1 : 0;
        return n;
    }
    static constexpr int VM = vm_at_b2();
    static_assert(dma_slot(15) < 64 && dma_slot(0) >= B1, "pieces go out between barrier 1 and the end of the k-step");
    static_assert(DNUM >= DDEN, "one piece per slot at most");
};

template <typename T, int VAR>
__global__ __launch_bounds__(256, 1) void k_gemm256v(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                     void *__restrict__ out_v, int out_dtype, int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    using Plan = G256VPlan<VAR>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;

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

    // ---- LDS-DMA: wave w moves A pieces 8w..8w+7 and B pieces 8w..8w+7 (8 rows x 128 B each).  Piece p, lane l: row
    // 8p + (l >> 3), source chunk (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ (4 (p & 1) + (l >> 4)) -> two per-lane offsets
    // (even / odd p); the row block 8p and the k position go into the scalar offset.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
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
    }
    int voff[2];
#pragma unroll
    for (int par = 0; par < 2; par++) voff[par] = (int)((lane >> 3) * K * 2 + 16 * ((lane & 7) ^ (4 * par + (lane >> 4))));
    const int row_block_bytes = (int)(8 * K * 2);                       // one piece further down
    const int wave_soff = __builtin_amdgcn_readfirstlane(wave * 8 * row_block_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    const uint32_t lds_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 8192u));
    // piece q of the wave's 16 (0-7: A, 8-15: B) of the tile at byte position kb into stage `stage`
    auto issue_piece = [&](auto qq, int stage, int kb) {
        constexpr int q = decltype(qq)::value, pl = q & 7;
        const uint32_t dst = lds_wave + (uint32_t)((q < 8 ? P_A : P_B) + stage * P_IMG + pl * 1024);
        const int soff = wave_soff + pl * row_block_bytes + kb;
        const int vo = voff[pl & 1];
        const i32x4_t rs = (q < 8) ? rs_a : rs_b;
        if constexpr ((VAR & 512) != 0) {
            // M0 carried from piece to piece (set for piece 0 by `m0_start`, advanced AFTER each load): no M0 write in front
            // of the load.  Relies on nothing else in the k-step touching M0 (checked in the disassembly).
            constexpr int step = (q == 7) ? (P_B - P_A - 7 * 1024) : 1024;
            asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, %3" ::"v"(vo), "s"(rs), "s"(soff), "n"(step) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "m0");
        }
    };

    auto m0_start = [&](int stage) {
        if constexpr ((VAR & 512) != 0) asm volatile("s_mov_b32 m0, %0" ::"s"(lds_wave + (uint32_t)(P_A + stage * P_IMG)) : "memory", "m0");
    };

    // ---- fragment reads: chunk 2s + fh of row (32 i + fr), swizzled by the row
    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 128 * ROW_BYTES + f;
    }
    Frag wf[4][4], xf[4][4];     // [k16 sub-step][tile]
    auto read_pair = [&](int stage, auto ss, auto ii) {
        constexpr int s = decltype(ss)::value, idx = decltype(ii)::value;
        wf[s][idx] = *reinterpret_cast<const Frag *>(smem + fw[s] + stage * P_IMG + idx * 32 * ROW_BYTES);
        xf[s][idx] = *reinterpret_cast<const Frag *>(smem + fx[s] + stage * P_IMG + idx * 32 * ROW_BYTES);
    };
    f32x16 acc[4][4];   // never zero-filled: the first k-step's first MFMA group takes a literal-zero C operand

    const int64_t nk = K >> 6;
    auto kbytes = [&](int64_t t) { return (int)((t < nk ? t : nk - 1) << 7); };   // past the end: the last tile again

    // ---- prologue: tile 0 -> stage 0, tile 1 -> stage 1; half 0 of tile 0 -> registers
    m0_start(0);
    g256v_static_for<16>([&](auto q) { issue_piece(q, 0, 0); });
    m0_start(1);
    g256v_static_for<16>([&](auto q) { issue_piece(q, 1, kbytes(1)); });
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    g256v_static_for<4>([&](auto i) { read_pair(0, std::integral_constant<int, 0>{}, i); });
    g256v_static_for<4>([&](auto i) { read_pair(0, std::integral_constant<int, 1>{}, i); });

    // ---- one k-step = 64 fenced slots (MFMA + filler).  Stage C holds tile j, stage Nn tile j+1 (landing).
    //   slots 0-7      read half 1 of tile j (sub-steps 2, 3), two fragments per slot
    //   slot  B1       lgkmcnt(0) + barrier: every wave has tile j in registers -> stage C is free
    //   slots D0..     the wave's 16 pieces of tile j+2 -> stage C
    //   slot  B2       vmcnt(16) + barrier: tile j+1 (issued one k-step ago) has landed for every wave
    //   slots R0..R0+7 read half 0 of tile j+1 (sub-steps 0, 1)
    constexpr bool NODMA = (VAR & 8) != 0, NOREAD = (VAR & 16) != 0, NOBAR = (VAR & 32) != 0;   // timing-only ablations
    constexpr bool NOWAIT = (VAR & 128) != 0, KFIX = (VAR & 256) != 0;   // no vmcnt at barrier 2 / every k-step reloads tile 0
    auto kstep = [&](auto cc, auto first, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        constexpr bool FIRST = decltype(first)::value;
        const int kb2 = KFIX ? 0 : kbytes(j + 2);
        g256v_static_for<64>([&](auto tt) {
            constexpr int t = decltype(tt)::value, s = t >> 4, r = t & 15, i = r >> 2, jj = r & 3;
            if constexpr (t == Plan::B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                m0_start(C);
            }
            if constexpr (t == Plan::B2) {
                if constexpr (!NODMA && !NOWAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Plan::VM) : "memory");
                if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (FIRST && s == 0) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[i][jj] = Mfma<T>::run(wf[s][i], xf[s][jj], zero);
            } else {
                acc[i][jj] = Mfma<T>::run(wf[s][i], xf[s][jj], acc[i][jj]);
            }
            // fragment reads: 16 of half 1 of this tile from slot 0, 16 of half 0 of the next tile from slot R0; read n of a
            // half = sub-step (n >> 3), weight fragments n & 7 < 4, activation fragments n & 7 >= 4 -- pairs (w, x) first
            auto read_one = [&](int stage, auto hh, auto nn) {
                constexpr int h = decltype(hh)::value, n = decltype(nn)::value, sub = 2 * h + (n >> 3), idx = (n & 7) >> 1;
                if constexpr ((n & 1) == 0) wf[sub][idx] = *reinterpret_cast<const Frag *>(smem + fw[sub] + stage * P_IMG + idx * 32 * ROW_BYTES);
                else xf[sub][idx] = *reinterpret_cast<const Frag *>(smem + fx[sub] + stage * P_IMG + idx * 32 * ROW_BYTES);
            };
            if constexpr (!NOREAD) {
                g256v_static_for<Plan::RPS>([&](auto uu) {
                    constexpr int u = decltype(uu)::value;
                    if constexpr (t * Plan::RPS + u < 16) read_one(C, std::integral_constant<int, 1>{}, std::integral_constant<int, t * Plan::RPS + u>{});
                    if constexpr (t >= Plan::R0 && (t - Plan::R0) * Plan::RPS + u < 16)
                        read_one(Nn, std::integral_constant<int, 0>{}, std::integral_constant<int, (t - Plan::R0) * Plan::RPS + u>{});
                });
            }
            if constexpr (!NODMA && Plan::piece_at(t) >= 0) issue_piece(std::integral_constant<int, Plan::piece_at(t) < 0 ? 0 : Plan::piece_at(t)>{}, C, kb2);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // the first k-step is peeled for the literal-zero accumulate; nk >= 2
    kstep(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    int64_t j = 1;
    for (; j + 1 < nk; j += 2) {
        kstep(std::integral_constant<int, 1>{}, std::false_type{}, j);
        kstep(std::integral_constant<int, 0>{}, std::false_type{}, j + 1);
    }
    if (j < nk) kstep(std::integral_constant<int, 1>{}, std::false_type{}, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: stage memory reused as store staging (two 64-row halves per wave, one after the other)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63;
    if (out_dtype != MBNB_F32) {
        char *wave_lds = smem + wave * 64 * 264;
        if (out_dtype == MBNB_F16) {
            epilogue_staged<T, f16_t, 4, 0>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 128, n0 + wn * 128, lane_e);
            epilogue_staged<T, f16_t, 4, 2>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 128 + 64, n0 + wn * 128, lane_e);
        } else {
            epilogue_staged<T, bf16_t, 4, 0>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 128, n0 + wn * 128, lane_e);
            epilogue_staged<T, bf16_t, 4, 2>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 128 + 64, n0 + wn * 128, lane_e);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jx = 0; jx < 4; jx++) {
            const int64_t m = m0 + wm * 128 + jx * 32 + (lane_e & 31);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t nn = n0 + wn * 128 + i * 32 + 8 * g + 4 * (lane_e >> 5);
                if (m >= M || nn >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[i][jx][4 * g + e]));
                    if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(sv));
                }
                store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
}

}  // namespace mbnb
