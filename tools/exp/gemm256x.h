// gemm256x.h — k_gemm256x: k_gemm256v (four waves, 128 x 128 per wave, whole-k-step fragments in registers, LDS-DMA one
// k-step ahead) with the contraction on v_mfma_f32_16x16x32 — the shape the vendor's kernel uses and the one that holds the
// higher clock under MFMA load (MI355X_MICROARCH.md, DVFS give-back item 7).  128 fenced slots of one 16-cycle MFMA per k-step:
//   slots 0,2,..,30      the 16 fragment reads of slice 1 (k 32-63) of this tile
//   slot  B1             lgkmcnt(0) + barrier: stage C free
//   slots D0 + 3 i       the wave's 16 LDS-DMA pieces of tile j+2 -> stage C
//   slot  B2             vmcnt(pieces of this k-step already issued) + barrier: tile j+1 visible
//   slots R0, R0+2, ..   the 16 fragment reads of slice 0 of tile j+1
// Sums differ from the 32x32x16 kernels in the last bits (one k32 per MFMA instead of two k16); parity is against the oracle.
#pragma once
#include "gemm256v.h"

namespace mbnb {

template <int VAR> struct G256XPlan {
    static constexpr int RSTRIDE = 2;
    static constexpr int B1 = (VAR & 1) ? 40 : 36;
    static constexpr int D0 = B1;
    static constexpr int DSTRIDE = ((VAR & 2) || (VAR & 8)) ? 4 : 3;
    static constexpr bool WSTAG = (VAR & 8) != 0;   // wave w issues piece i in slot D0 + 4 i + w: one piece per slot and CU
    static constexpr int B2 = WSTAG ? 100 : ((VAR & 4) ? 88 : 96);
    static constexpr int R0 = B2;
    static constexpr int RSTRIDE2 = WSTAG ? 1 : RSTRIDE;
    static constexpr int dma_slot(int i) { return D0 + i * DSTRIDE; }
    static constexpr int piece_at(int t) {
        for (int i = 0; i < 16; i++)
            if (dma_slot(i) == t) return i;
        return -1;
    }
    static constexpr int vm_at_b2() {
        if (WSTAG) return 16;
        int n = 0;
        for (int i = 0; i < 16; i++) n += dma_slot(i) < B2 ? 1 : 0;
        return n;
    }
    static constexpr int VM = vm_at_b2();
    static_assert(dma_slot(15) < 128, "pieces go out between barrier 1 and the end of the k-step");
    static_assert(R0 + 15 * RSTRIDE2 < 128, "the next tile's slice 0 is in registers before the k-step ends");
    static_assert(!WSTAG || dma_slot(15) + 3 < B2, "staggered pieces are all out before barrier 2");
};

template <typename T, int VAR>
__global__ __launch_bounds__(256, 1) void k_gemm256x(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                     void *__restrict__ out_v, int out_dtype, int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma16<T>::frag;
    using Plan = G256XPlan<VAR>;
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
    int row_block_bytes = (int)(8 * K * 2);                       // one piece further down
    int wave_soff = __builtin_amdgcn_readfirstlane(wave * 8 * row_block_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    uint32_t lds_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 8192u));
    // piece q of the wave's 16 (0-7: A, 8-15: B) of the tile at byte position kb into stage `stage`.  The wave-uniform
    // operands travel in a DmaCtx made by `dma_ctx()` INSIDE the loop copy that uses them: defined there by readfirstlane
    // they are SGPRs for certain (across the per-wave branch the compiler otherwise keeps them in VGPRs, which the
    // "s" operands of the instruction cannot take)
    struct DmaCtx { i32x4_t ra, rb; int rbb, wso; uint32_t lw; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rb[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
        c.rbb = __builtin_amdgcn_readfirstlane(row_block_bytes);
        c.wso = __builtin_amdgcn_readfirstlane(wave_soff);
        c.lw = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_wave);
        return c;
    };
    auto issue_piece = [&](auto qq, int stage, int kb, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value, pl = q & 7;
        const uint32_t dst = c.lw + (uint32_t)((q < 8 ? P_A : P_B) + stage * P_IMG + pl * 1024);
        const int soff = c.wso + pl * c.rbb + kb;
        const int vo = voff[pl & 1];
        const i32x4_t rs = (q < 8) ? c.ra : c.rb;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "m0");
    };
    // ---- fragment reads (16 x 16 x 32): lane l = row l & 15 of the fragment's 16, k chunk 4 ks + (l >> 4), swizzled by the row
    const int r16 = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int f = r16 * ROW_BYTES + (((4 * ks + fq) ^ (r16 >> 1)) << 4);
        fw[ks] = P_B + wn * 128 * ROW_BYTES + f;
        fx[ks] = P_A + wm * 128 * ROW_BYTES + f;
    }
    Frag wf[2][8], xf[2][8];     // [k32 slice][16-row fragment]
    // read n of a slice, in the order the MFMAs want them: w0, x0..x7, w1..w7
    auto read_one = [&](int stage, auto kk, auto nn) {
        constexpr int ks = decltype(kk)::value, n = decltype(nn)::value;
        if constexpr (n == 0) wf[ks][0] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG);
        else if constexpr (n <= 8) xf[ks][n - 1] = *reinterpret_cast<const Frag *>(smem + fx[ks] + stage * P_IMG + (n - 1) * 16 * ROW_BYTES);
        else wf[ks][n - 8] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG + (n - 8) * 16 * ROW_BYTES);
    };
    f32x4 acc[8][8];   // never zero-filled: the first k-step's slice-0 MFMAs take a literal-zero C operand

    const int nk = (int)(K >> 6);
    auto kbytes = [&](int t) { return (t < nk ? t : nk - 1) << 7; };   // past the end: the last tile again

    // ---- prologue: tile 0 -> stage 0, tile 1 -> stage 1; slice 0 of tile 0 -> registers
    {
        const DmaCtx c0 = dma_ctx();
        g256v_static_for<16>([&](auto q) { issue_piece(q, 0, 0, c0); });
        g256v_static_for<16>([&](auto q) { issue_piece(q, 1, kbytes(1), c0); });
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    g256v_static_for<16>([&](auto n) { read_one(0, std::integral_constant<int, 0>{}, n); });

    auto kstep = [&](auto cc, auto first, auto wo_, int j, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value;
        constexpr bool FIRST = decltype(first)::value;
        const int kb2 = __builtin_amdgcn_readfirstlane(kbytes(j + 2));
        g256v_static_for<128>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t >> 6, f = (t & 63) >> 3, g = t & 7;
            if constexpr (t == Plan::B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (t == Plan::B2) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Plan::VM) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (FIRST && ks == 0) {
                const f32x4 zero = {0, 0, 0, 0};
                acc[f][g] = Mfma16<T>::run(wf[ks][f], xf[ks][g], zero);
            } else {
                acc[f][g] = Mfma16<T>::run(wf[ks][f], xf[ks][g], acc[f][g]);
            }
            if constexpr ((t % Plan::RSTRIDE) == 0 && t / Plan::RSTRIDE < 16)
                read_one(C, std::integral_constant<int, 1>{}, std::integral_constant<int, (t / Plan::RSTRIDE) & 15>{});
            if constexpr (t >= Plan::R0 && ((t - Plan::R0) % Plan::RSTRIDE2) == 0 && (t - Plan::R0) / Plan::RSTRIDE2 < 16)
                read_one(Nn, std::integral_constant<int, 0>{}, std::integral_constant<int, ((t - Plan::R0) / Plan::RSTRIDE2) & 15>{});
            if constexpr (Plan::WSTAG) {
                if constexpr (t >= Plan::D0 && t < Plan::D0 + 64 && ((t - Plan::D0) & 3) == WO)
                    issue_piece(std::integral_constant<int, ((t - Plan::D0) >> 2) & 15>{}, C, kb2, dc);
            } else if constexpr (Plan::piece_at(t) >= 0) {
                issue_piece(std::integral_constant<int, Plan::piece_at(t) < 0 ? 0 : Plan::piece_at(t)>{}, C, kb2, dc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // one copy of the loop per wave when the pieces are staggered by wave (a branch per slot would stall the MFMA stream)
    auto main_loop = [&](auto wo) {
        const DmaCtx dc = dma_ctx();
        kstep(std::integral_constant<int, 0>{}, std::true_type{}, wo, 0, dc);
        int j = 1;
        for (; j + 1 < nk; j += 2) {
            kstep(std::integral_constant<int, 1>{}, std::false_type{}, wo, j, dc);
            kstep(std::integral_constant<int, 0>{}, std::false_type{}, wo, j + 1, dc);
        }
        if (j < nk) kstep(std::integral_constant<int, 1>{}, std::false_type{}, wo, j, dc);
    };
    if constexpr (Plan::WSTAG) {
        if (wave == 0) main_loop(std::integral_constant<int, 0>{});
        else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
        else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
        else main_loop(std::integral_constant<int, 3>{});
    } else {
        main_loop(std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: acc[f][g][r] = out[m0 + 128 wm + 16 g + (lane & 15)][n0 + 128 wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 128;
    if constexpr ((VAR & 16) != 0) {   // timing only: no epilogue (one conditional store keeps the accumulators alive)
        float sv = 0.0f;
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int g = 0; g < 8; g++) sv += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
        if (sv == 12345.678f) static_cast<float *>(out_v)[0] = sv;
        return;
    }
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int64_t m = m0 + wm * 128 + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(sv));
                }
                if (m < M && nn < N) store4(o + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
    if constexpr ((VAR & 64) != 0) {   // experiment: no LDS staging -- 8-byte non-temporal stores straight from the accumulators
        uint16_t *o16 = static_cast<uint16_t *>(out_v);
#pragma unroll
        for (int g = 0; g < 8; g++)
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int64_t m = m0 + wm * 128 + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    v[e] = to_f32(from_f32<T>(sv));
                }
                const u32x2 pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                if (m < M && nn + 4 <= N) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2 *>(o16 + m * N + nn));
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
    // 16-bit outputs: the wave's tile goes through its private 16.5 KiB of LDS (264-byte row pitch) in two halves of 64 rows
    // and leaves as 16-byte stores of whole 256-byte row segments
    constexpr int ROWB = 264;
    char *wave_lds = smem + wave * 64 * ROWB;
    uint16_t *out = static_cast<uint16_t *>(out_v);
    const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
    g256v_static_for<2>([&](auto hh) {
        constexpr int H = decltype(hh)::value;
        const int64_t m_base = m0 + wm * 128 + 64 * H;
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const int nl = 16 * f + 4 * efq;
            float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int64_t n = n_base + nl + e;
                    bv[e] = to_f32(bias[n < N ? n : N - 1]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][4 * H + g][e]));
                    v[e] = to_f32(from_f32<T>(sv + bv[e]));
                }
                u32x2 pk;
                if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = pk;
            }
        }
        const int ch = lane_e & 15;  // 4 rows x 16 chunks of 16 B per instruction
        u32x4 piece[16];
#pragma unroll
        for (int p = 0; p < 16; p++) {
            const char *srcp = wave_lds + (p * 4 + (lane_e >> 4)) * ROWB + ch * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
            piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        const int64_t n = n_base + ch * 8;
        if (n < N) {
            if (vec_ok && n + 8 <= N) {
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const int64_t m = m_base + p * 4 + (lane_e >> 4);
                    if (m < M) {
                        if constexpr ((VAR & 32) != 0) __builtin_nontemporal_store(piece[p], reinterpret_cast<u32x4 *>(out + m * N + n));
                        else *reinterpret_cast<u32x4 *>(out + m * N + n) = piece[p];
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const int64_t m = m_base + p * 4 + (lane_e >> 4);
                    if (m >= M) continue;
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging reads of this half are done before the next half's writes
    });
}

}  // namespace mbnb
