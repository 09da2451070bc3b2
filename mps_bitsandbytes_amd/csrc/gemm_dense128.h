// gemm_dense128.h — k_gemm_dense128: the decode-once GEMM (gemm_dense.h: out = X [M, K] * Wd [N, ldw]^T + bias on an already
// dequantised 16-bit weight) on 128 x 128 x 64 tiles, for batches that give the 256 x 256 / 256 x 128 tiles of k_gemm_dense
// fewer workgroups than the chip has CUs (768 - 2048 rows at N = 4096: there k_gemm_dense splits K over f32 partials and pays a
// reduction pass, 14 us of 54 at M = 1024).  Same ingredients, re-proportioned:
//   * four waves as 2 (n) x 2 (m), 64 x 64 per wave = 4 x 4 fragments of 16 x 16 on v_mfma_f32_16x16x32 (64 accumulator
//     registers, pinned to AGPRs by issuing the MFMAs from assembly: gemm_fused4.h);
//   * THREE LDS stages of (A 16 KiB + B 16 KiB): a k-step is only 32 MFMAs (512 cycles), less than an L2 round trip, so the
//     8 LDS-DMA pieces of tile j+2 go out at the START of k-step j (their stage was freed one k-step ago) and have more than a
//     whole k-step to land; one barrier per k-step (behind the vmcnt for tile j+1, in front of its first fragment read) covers
//     both hazards: tile j+1 visible, and every wave done reading tile j-1 -- the stage the NEXT k-step's pieces overwrite;
//   * per k-step and wave 32 slots of one MFMA + fillers: slots 0-7 the DMA pieces, slot 18 vmcnt + barrier, slots 18-31 the 16
//     fragment reads of tile j+1 (both k32 slices: 64 registers, double-buffered against the 64 of tile j).
// Output bits: a row's sum runs over k in the same order as in k_gemm_dense (k32 slices in order, one MFMA chain per
// accumulator), so the result equals the unsplit k_gemm_dense result bit for bit (tests assert it).
// Requirements (launcher): K % 64 == 0, K >= 192, 128 * max(K, ldw) * 2 < 2^31, 16-byte aligned rows.
#pragma once
#include "gemm_dense.h"

namespace mbnb {

#ifndef G128_EPI_ONE_PART
#define G128_EPI_ONE_PART 0   // diagnostic builds: 1 = the 16-bit epilogue converts the whole 64-row tile before its first store
#endif
#ifndef G128_ABL
#define G128_ABL 0       // diagnostic builds: 1 no LDS-DMA pieces in the loop, 2 no fragment reads in the loop (timing only)
#endif
#ifndef G128_STAMPS
#define G128_STAMPS 0     // diagnostic builds: 1 the vmcnt wait of the k-step's barrier, 2 the barrier itself
#endif
#if G128_STAMPS
__device__ unsigned long long g_d128_stamps[16];
#endif

constexpr int G128_STAGE = 32768;            // A image 16 KiB (128 rows x 128 B) + B image 16 KiB
constexpr int G128_LDS = 3 * G128_STAGE;     // the epilogue's staging (4 x 64 rows x 136 B) fits inside

template <typename T>
__global__ __launch_bounds__(256, 1) void k_gemm_dense128(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                          void *__restrict__ out_v, int out_dtype, int64_t M, int64_t N, int64_t K,
                                                          int64_t ldw) {
    using Frag = typename Mfma16<T>::frag;
    constexpr int PM = 8, PN = 4;            // XCD patch of 32 tiles (m x n)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;

    const int64_t tiles_m = (M + 127) >> 7, tiles_n = (N + 127) >> 7;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    int64_t tm, tn;
    if ((tiles_m % PM == 0) && (tiles_n % PN == 0)) {
        const int64_t patch = bid >> 5, within = bid & 31;
        const int64_t patches_m = tiles_m / PM;
        tm = (patch % patches_m) * PM + (within % PM);
        tn = (patch / patches_m) * PN + (within / PM);
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
    const int64_t m0 = tm << 7, n0 = tn << 7;
    const int nk = (int)(K >> 6);

    // ---- LDS-DMA: wave w moves A pieces 4w..4w+3 and B pieces 4w..4w+3 (8 rows x 128 B each); rows in the per-lane offset (range
    // checked by the descriptor: rows past M / N read as zeros), k in the scalar offset
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K), pb = reinterpret_cast<uint64_t>(Wd + n0 * ldw);
        const int64_t rows_a = M - m0 < 128 ? M - m0 : 128, rows_b = N - n0 < 128 ? N - n0 : 128;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)(rows_b * ldw * 2), 0x00020000};
    }
    int voff_a[4], voff_b[4];
#pragma unroll
    for (int pl = 0; pl < 4; pl++) {
        const int row = 8 * (4 * wave + pl) + (lane >> 3);
        const int sw = 16 * ((lane & 7) ^ ((row >> 1) & 7));
        voff_a[pl] = (int)(row * K * 2) + sw - GD_M0_GROUP * pl * 1024;       // the piece's 1024 pl travel in the instruction offset (issue_piece)
        voff_b[pl] = (int)(row * ldw * 2) + sw - GD_M0_GROUP * pl * 1024;
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    struct DmaCtx { i32x4_t ra, rb; uint32_t lw; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rb[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
        c.lw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(wave * 4096)));
        return c;
    };
    // piece q of the wave's 8 (0-3: A, 4-7: B) of the tile at byte position kb into stage `stage`
    auto issue_piece = [&](auto qq, int stage, int kb, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value, pl = q & 3;
        const uint32_t dst = c.lw + (uint32_t)(stage * G128_STAGE + (q < 4 ? 0 : 16384) + pl * 1024);
        const int vo = (q < 4) ? voff_a[pl] : voff_b[pl];
        const i32x4_t rs = (q < 4) ? c.ra : c.rb;
        // the operand's four pieces share ONE M0 write: the instruction offset (added to the LDS address and to the global address alike)
        // carries the piece, the per-lane offsets are 1024 pl smaller (gemm_dense.h, GD_M0_GROUP)
        if constexpr (GD_M0_GROUP && pl != 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rs), "s"(kb), "n"(pl * 1024) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
    };

    // ---- fragment reads: lane l = row l & 15 of the fragment's 16, k chunk 4 ks + (l >> 4), swizzled by the row
    const int r16 = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int f = r16 * ROW_BYTES + (((4 * ks + fq) ^ (r16 >> 1)) << 4);
        fx[ks] = wm * 64 * ROW_BYTES + f;
        fw[ks] = 16384 + wn * 64 * ROW_BYTES + f;
    }
    Frag wf[2][2][4], xf[2][2][4];     // [tile parity][k32 slice][fragment]
    auto read_frag = [&](int stage, auto pp, auto nn) {     // n: 0-3 x slice 0, 4-7 w slice 0, 8-11 x slice 1, 12-15 w slice 1
        constexpr int P = decltype(pp)::value, n = decltype(nn)::value, ks = n >> 3, i = n & 3;
        if constexpr ((n & 4) == 0) xf[P][ks][i] = *reinterpret_cast<const Frag *>(smem + fx[ks] + stage * G128_STAGE + i * 16 * ROW_BYTES);
        else wf[P][ks][i] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * G128_STAGE + i * 16 * ROW_BYTES);
    };
    f32x4 acc[4][4];
    auto mfma_acc = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    auto mfma_zero = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    };
    auto kbytes = [&](int t) { return (t < nk ? t : nk - 1) << 7; };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: tiles 0, 1 -> stages 0, 1; tile 0's fragments -> registers (parity 0)
    {
        const DmaCtx c0 = dma_ctx();
        gd_static_for<8>([&](auto q) { issue_piece(q, 0, 0, c0); });
        gd_static_for<8>([&](auto q) { issue_piece(q, 1, kbytes(1), c0); });
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    gd_static_for<16>([&](auto n) { read_frag(0, I0{}, n); });

#if G128_STAMPS
    uint64_t g_sum = 0, g_cnt = 0, g_t0 = 0;      // diagnostic builds (tools/exp/d128_stamps.hip)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(g_t0) :: "memory");
#endif
    // ---- one k-step: tile j in registers (parity P), tile j+1 in stage s1 (landing), tile j+2 requested into stage s2
    auto kstep = [&](auto pp, auto first, auto wo_, int s1, int s2, int j, const DmaCtx &dc) {
        constexpr int P = decltype(pp)::value, WO = decltype(wo_)::value;
        constexpr bool FIRST = decltype(first)::value;
        using PN_ = std::integral_constant<int, P ^ 1>;
        const int kb2 = __builtin_amdgcn_readfirstlane(kbytes(j + 2));
        gd_static_for<32>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t >> 4, f = (t & 15) >> 2, g = t & 3;
            if constexpr (t == 18) {
                // tile j+1 (requested one k-step ago) has landed for this wave; behind the barrier for all of them -- and every wave
                // has finished reading tile j-1's stage... (its reads were issued in k-step j-1 and waited for by its MFMAs)
#if G128_STAMPS == 1
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory"); g_sum += b_ - a_; g_cnt++; }
#else
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the reads of tile j (issued a k-step ago) are complete: free by now
#if G128_STAMPS == 2
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory"); g_sum += b_ - a_; g_cnt++; }
#else
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
            }
            if constexpr (FIRST && ks == 0) mfma_zero(acc[f][g], wf[P][ks][f], xf[P][ks][g]);
            else mfma_acc(acc[f][g], wf[P][ks][f], xf[P][ks][g]);
            // pieces of tile j+2: stage s2 held tile j-1, which every wave finished reading before the barrier of k-step j-1
            if constexpr (!(G128_ABL & 1) && t >= WO && t < WO + 16 && ((t - WO) & 1) == 0)
                issue_piece(std::integral_constant<int, ((t - WO) >> 1) & 7>{}, s2, kb2, dc);
            // fragments of tile j+1 (16 reads) behind the barrier: slots 18 .. 31 (two in the first two slots)
            if constexpr (!(G128_ABL & 2) && t >= 18) {
                constexpr int n0_ = t == 18 ? 0 : (t == 19 ? 2 : t - 16);
                read_frag(s1, PN_{}, std::integral_constant<int, n0_ & 15>{});
                if constexpr (t < 20) read_frag(s1, PN_{}, std::integral_constant<int, (n0_ + 1) & 15>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto main_loop = [&](auto wo) {
        const DmaCtx dc = dma_ctx();
        int s0 = 0, s1 = 1, s2 = 2;
        kstep(I0{}, std::true_type{}, wo, s1, s2, 0, dc);
        int j = 1;
        { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
        for (; j + 1 < nk; j += 2) {
            kstep(I1{}, std::false_type{}, wo, s1, s2, j, dc);
            { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
            kstep(I0{}, std::false_type{}, wo, s1, s2, j + 1, dc);
            { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
        }
        if (j < nk) kstep(I1{}, std::false_type{}, wo, s1, s2, j, dc);
    };
    if ((wave & 1) == 0) main_loop(I0{});
    else main_loop(I1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if G128_STAMPS
    {
        uint64_t g_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(g_t1) :: "memory");
        if (blockIdx.x == 17 && (threadIdx.x & 63) == 0) {
            unsigned long long *o = g_d128_stamps + 4 * (threadIdx.x >> 6);
            o[0] = g_sum; o[1] = g_cnt; o[2] = g_t1 - g_t0;
        }
    }
#endif

    // ---- epilogue: acc[f][g][r] = out[m0 + 64 wm + 16 g + (lane & 15)][n0 + 64 wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 64, m_base = m0 + wm * 64;
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t m = m_base + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
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
    // 16-bit outputs: the wave's 64 x 64 tile through its private 8.5 KiB of LDS (136-byte row pitch), out as 16-byte stores of
    // whole 128-byte row segments
    constexpr int ROWB = 136;
    char *wave_lds = smem + wave * 64 * ROWB;
    uint16_t *out = static_cast<uint16_t *>(out_v);
    const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
    const bool same_out = out_dtype == (std::is_same_v<T, f16_t> ? MBNB_F16 : MBNB_BF16);
    u32x2 bias_all[4];
    if (bias != nullptr) {
        const uint16_t *bp = reinterpret_cast<const uint16_t *>(bias);
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int64_t n = n_base + 16 * f + 4 * efq;
            uint32_t t[4];
#pragma unroll
            for (int e = 0; e < 4; e++) t[e] = bp[n + e < N ? n + e : N - 1];
            bias_all[f] = u32x2{t[0] | (t[1] << 16), t[2] | (t[3] << 16)};
        }
    }
#if G128_EPI_ONE_PART
    auto epilogue16 = [&](auto wb_t) {
        constexpr bool WB = decltype(wb_t)::value;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int nl = 16 * f + 4 * efq;
            float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (WB) {
#pragma unroll
                for (int e = 0; e < 4; e++) bv[e] = unpack_lo<T>(bias_all[f][e >> 1] >> (16 * (e & 1)));
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    v[e] = sv + bv[e];
                }
                if (!same_out) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = to_f32(from_f32<T>(v[e]));
                }
                u32x2 pk;
                if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = pk;
            }
        }
        const int ch = lane_e & 7;   // 8 rows x 8 chunks of 16 B per instruction
        u32x4 piece[8];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const char *srcp = wave_lds + (p * 8 + (lane_e >> 3)) * ROWB + ch * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
            piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        const int64_t n = n_base + ch * 8;
        if (n < N) {
            if (vec_ok && n + 8 <= N) {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const int64_t m = m_base + p * 8 + (lane_e >> 3);
                    if (m < M) store_out16(reinterpret_cast<u32x4 *>(out + m * N + n), piece[p]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const int64_t m = m_base + p * 8 + (lane_e >> 3);
                    if (m >= M) continue;
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    };
#else
    auto epilogue16 = [&](auto wb_t) {
        constexpr bool WB = decltype(wb_t)::value;
        const int ch = lane_e & 7;   // 8 rows x 8 chunks of 16 B per instruction
        const int64_t n = n_base + ch * 8;
        // four parts of 16 rows, each through its own rows of the staging: the first stores leave after a quarter of the conversions
        // (the same split as k_gemm_dense's, common.h GD_EPI_GROUPS)
#pragma unroll
        for (int g = 0; g < 4; g++) {
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const int nl = 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    v[e] = sv;
                    if constexpr (WB) v[e] += unpack_lo<T>(bias_all[f][e >> 1] >> (16 * (e & 1)));
                }
                if (!same_out) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = to_f32(from_f32<T>(v[e]));
                }
                u32x2 pk;
                if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = pk;
            }
            u32x4 piece[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const char *srcp = wave_lds + (16 * g + p * 8 + (lane_e >> 3)) * ROWB + ch * 16;
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
                piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            if (n < N) {
                if (vec_ok && n + 8 <= N) {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const int64_t m = m_base + 16 * g + p * 8 + (lane_e >> 3);
                        if (m < M) store_out16(reinterpret_cast<u32x4 *>(out + m * N + n), piece[p]);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const int64_t m = m_base + 16 * g + p * 8 + (lane_e >> 3);
                        if (m >= M) continue;
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
    };
#endif
    if (bias != nullptr) epilogue16(std::true_type{});
    else epilogue16(std::false_type{});
}

}  // namespace mbnb
