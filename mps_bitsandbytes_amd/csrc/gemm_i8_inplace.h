// gemm_i8_inplace.h — k_gemm_i8_inplace: matmul_int8 (functional.py:788-793) for large aligned problems on the pipeline of
// k_gemm_dense (gemm_dense.h): 256 x 256 x 128 tiles, FOUR waves (one per SIMD), 128 (n) x 128 (m) per wave on
// v_mfma_i32_16x16x64_i8, fragments of a whole k-step in registers, the LDS-DMA of tile j+2 issued behind the barrier that
// frees its stage (a full k-step to land), one 16-cycle MFMA per fenced slot, pieces staggered by wave.
//
//   out[m, n] = cast( (sum_k A[m, k] * B[k, n]) * (sA[m] / 127) * (sB[n] / 127) )        A int8 [M, K], B int8 [K, N]
//
// B is read where it lies, [K, N] row-major: its LDS image is [128 k][256 n] (256-byte rows), the 16-byte chunk c of k-row
// k stored at position c ^ (((k & 7) << 1) | ((k >> 4) & 1)); the MFMA operand (16 consecutive k of one n per lane) comes
// out of it with two `ds_read_b64_tr_b8` (each: a 16-lane group passes 8 row addresses x 2 halves and receives column i of
// the 8 x 16 block in lane i).  The swizzle makes every 32-lane half of a read hit 64 distinct banks: 8 k-rows x 8 banks,
// the two 16-lane groups of a half (k-chunks kc, kc + 1: same k & 7) on neighbouring chunks.  The activation image is the
// 128-byte-row XOR-swizzled image of gemm_tile.h (128 k per row).
// Slots per k-step (128): 0, 2, .., 30 the 16 fragment loads of slice 1 (k 64-127; an A fragment = one ds_read_b128, a B
// fragment = one pair of transposing reads); 36 lgkmcnt(0) + barrier; 36 + 4 i + w piece i of wave w; 100 vmcnt(16) +
// barrier; 100 .. 115 the 16 fragment loads of slice 0 of tile j+1.
// Requirements (checked by the launcher): K % 128 == 0, N % 16 == 0, 16-byte aligned A / B, 256 * K < 2^31, K * N < 2^31.
//
// Round 2 parked this kernel: with the MFMA builtin hipcc let the accumulators wander between AGPRs and VGPRs (~900 spilled
// registers, ~450 v_accvgpr moves per two k-steps).  Round 3: the MFMAs are issued from assembly with the accumulator pinned
// to an AGPR tuple ("+a": same source and destination), as in gemm_fused4.h -- 222 VGPRs + 256 AGPRs, no spill, no move in the
// loop.  Output bits: the int32 sums are exact, the epilogue is k_gemm_dense<I8>'s (float(sum) * (sA / 127) * (sB / 127), one
// rounding) -- equal to the transposed path bit for bit (tests/test_gpu_parity.py).
#pragma once
#include "gemm_dense.h"

namespace mbnb {

#ifndef GI8_EPI_PARTS
#define GI8_EPI_PARTS 4   // parts the 16-bit epilogue stores the wave's 128 rows in (4: 32 rows each)
#endif
#ifdef GI8_STAMPS     // diagnostic builds (tools/exp/i8_stamps.hip): cycles of the k-loop of workgroup 17's four waves
__device__ unsigned long long g_gi8_stamps[8 + 4 * 256];
#endif

constexpr int GD_B1 = 36, GD_D0 = 36, GD_B2 = 100, GD_R0 = 100;   // slot plan of k_gemm_dense (GdPlan<8>)

typedef int v2i_t __attribute__((ext_vector_type(2)));

template <typename OutT>
__global__ __launch_bounds__(256, 1) void k_gemm_i8_inplace(const int8_t *__restrict__ A, const int8_t *__restrict__ B,
                                                          const float *__restrict__ sA, const float *__restrict__ sB,
                                                          OutT *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
#ifdef GI8_STAMPS
    if (tid == 0) g_gi8_stamps[8 + 4 * blockIdx.x + 0] = wall_clock64();
#endif

    // tile -> workgroup: the walk of k_gemm_dense (gemm_dense.h, round 4): pseudo-patches of 32 tiles dealt to the XCDs in turn, 4 x 8 patches
    // that are ragged at the grid's edges (round 3 fell back to a column-major order whenever tiles_m % 4 or tiles_n % 8 was not 0)
    const int tiles_m = (int)((M + 255) >> 8), tiles_n = (int)((N + 255) >> 8);
    int tm, tn;
    gd_patch_walk<4, 8>(gd_xcd_major(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm << 8, n0 = (int64_t)tn << 8;
    const int nk = (int)(K >> 7);

    // ---- LDS-DMA.  A: wave w moves pieces 8w..8w+7 of 8 rows x 128 B (row in the per-lane offset, k in the scalar offset,
    // rows past M read as zeros).  B: wave w moves k-rows 32w..32w+31 as 8 pieces of 4 rows x 256 B; the k-row inside the tile
    // and the (swizzled) column chunk in the per-lane offset, the tile's k position (k0 * N) in the scalar offset.  The
    // descriptor's range check covers the per-lane offset only (not the scalar one), so it cannot stop a chunk of the last tile column
    // that lies past N: such a chunk would read the NEXT k-row's first columns (harmless: those columns are never stored) and, on
    // the tile's last k-rows, up to 240 bytes past the end of B (a fault when B ends on a page boundary).  Those lanes get a
    // per-lane offset beyond num_records instead: the range check zero-fills them and nothing outside B is touched.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(A + m0 * K), pb = reinterpret_cast<uint64_t>(B + n0);
        const int64_t rows_a = M - m0 < 256 ? M - m0 : 256;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K), 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)(K * N - n0), 0x00020000};
    }
    int voff_a[8], voff_b[8];
#pragma unroll
    for (int pl = 0; pl < 8; pl++) {
        const int row = 8 * (8 * wave + pl) + (lane >> 3);
        voff_a[pl] = (int)(row * K) + 16 * ((lane & 7) ^ ((row >> 1) & 7)) - GD_M0_GROUP * (pl & 3) * 1024;
        const int krow = 32 * wave + 4 * pl + (lane >> 4);
        const int c = (lane & 15) ^ (((krow & 7) << 1) | ((krow >> 4) & 1));
        voff_b[pl] = (n0 + 16 * c < N ? (int)(krow * N) + 16 * c : 0x7FFF0000) - GD_M0_GROUP * (pl & 3) * 1024;
    }
    const int kstep_b = (int)(128 * N);     // bytes of B between two k-steps
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    struct DmaCtx { i32x4_t ra, rb; uint32_t lw; int ksb; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rb[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
        c.lw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 8192u));
        c.ksb = __builtin_amdgcn_readfirstlane(kstep_b);
        return c;
    };
    // piece q of the wave's 16 (0-7: A, 8-15: B) of k-tile t into stage `stage`
    auto issue_piece = [&](auto qq, int stage, int t, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value, pl = q & 7;
        const uint32_t dst = c.lw + (uint32_t)((q < 8 ? P_A : P_B) + stage * P_IMG + pl * 1024);
        const int vo = (q < 8) ? voff_a[pl] : voff_b[pl];
        const i32x4_t rs = (q < 8) ? c.ra : c.rb;
        const int soff = (q < 8) ? (t << 7) : t * c.ksb;
        // four pieces share ONE M0 write: the instruction offset (added to the LDS address and to the global address alike) carries the
        // piece inside the group, the per-lane offsets are that much smaller (gemm_dense.h, GD_M0_GROUP; launcher: K >= 256, N >= 256)
        if constexpr (GD_M0_GROUP && (pl & 3) != 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rs), "s"(soff), "n"((pl & 3) * 1024) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "m0");
    };

    // ---- fragment loads.  Activations (second MFMA operand): lane l = row l & 15, k chunk 4 ks + (l >> 4) of the 128-byte row.
    // B (first operand): lane l = 16 kc + 2 q + p passes the address of k-row 64 ks + 16 kc + 8 r + q, half p of n-chunk
    // 8 wn + f, for r = 0, 1; it receives 8 + 8 consecutive k of column n = 16 (8 wn + f) + (l & 15).
    // The swizzle term of a lane ((q << 1) | (kc & 1)) does not depend on the slice or on r, so one address register per
    // fragment f serves all four reads of it: slice, r and the stage are immediate offsets (64 ks + 8 r rows of 256 B).
    const int r16 = lane & 15, kc = lane >> 4;
    int fx[2], fb[8];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) fx[ks] = P_A + wm * 128 * ROW_BYTES + r16 * ROW_BYTES + (((4 * ks + kc) ^ (r16 >> 1)) << 4);
    {
        const int q = r16 >> 1, p = r16 & 1;
#pragma unroll
        for (int f = 0; f < 8; f++) fb[f] = P_B + (16 * kc + q) * 256 + ((((8 * wn) | f) ^ ((q << 1) | (kc & 1))) << 4) + 8 * p;
    }
    i32x4_t wf[2][8], xf[2][8];     // [k64 slice][16-row fragment]
    // load n of a slice, in the order the MFMAs want them: w0, x0..x7, w1..w7
    auto read_one = [&](int stage, auto kk, auto nn) {
        constexpr int ks = decltype(kk)::value, n = decltype(nn)::value;
        if constexpr (n >= 1 && n <= 8) {
            xf[ks][n - 1] = *reinterpret_cast<const i32x4_t *>(smem + fx[ks] + stage * P_IMG + (n - 1) * 16 * ROW_BYTES);
        } else {
            constexpr int f = (n == 0) ? 0 : n - 8;
            const v2i_t lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(smem + fb[f] + stage * P_IMG + ks * 16384));
            const v2i_t hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(smem + fb[f] + stage * P_IMG + ks * 16384 + 2048));
            wf[ks][f] = i32x4_t{lo[0], lo[1], hi[0], hi[1]};
        }
    };
    i32x4_t acc[8][8];   // never zero-filled: the first k-step's slice-0 MFMAs take a literal-zero C operand

    auto ktile = [&](int t) { return t < nk ? t : nk - 1; };   // past the end: the last tile again (never used)

    {
        const DmaCtx c0 = dma_ctx();
        gd_static_for<16>([&](auto q) { issue_piece(q, 0, 0, c0); });
        gd_static_for<16>([&](auto q) { issue_piece(q, 1, ktile(1), c0); });
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    gd_static_for<16>([&](auto n) { read_one(0, std::integral_constant<int, 0>{}, n); });

    auto kstep = [&](auto cc, auto first, auto wo_, int j, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value;
        constexpr bool FIRST = decltype(first)::value;
        const int t2 = __builtin_amdgcn_readfirstlane(ktile(j + 2));
        gd_static_for<128>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t >> 6, f = (t & 63) >> 3, g = t & 7;
            if constexpr (t == GD_B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (t == GD_B2) {
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // MFMAs from assembly, accumulator pinned to its AGPR tuple (see the header)
            if constexpr (FIRST && ks == 0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=a"(acc[f][g]) : "v"(wf[ks][f]), "v"(xf[ks][g]));
            else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[f][g]) : "v"(wf[ks][f]), "v"(xf[ks][g]));
            if constexpr ((t & 1) == 0 && t < 32) read_one(C, std::integral_constant<int, 1>{}, std::integral_constant<int, (t >> 1) & 15>{});
            if constexpr (t >= GD_R0 && t < GD_R0 + 16) read_one(Nn, std::integral_constant<int, 0>{}, std::integral_constant<int, (t - GD_R0) & 15>{});
            if constexpr (t >= GD_D0 && t < GD_D0 + 64 && ((t - GD_D0) & 3) == WO)
                issue_piece(std::integral_constant<int, ((t - GD_D0) >> 2) & 15>{}, C, t2, dc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
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
#ifdef GI8_STAMPS
    uint64_t gi_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gi_t0) :: "memory");
    const unsigned long long gi_r1 = wall_clock64();
#endif
    if (wave == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GI8_STAMPS
    {
        uint64_t gi_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gi_t1) :: "memory");
        if (blockIdx.x == 17 && (threadIdx.x & 63) == 0) g_gi8_stamps[threadIdx.x >> 6] = gi_t1 - gi_t0;
        if (threadIdx.x == 0) { g_gi8_stamps[8 + 4 * blockIdx.x + 1] = gi_r1; g_gi8_stamps[8 + 4 * blockIdx.x + 2] = wall_clock64(); }
    }
#endif

    // ---- epilogue: acc[f][g][r] = sum for out[m0 + 128 wm + 16 g + (lane & 15)][n0 + 128 wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 128;
    if constexpr (sizeof(OutT) == 4) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int64_t m = m0 + wm * 128 + 16 * g + er16;
            const float sa = sA[m < M ? m : M - 1] / 127.0f;
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int64_t nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int a;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a) : "a"(acc[f][g][e]));
                    v[e] = (float)a * sa * (sB[nn + e < N ? nn + e : N - 1] / 127.0f);
                }
                if (m < M && nn < N) store4(out + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    } else {
        // 16-bit outputs: through the wave's private 16.5 KiB of LDS (264-byte row pitch) in two halves of 64 rows, out as
        // 16-byte stores of whole 256-byte row segments
        constexpr int ROWB = 264;
        char *wave_lds = smem + wave * 64 * ROWB;
        uint16_t *o16 = reinterpret_cast<uint16_t *>(out);
        const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        // the lane's 8 x 4 column scales sB / 127, requested once for both halves (inside the fragment loops each load would
        // cost a memory latency: the accumulator reads are ordered asm)
        float bv_all[8][4];
        {
            const bool sb_vec = (reinterpret_cast<uintptr_t>(sB) & 15) == 0;
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int64_t n = n_base + 16 * f + 4 * efq;
                f32x4 t;
                if (sb_vec && n + 4 <= N) t = *reinterpret_cast<const f32x4 *>(sB + n);
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) t[e] = sB[n + e < N ? n + e : N - 1];
                }
#pragma unroll
                for (int e = 0; e < 4; e++) bv_all[f][e] = t[e] / 127.0f;
            }
        }
        // Round 3: the wave's 128 rows go out in FOUR parts of 32 (two fragments) instead of two of 64 -- the stores of part q are in flight
        // while part q + 1 is read, converted and scaled (4.5 VALU per output: at two parts the epilogue took 8.8 us where the bf16 GEMM's
        // takes 6, tools/exp/i8_stamps.py) --, and the two scale products are packed pairs (v_pk_mul_f32: two IEEE products, the same bits).
        constexpr int NQ = GI8_EPI_PARTS, GQ = 8 / NQ;   // parts, fragments of 16 rows per part
        gd_static_for<NQ>([&](auto hh) {
            constexpr int H = decltype(hh)::value;
            const int64_t m_base = m0 + wm * 128 + 16 * GQ * H;
            float sa[GQ];
#pragma unroll
            for (int g = 0; g < GQ; g++) {
                const int64_t m = m_base + 16 * g + er16;
                sa[g] = sA[m < M ? m : M - 1] / 127.0f;
            }
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int nl = 16 * f + 4 * efq;
#pragma unroll
                for (int g = 0; g < GQ; g++) {
                    int a[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(acc[f][GQ * H + g][e]));
                    f32x2 v01 = f32x2{(float)a[0], (float)a[1]}, v23 = f32x2{(float)a[2], (float)a[3]};
                    const f32x2 s2 = f32x2{sa[g], sa[g]};
                    v01 = v01 * s2 * f32x2{bv_all[f][0], bv_all[f][1]};
                    v23 = v23 * s2 * f32x2{bv_all[f][2], bv_all[f][3]};
                    *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = u32x2{pack2<OutT>(v01[0], v01[1]), pack2<OutT>(v23[0], v23[1])};
                }
            }
            const int ch = lane_e & 15;
            constexpr int NPC = 4 * GQ;            // 16-byte pieces per lane and part: 4 rows per piece index
            u32x4 piece[NPC];
#pragma unroll
            for (int p = 0; p < NPC; p++) {
                const char *srcp = wave_lds + (p * 4 + (lane_e >> 4)) * ROWB + ch * 16;
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
                piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            const int64_t n = n_base + ch * 8;
            if (n < N) {
                if (vec_ok && n + 8 <= N) {
#pragma unroll
                    for (int p = 0; p < NPC; p++) {
                        const int64_t m = m_base + p * 4 + (lane_e >> 4);
                        if (m < M) store_out16_nt(reinterpret_cast<u32x4 *>(o16 + m * N + n), piece[p]);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < NPC; p++) {
                        const int64_t m = m_base + p * 4 + (lane_e >> 4);
                        if (m >= M) continue;
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (n + e < N) o16[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
    }
#ifdef GI8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) g_gi8_stamps[8 + 4 * blockIdx.x + 3] = wall_clock64();
#endif
}

}  // namespace mbnb
