// gemm_small.h — k_gemm_small: fused NF4/FP4 dequant + MFMA GEMM for FEW activation rows (16/32 < M <= 256), blocksize >= 32.
//
// Why another kernel: at these M every weight element feeds only M rows of MFMA work, a workgroup's k-steps are short, and
// the 128 x 64 kernel of gemm_mid.h spends its k-step on LDS traffic and the barrier for the DECODED WEIGHT image (20-30 us
// for a 4096 x 4096 layer whose weights stream in 2 us).  Here the weight never touches LDS:
//   * workgroup = 64 NF weight rows (n) x one K slice x up to 128 activation rows; 4 waves, wave w owns NF fragments of 16 weight
//     rows and ALL the activation rows: NF x MF accumulators of 16 x 16.  NF = 2 halves the activation bytes a workgroup takes
//     in per weight byte -- the kernel is bound by that inflow (64 n-tiles x 1 MB of activations at M = 128 on a 4096 x 4096
//     layer = 64 MB through the L2 -> LDS path) -- and the fragment reads per MFMA;
//   * a lane decodes exactly the MFMA operand it needs, from registers to registers: for v_mfma_f32_16x16x32 lane l holds
//     8 consecutive k of weight row l & 15; the k order inside a 256-k step is permuted so that those 8 k are one packed
//     dword of a 16-byte load: lane (row r, kc = l >> 4) loads the 16 bytes at k = 32 (4 j + kc) .. + 31 (j = 0, 1) and uses
//     dword s of them as the operand of MFMA slice (j, s), i.e. k = 32 (4 j + kc) + 8 s .. + 7.  The activation operand of that
//     slice is read from LDS at the same k (chunk 16 j + 4 kc + s of the row): any permutation is free on that side;
//   * activations: one LDS image of [16 MF rows][256 k] per stage (512-byte rows), two stages, by LDS-DMA one k-step ahead
//     (buffer form: rows past M read as zeros); bank swizzle chunk ^ (r ^ 4 ((r >> 2 ^ r >> 3) & 1)) on the low four chunk bits
//     makes the 16-lane groups of a ds_read_b128 conflict free (both k-chunk halves of a group land on complementary sets);
//   * absmax: the two values a lane needs per step come straight from global (double-quantised: code + absmax2), one step ahead;
//   * one barrier per 256-k step (64 MFMAs per wave), decode = byte table (ds_read_b64 per packed byte) * absmax in f32 -> RNE
//     16 bit: the bits dequantize_4bit produces.
// Split-K: grid (n tiles, slices, m tiles); f32 partials row-major into the workspace, k_splitk_reduce_rm adds them in slice
// order.  Requirements (launcher): blocksize >= 32, K % 256 == 0, K_weight % 256 == 0, k_per_slice % 256 == 0 and <= 2048 (8 steps), 16-byte aligned X
// rows / packed rows.
#pragma once
#include "gemm256.h"
#include <utility>

namespace mbnb {

template <int... I, class F> __device__ __forceinline__ void gs_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void gs_static_for(F &&f) {
    gs_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

template <int MF> constexpr int gemm_small_lds_bytes() { return 2 * 16 * MF * 512; }
#ifndef GS_PK_MUL
#define GS_PK_MUL 0  // 1 (diagnostic builds): one v_pk_mul_f32 per decoded byte instead of two v_mul_f32 -- same bits, SLOWER (27.9 -> 29.0 us at 512 x 4096^2, profiles/r03_small_pk_mul_ab.txt)
#endif
#ifndef GS_ABL
#define GS_ABL 0     // diagnostic builds (tools/exp/small_stamps.hip): 1 no LDS-DMA pieces in the steps, 2 no decode, 4 no fragment reads (timing only)
#endif
#ifdef GS_STAMPS
__device__ unsigned long long g_gs_stamps[256];      // diagnostic builds (tools/exp/small_stamps.hip): cycle stamps of one wave's steps
#ifndef GS_STAMP_TID
#define GS_STAMP_TID 0
#endif
#endif

// MAXS_: steps of a slice whose weights the prologue loads into registers (12 registers per step and fragment); 0 = 8 (NF = 1) / 4
// (NF = 2).  16 (round 3): K = 4096 in ONE slice -- 384 < M <= 512 rows on a 4096-wide layer are 256 workgroups in one round with
// no partials and no reduction launch.
template <typename T, bool NESTED, int MF, int NF = 1, int MAXS_ = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_small(const T *__restrict__ X, const uint8_t *__restrict__ packed, AbsmaxView am,
                                                       const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                       float *__restrict__ partial, int64_t M, int64_t N, int64_t K,
                                                       int64_t K_weight, int64_t k_per_slice, int qt, int bs_shift) {
    using Frag = typename Mfma16<T>::frag;
    constexpr int ROWS = 16 * MF, STAGE = ROWS * 512, NPW = ROWS / 8;   // rows of A per tile, bytes per stage, DMA pieces per wave
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];   // byte table: entry b = (code[b & 15], code[b >> 4])
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kc = lane >> 4;
    constexpr int MAXS = MAXS_ ? MAXS_ : (NF == 1 ? 8 : 4);     // steps of a slice: its weights live in registers
    const int64_t n0 = (int64_t)blockIdx.x * (64 * NF), m0 = (int64_t)blockIdx.z * ROWS;
    const int slice = blockIdx.y;
    const int64_t k_begin = (int64_t)slice * k_per_slice;
    const int64_t k_len = K - k_begin < k_per_slice ? K - k_begin : k_per_slice;
    const int nsteps = (int)(k_len >> 8);

    {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = tid * 2 + h, bb = e >> 1, nib = (e & 1) ? (bb >> 4) : (bb & 15);
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (nib == i) v = (qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
            s_lut2[e] = v;
        }
    }

    // ---- activations by LDS-DMA: piece p = rows 2p, 2p+1 (1 KiB); wave w moves pieces NPW w .. NPW w + NPW-1.  Lane l: row
    // 2p + (l >> 5), LDS position l & 31 holds source chunk (pos & 16) | ((pos & 15) ^ swz(row)).  Row in the per-lane offset
    // (range-checked: rows past M read as zeros), k position in the scalar offset.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K + k_begin);
        const int64_t rows_a = M - m0 < ROWS ? M - m0 : ROWS;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
#pragma unroll
        for (int e = 0; e < 4; e++) rs_a[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
    }
    auto swz = [](int r) { return r ^ ((((r >> 2) ^ (r >> 3)) & 1) << 2); };
    int voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int row = 2 * (NPW * wave + i) + (lane >> 5), pos = lane & 31;
        voff[i] = (int)(row * K * 2) + 16 * ((pos & 16) | ((pos & 15) ^ swz(row & 15))) - GD_M0_GROUP * (i & 3) * 1024;
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
    auto issue_a = [&](int stage, int step) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(step << 9);    // 256 k x 2 B (uniform; pinned in an SGPR for the "s" operand)
        gs_static_for<NPW>([&](auto ii) { issue_piece(ii, stage, soff); });
    };

    // ---- weights: wave w owns NF fragments of 16 rows, n0 + 16 (NF w + f) + r16; lane (row r16, k chunk kc) -> 16 bytes at
    // k = 32 (4 j + kc) of the step, j = 0, 1
    const uint8_t *wrow[NF];
    int64_t am_row[NF];
    const int64_t nblk = K_weight >> bs_shift;      // absmax blocks per weight row (blocksize = 1 << bs_shift >= 32: a lane's 32-k chunk
                                                    // never straddles a block)
#pragma unroll
    for (int f = 0; f < NF; f++) {
        int64_t nrow = n0 + 16 * (NF * wave + f) + r16;
        nrow = nrow < N ? nrow : N - 1;
        wrow[f] = packed + nrow * (K_weight >> 1) + (k_begin >> 1) + 16 * kc;
        am_row[f] = nrow * nblk;
    }
    // The loads go out from inline assembly and are waited for by hand (a vmcnt(0) that carries the destination registers as
    // operands so that no use can move above it): through the compiler its wait for the CURRENT step's weights would be a
    // vmcnt(0) inside the step, behind the LDS-DMA of the next stage it cannot see -- every step would wait for its own prefetch.
    struct WRegs { u32x4 w[NF][2]; uint32_t a[NF][2]; float a2[NF][2]; };
    auto load_w = [&](int step, WRegs &r) __attribute__((always_inline)) {
        gs_static_for<NF * 2>([&](auto fj) {
            {
                constexpr int f = decltype(fj)::value >> 1, j = decltype(fj)::value & 1;
                const uint8_t *pw = wrow[f] + (int64_t)step * 128 + 64 * j;
                u32x4 &dw = r.w[f][j];
                uint32_t &da = r.a[f][j];
                float &d2 = r.a2[f][j];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dw) : "v"(pw) : "memory");
                const int64_t bi = am_row[f] + ((k_begin + 256 * (int64_t)step + 128 * j + 32 * kc) >> bs_shift);
                if constexpr (NESTED) {
                    const int8_t *pc = am.i8 + bi;
                    const float *p2 = am.am2 + bi / am.bs2;
                    asm volatile("global_load_sbyte %0, %1, off" : "=v"(da) : "v"(pc) : "memory");
                    asm volatile("global_load_dword %0, %1, off" : "=v"(d2) : "v"(p2) : "memory");
                } else {
                    const float *pf = am.f32 + bi;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(da) : "v"(pf) : "memory");
                    d2 = 0.0f;
                }
            }
        });
    };
    auto absmax_of = [&](const WRegs &r, float (&ra)[NF][2]) __attribute__((always_inline)) {
        gs_static_for<NF * 2>([&](auto fj) {
            constexpr int f = decltype(fj)::value >> 1, j = decltype(fj)::value & 1;
            if constexpr (NESTED) ra[f][j] = (float)(int)r.a[f][j] * (r.a2[f][j] / 127.0f);   // dequantize_blockwise arithmetic (functional.py:592-594)
            else ra[f][j] = __builtin_bit_cast(float, r.a[f][j]);
        });
    };

    // ---- activation fragment addresses: row 16 g + r16, chunk 16 j + 4 kc + s -> one register per s; g, j, stage immediates
    int fa[4];
#pragma unroll
    for (int s = 0; s < 4; s++) fa[s] = r16 * 512 + 16 * ((4 * kc + s) ^ swz(r16));

    f32x4 acc[NF][MF];
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int g = 0; g < MF; g++) acc[f][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // Prologue.  Round 3: only the weights of the first THREE steps are requested up front; the request of step t + 3 goes out at the END
    // of step t, behind that step's last LDS-DMA piece, so that the wait at the start of step t + 1 -- vmcnt(NW): everything but the
    // newest NW operations -- leaves it in flight for a whole step; it has landed by the wait of step t + 2 (in-order counter), one step
    // before its first use (the tail of step t + 2 looks the first two slices of step t + 3 up, below), where the registers are passed
    // through an empty asm.  (Until then ALL steps of the slice were requested and awaited here: 128 KiB per workgroup at 16 steps, 33.5 MB
    // over the grid of a 512 x 4096 x 4096 call -- 10 of its 35 us went by before the first MFMA, tools/exp/small_stamps.py.)  The steps
    // are fully unrolled, so every step's registers are their own: no ring, no copies.
    constexpr int NW = NF * 2 * (NESTED ? 3 : 2);     // vector-memory instructions of one step's weight request
    constexpr int PD = 3;
    WRegs wr[MAXS];
    issue_a(0, 0);
    gs_static_for<PD>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if (i < nsteps) {
            load_w(i, wr[i]);
        } else {
            gs_static_for<NF * 2>([&](auto fj) {
                constexpr int f = decltype(fj)::value >> 1, j = decltype(fj)::value & 1;
                wr[i].w[f][j] = u32x4{0, 0, 0, 0};
                wr[i].a[f][j] = 0;
                wr[i].a2[f][j] = 0.0f;
            });
        }
    });
    auto landed = [&](WRegs &r) __attribute__((always_inline)) {    // the registers travel past the wait: no use can move above it
        gs_static_for<NF>([&](auto ff) {
            constexpr int f = decltype(ff)::value;
            u32x4 &w0 = r.w[f][0], &w1 = r.w[f][1];
            uint32_t &a0 = r.a[f][0], &a1 = r.a[f][1];
            float &b0 = r.a2[f][0], &b1 = r.a2[f][1];
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)::"memory");
        });
    };
    // everything but the newest request (step 2's, when the slice has one) has landed
    auto wait_but_newest = [&](bool newest_in_flight) __attribute__((always_inline)) {
        const int more = __builtin_amdgcn_readfirstlane(newest_in_flight ? 1 : 0);
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lgs_w0%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lgs_w1%=\n.Lgs_w0%=:\n\ts_waitcnt vmcnt(0)\n.Lgs_w1%=:"
                     ::"s"(more), "n"(NW) : "scc", "memory");
    };

    // One 256-k step = 8 slices (j, s) of NF x MF MFMAs.  One wave per SIMD: nothing hides an LDS round trip between a read and
    // the MFMA that uses it, so the slices are software-pipelined by hand -- the MF activation fragments of slice i+1 and the
    // byte-table lookups of slice i+2 are issued before the MFMAs of slice i, the products / packing of slice i+1 follow
    // (independent VALU work the matrix pipe runs beside).  Round 3: the pipeline runs ACROSS the step boundary -- the lookups of the
    // next step's slices 0 and 1 and the products of its slice 0 need only that step's weights (registers), so they ride in the last two
    // slices of this step; behind the barrier only the activation fragments of slice 0 are waited for (each step used to start with
    // lookup -> products -> fragment reads -> first MFMA in series: ~550 of its ~2800 cycles, small_stamps.py with GS_ABL = 7).
    // next >= 0: the LDS-DMA pieces of step `next` go out into the other stage from INSIDE the step, NPW / 8 per slice behind that
    // slice's MFMAs (all of them up front cost the wave ~16 x 80 issue cycles before its first MFMA: 2.4 -> 1.x us per step)
    const char *lut2 = reinterpret_cast<const char *>(s_lut2);
    Frag wf_c[NF];                 // finished weight fragment of the coming step's slice 0
    f32x2 lk_c[NF][4];             // table lookups of the coming step's slice 1
    auto lookup = [&](const WRegs &w, auto ii, f32x2 (&dst)[NF][4]) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value, j = i >> 2, sl = i & 3;
        gs_static_for<NF>([&](auto ff) {
            constexpr int f = decltype(ff)::value;
            const uint32_t wd = w.w[f][j][sl];
#pragma unroll
            for (int b = 0; b < 4; b++) dst[f][b] = *reinterpret_cast<const f32x2 *>(lut2 + (((wd >> (8 * b)) & 0xFFu) << 3));
        });
    };
    auto finish = [&](const f32x2 (&src)[NF][4], const float (&ra)[NF][2], auto ii, Frag (&wf)[NF]) __attribute__((always_inline)) {
        constexpr int i = decltype(ii)::value, j = i >> 2;
        gs_static_for<NF>([&](auto ff) {
            constexpr int f = decltype(ff)::value;
            u32x4 o;
#pragma unroll
            for (int b = 0; b < 4; b++) {
#if GS_PK_MUL
                // both products of a byte's pair in one v_pk_mul_f32 (two IEEE products: the same bits)
                f32x2 pr;
                const f32x2 sc2 = f32x2{ra[f][j], ra[f][j]};
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(src[f][b]), "v"(sc2));
                o[b] = pack2<T>(pr[0], pr[1]);
#else
                float p0, p1;
                const float l0 = src[f][b][0], l1 = src[f][b][1], sc = ra[f][j];
                asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(l0), "v"(sc));
                asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(l1), "v"(sc));
                o[b] = pack2<T>(p0, p1);
#endif
            }
            wf[f] = __builtin_bit_cast(Frag, o);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // wnx: the next step's weights (landed a step ago); HAS_NX: that step exists in the unrolled sequence (its registers may still hold
    // nothing when the slice ends earlier: the lookups then read the table at whatever byte values they hold and are never used)
    auto compute = [&](int stage, const WRegs &w, const WRegs &wnx, auto has_nx, int next) __attribute__((always_inline)) {
        constexpr bool HAS_NX = decltype(has_nx)::value;
        const int nsoff = __builtin_amdgcn_readfirstlane((next < 0 ? 0 : next) << 9);
        float ra[NF][2], ra_n[NF][2];
        absmax_of(w, ra);
        if constexpr (HAS_NX) absmax_of(wnx, ra_n);
        Frag xf[2][MF];
        f32x2 lk[2][NF][4];
        auto issue_x = [&](auto ii, auto pp) __attribute__((always_inline)) {
            constexpr int i = decltype(ii)::value, P = decltype(pp)::value, j = i >> 2, sl = i & 3;
#pragma unroll
            for (int g = 0; g < MF; g++) xf[P][g] = *reinterpret_cast<const Frag *>(smem + stage * STAGE + fa[sl] + g * 16 * 512 + j * 256);
        };
        issue_x(I0{}, I0{});
        Frag wf[NF], wn[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) {
            wf[f] = wf_c[f];
#pragma unroll
            for (int b = 0; b < 4; b++) lk[1][f][b] = lk_c[f][b];
        }
        gs_static_for<8>([&](auto ii) {
            constexpr int i = decltype(ii)::value, P = i & 1;
            if constexpr (i < 7 && !(GS_ABL & 4)) issue_x(std::integral_constant<int, (i + 1) & 7>{}, std::integral_constant<int, P ^ 1>{});
            if constexpr (i < 7 && !(GS_ABL & 2)) finish(lk[P ^ 1], ra, std::integral_constant<int, (i + 1) & 7>{}, wn);
            if constexpr (i < 6 && !(GS_ABL & 2)) lookup(w, std::integral_constant<int, (i + 2) & 7>{}, lk[P]);
            if constexpr (i >= 6 && HAS_NX && !(GS_ABL & 2)) lookup(wnx, std::integral_constant<int, i - 6>{}, lk[P]);      // slices 0, 1 of the next step
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int g = 0; g < MF; g++) acc[f][g] = Mfma16<T>::run(wf[f], xf[P][g], acc[f][g]);
            if (next >= 0 && !(GS_ABL & 1)) {
                gs_static_for<NPW / 8>([&](auto pp) { issue_piece(std::integral_constant<int, i * (NPW / 8) + decltype(pp)::value>{}, stage ^ 1, nsoff); });
            }
            if constexpr (i < 7) {
#pragma unroll
                for (int f = 0; f < NF; f++) wf[f] = wn[f];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (HAS_NX) {
            finish(lk[0], ra_n, I0{}, wf_c);
#pragma unroll
            for (int f = 0; f < NF; f++)
#pragma unroll
                for (int b = 0; b < 4; b++) lk_c[f][b] = lk[1][f][b];
        }
    };

    // step t: activations in stage t & 1 (A(t+1) goes out during step t, behind the barrier that frees its stage), weights in wr[t]
    auto step = [&](auto tt) __attribute__((always_inline)) {
        constexpr int TT = decltype(tt)::value;
#ifdef GS_STAMPS
        const uint64_t ta = __builtin_readcyclecounter();
#endif
        // the newest weight request -- step TT + 2's, issued at the end of step TT - 1 (step 2's by the prologue) -- stays in flight
        wait_but_newest(TT + 2 < nsteps);
#ifdef GS_STAMPS
        const uint64_t tw = __builtin_readcyclecounter();
#endif
        if constexpr (TT == 0) {
            // the first step's slices 0 and 1 enter the pipeline here (the byte table is visible behind this barrier)
            landed(wr[0]);
            landed(wr[1]);
            __syncthreads();
            float ra0[NF][2];
            absmax_of(wr[0], ra0);
            f32x2 l0[NF][4];
            lookup(wr[0], I0{}, l0);
            lookup(wr[0], I1{}, lk_c);
            finish(l0, ra0, I0{}, wf_c);
        } else {
            __syncthreads();        // A(t) visible; every wave is done reading the other stage (step t - 1)
            if constexpr (TT + 1 < MAXS) landed(wr[TT + 1]);
        }
#ifdef GS_STAMPS
        const uint64_t tb = __builtin_readcyclecounter();
#endif
        constexpr int NX = TT + 1 < MAXS ? TT + 1 : TT;
        compute(TT & 1, wr[TT], wr[NX], std::integral_constant<bool, (TT + 1 < MAXS)>{}, TT + 1 < nsteps ? TT + 1 : -1);
        if constexpr (TT + PD < MAXS) {
            if (TT + PD < nsteps) load_w(TT + PD, wr[TT + PD]);
        }
#ifdef GS_STAMPS
        if (tid == GS_STAMP_TID && blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0) {
            g_gs_stamps[4 * TT + 0] = ta; g_gs_stamps[4 * TT + 1] = tw; g_gs_stamps[4 * TT + 2] = tb; g_gs_stamps[4 * TT + 3] = __builtin_readcyclecounter();
        }
#endif
    };
    gs_static_for<MAXS>([&](auto tt) {
        if (decltype(tt)::value < nsteps) step(tt);
    });

    // ---- epilogue: acc[f][g][r] = out[m0 + 16 g + (lane & 15)][n0 + 16 (NF wave + f) + 4 (lane >> 4) + r]
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const int64_t nn = n0 + 16 * (NF * wave + f) + 4 * kc;
        if (partial != nullptr) {
            float *o = partial + (int64_t)slice * M * N;
#pragma unroll
            for (int g = 0; g < MF; g++) {
                const int64_t m = m0 + 16 * g + r16;
                float v[4] = {acc[f][g][0], acc[f][g][1], acc[f][g][2], acc[f][g][3]};
                if (m < M && nn < N) store4_partial(o + m * N + nn, v, nn, N);
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < MF; g++) {
            const int64_t m = m0 + 16 * g + r16;
            if (m >= M || nn >= N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float sv = acc[f][g][e];
                if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
                v[e] = to_f32(from_f32<T>(sv));
            }
            if (out_dtype == MBNB_F32) store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
            else if (out_dtype == MBNB_F16) store4(static_cast<f16_t *>(out_v) + m * N + nn, v, nn, N);
            else store4(static_cast<bf16_t *>(out_v) + m * N + nn, v, nn, N);
        }
    }
}

}  // namespace mbnb
