// gemm_dense.h — k_gemm_dense: out = X [M, K] * Wd [N, ldw]^T (+ bias) on a weight that has ALREADY been dequantised into the
// caller's workspace ("decode once": dequantize_4bit writes Wd, this kernel multiplies; matmul_4bit_dispatch pairs the two
// for large M, where the fused kernels re-decode every weight tile once per 256 activation rows).
//
// 256 x 256 x 64 tiles, FOUR waves (one per SIMD), 128 (n) x 128 (m) per wave on v_mfma_f32_16x16x32 (the shape that
// holds the higher clock under MFMA load, MI355X_MICROARCH.md "DVFS give-back" item 7).  The software pipeline follows the
// schedule of the vendor's hand-written gfx950 kernel for this shape, read from its disassembly:
//   * two LDS stages of (A 32 KiB + B 32 KiB); the fragments of a WHOLE k-step live in registers (32 ds_read_b128 = 128
//     VGPRs), so a stage is free once its second k32 slice has been read -- a quarter into the k-step, not at its end;
//   * the wave's 16 LDS-DMA pieces of tile j+2 go out behind that point and have a full k-step to land;
//   * `buffer_load_dwordx4 ... offen lds`: per-lane row offsets that never change, the k position in one SGPR -> no VALU
//     per piece; rows past M / N read as zeros through the descriptor's range check (no clamps);
//   * one 16-cycle MFMA per fenced slot, 128 slots per k-step, every filler behind exactly one MFMA:
//       slots 0, 2, .., 30    the 16 fragment reads of slice 1 (k 32-63) of tile j
//       slot  B1 = 36         lgkmcnt(0) + barrier: stage C free
//       slots 36 + 4 i + w    piece i of wave w (one piece per slot and CU: one copy of the loop per wave -- a branch
//                             per slot stalls the MFMA stream, measured 171 vs 94 us)
//       slot  B2 = 100        vmcnt(16) + barrier: tile j+1 (issued one k-step ago) visible
//       slots 100 .. 115      the 16 fragment reads of slice 0 of tile j+1
// Measured (tools/exp/ab_dense.py, profiles/r02_dense_ab.txt): 4096^3 bf16 94-96 us against 86-90 us for the vendor BLAS on
// the same operands (k-loop 81.7 us per 64 k-steps vs 83; the 32 MB store burst of the epilogue costs 8 us here) and
// 116-124 us for the fused k_gemm256s.
//
// Round 4: the kernel is gemm_dense_body<..., FN> (a device function: one tile = 32 FN columns x 32 FM rows at (m0, n0)) under two wrappers --
// k_gemm_dense (uniform 256-wide columns: every form, 16-bit / split-K / 256 x 128 tiles / int8 / int8 + outliers) and k_gemm_dense_nb (column-balanced
// grids of 224 / 192 / 160-wide columns, FN = 7 / 6 / 5; plan and measurements: gemm_dense.hip) -- and the tile -> workgroup map is
// gd_xcd_major + gd_patch_walk below (pseudo-patches dealt to the XCDs in turn, ragged 4 x 8 patches).
//
// Split-K (grid = tiles x slices): a slice covers `k_per_slice` of K and writes its f32 partial tile row-major into
// partial[slice][M][N]; k_splitk_reduce_rm (gemm_mid.h) adds the slices in index order, then bias, rounding, cast.
// Requirements (checked by the launcher): K % 64 == 0, k_per_slice % 64 == 0, 256 * max(K, ldw) * 2 < 2^31, 16-byte
// aligned X / Wd rows (K % 8 == 0, ldw % 8 == 0, 16-byte aligned bases).  Output bits: the sums of a 16x16x32 MFMA chain equal those of the 32x32x16 chains of k_gemm256s
// on every shape tested (tests/test_gpu_parity.py asserts equality with the fused kernel -- k_gemm_fused4 since round 4 --, and parity with the oracle).
#pragma once
#include "gemm256.h"
#include <type_traits>
#include <utility>

namespace mbnb {

template <int... I, class F> __device__ __forceinline__ void gd_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void gd_static_for(F &&f) {
    gd_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

#ifndef GD_ABL
#define GD_ABL 0        // diagnostic builds (timing only): 1 no LDS-DMA pieces in the k-loop, 2 no fragment reads in the k-loop, 4 no barriers in the k-loop
#endif
#ifndef GD_B_TILE_MAJOR
#define GD_B_TILE_MAJOR 0   // diagnostic builds (tools/exp/dense_tm_exp.hip): Wd is a TILE-MAJOR scratch [n tile of 256 rows][k tile of 64][row][128 B] --
                            // every LDS-DMA piece of the weight operand is 1 KiB contiguous instead of 8 lines `ldw` apart (uniform 256-wide tiles only)
#endif
#ifndef GD_STAMPS
#define GD_STAMPS 0     // diagnostic builds: 1 the vmcnt wait of barrier 2, 2 barrier 2 itself, 3 the lgkmcnt wait of barrier 1, 4 barrier 1 itself
#endif
#if GD_STAMPS
__device__ unsigned long long g_gd_stamps[16];
#endif
constexpr int GD_LDS = 4 * P_IMG;   // A0 A1 B0 B1; the epilogue's store staging (4 x 16.5 KiB) fits inside

// Slot plan of a k-step for a wave tile of 8 (n) x FM (m) fragments of 16 x 16: NS = 16 FM slots, NR = 8 + FM fragment reads
// per k32 slice, NP = 8 + FM LDS-DMA pieces per wave (A: FM, B: 8).
//   FM = 8 (256 x 256 tile): reads of slice 1 every other slot from 0; B1 = 36; piece i of wave w in slot 36 + 4 i + w (one
//           piece per slot and CU); B2 = R0 = 100; the next tile's slice-0 reads in slots 100 .. 115.
//   FM = 4 (256 n x 128 m tile, for grids that would leave CUs idle with 256 x 256): 64 slots; reads in slots 0 .. 11;
//           B1 = 16; piece i of wave w in slot 16 + 2 i + (w & 1) (two pieces per slot and CU); B2 = R0 = 42.
// Round 4: FN < 8 n-fragments per wave (tiles of 32 FN columns: 224 / 192 / 160 wide, FM = 8 only) for the column-balanced grids of
// k_gemm_dense_nb below -- NS = 2 FN FM slots, NR = NP = FN + FM:
//   FN = 7: reads every other slot 0 .. 28; B1 = 32; pieces 32 + 4 i + w (i < 15); B2 = R0 = 94 (reads 94 .. 108 of 112).
//   FN = 6: reads 0 .. 13; B1 = 16; pieces 16 + 4 i + w (i < 14); B2 = R0 = 78 (reads 78 .. 91 of 96).
//   FN = 5: reads 0 .. 12; B1 = 13; pieces 13 + 4 i + w (i < 13); B2 = R0 = 66 (reads 66 .. 78 of 80).
template <int FM, int FN = 8> struct GdPlan {
    static_assert(FN == 8 || (FM == 8 && FN >= 5 && FN <= 7), "narrow tiles exist for the 256-row tile only");
    static constexpr int NS = 2 * FN * FM, NR = FN + FM, NP = FN + FM;
    static constexpr int RS1 = (FM == 8 && FN >= 7) ? 2 : 1;      // slot stride of the slice-1 reads
    static constexpr int B1 = FM != 8 ? 16 : (FN == 8 ? 36 : FN == 7 ? 32 : FN == 6 ? 16 : 13), D0 = B1;
    static constexpr int DS = FM == 8 ? 4 : 2;                    // slot stride of a wave's pieces; wave offset = w & (DS - 1)
    static constexpr int B2 = FM != 8 ? 42 : (FN == 8 ? 100 : FN == 7 ? 94 : FN == 6 ? 78 : 66), R0 = B2;
    static_assert(R0 >= FN * FM, "the next tile's slice 0 replaces this tile's only after its MFMAs");
    static_assert((NR - 1) * RS1 < B1, "slice 1 is in registers before barrier 1");
    static_assert(D0 + NP * DS <= B2, "every piece is issued before the vmcnt of barrier 2");
    static_assert(R0 + NR <= NS, "the next tile's slice 0 is in registers before the k-step ends");
};

// ---- tile -> workgroup map (round 4).  Blocks b, b + 8, ... share an XCD's L2 (MI355X_MICROARCH.md: blocks are dealt round-robin over
// the 8 XCDs), so the walk over the tile grid is cut into pseudo-patches of 32 consecutive tiles and XCD x takes pseudo-patches x, x + 8, ...:
// what an XCD's 32 CUs run together is one compact patch, and -- the walk starts at the WIDE tile columns of a column-balanced grid -- every
// XCD gets the same mix of wide and narrow tiles to within one patch.  The walk itself: patches of PM x PN tiles, ragged at the grid's
// edges (round 3 walked the columns beyond the last full patch one by one: 3 of 43 at N = 11008, and the whole grid when tiles_m % PM != 0),
// patch columns outermost, inside a patch the m index fastest.
__device__ __forceinline__ int gd_xcd_major(int bid, int nwg) {
    const int full = (nwg >> 8) << 8;             // whole groups of 8 pseudo-patches
    if (bid < full) {
        const int i = bid >> 3;
        return (((i >> 5) << 3) + (bid & 7)) * 32 + (i & 31);
    }
    const int rest = nwg - full, r = bid - full, q = rest >> 3, rr = rest & 7, xcd = r & 7;     // the last, partial group: one contiguous run per XCD
    return full + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (r >> 3);
}
template <int PM, int PN> __device__ __forceinline__ void gd_patch_walk(int v, int tiles_m, int tiles_n, int &tm, int &tn) {
    const int col_tiles = tiles_m * PN, pcs = tiles_n / PN, prs = tiles_m / PM;
    int pc = v / col_tiles, w = PN;
    if (pc >= pcs) { pc = pcs; w = tiles_n - pcs * PN; }
    const int v1 = v - pc * col_tiles;            // inside the patch column: tiles_m x w tiles
    int pr = v1 / (PM * w), h = PM;
    if (pr >= prs) { pr = prs; h = tiles_m - prs * PM; }
    const int v2 = v1 - pr * PM * w;              // inside the patch: h x w tiles, m fastest
    tm = pr * PM + v2 % h;
    tn = pc * PN + v2 / h;
}

// I8 = true: the same kernel as an int8 GEMM (matmul_int8, functional.py:788-793).  X = A int8 [M, 2K bytes], Wd = B^T int8
// [N, 2 ldw bytes] -- a row of 2K int8 moves and lands exactly like a row of K 16-bit values, and a lane's 16 bytes of a k32
// slice are the 16 consecutive int8 of a k64 slice of v_mfma_i32_16x16x64_i8 -- T is the 16-bit container type, K and ldw count
// byte PAIRS.  Accumulators hold the int32 sums (as raw bits); epilogue: float(sum) * (sA[m] / 127) * (sB[n] / 127), cast.
// OUTL (with I8, 16-bit outputs): the epilogue adds OutlierAwareLinear's second term and bias (OutlierEpilogue, common.h) with
// the reference's rounding chain -- per 16 x 16 output fragment ONE 16 x 16 x 32 MFMA of the 16-bit outlier weights [n, <= 32]
// and the row's compact outlier activations [m, <= 32] (zero padded), operands straight from global / L2.  OUTL = 1: f16 outputs,
// 2: bf16 outputs (the rounding chain is compiled for one type: every instruction of this epilogue runs 256 times per lane).
// NCH: chunks of 32 outlier columns (1 or 2; f32 accumulation runs through the chunks, one rounding, as the reference's single GEMM).
// FN: n-fragments per wave (tile = 32 FN columns x 32 FM rows); the tile's origin (m0, n0) and its K slice come from the kernel wrappers below.
template <typename T, bool SPLITK, int FM, bool I8, int OUTL, int NCH, int FN>
__device__ __forceinline__ void gemm_dense_body(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                void *__restrict__ out_v, int out_dtype, float *__restrict__ partial,
                                                int64_t M, int64_t N, int64_t K, int64_t ldw, int64_t k_per_slice,
                                                const float *__restrict__ sA, const float *__restrict__ sB, const OutlierEpilogue &ep,
                                                const int64_t m0, const int64_t n0, const int slice) {
    static_assert(!(I8 && SPLITK), "the int8 form is not split");
    static_assert(OUTL == 0 || I8, "the outlier epilogue belongs to the int8 form");
    static_assert(OUTL == 0 || FN == 8, "the outlier epilogue is written for the 256-wide tile");
    using Frag = typename Mfma16<T>::frag;
    using Plan = GdPlan<FM, FN>;
    constexpr int TM = 32 * FM, TN = 32 * FN;   // rows of A / rows of Wd per tile: two waves of 16 FM / 16 FN
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int64_t k_begin = SPLITK ? (int64_t)slice * k_per_slice : 0;
    const int64_t k_len = SPLITK ? (K - k_begin < k_per_slice ? K - k_begin : k_per_slice) : K;
    const int nk = (int)(k_len >> 6);

    // ---- LDS-DMA: wave w moves A pieces FM w .. FM w + FM-1 and B pieces FN w .. FN w + FN-1 (8 rows x 128 B each).  Piece p, lane l: row
    // 8p + (l >> 3), source chunk (l & 7) ^ ((row >> 1) & 7).  The ROW goes into the per-lane offset (8 + 8 VGPRs that never
    // change), the k position into the scalar offset: the descriptor's range check covers the per-lane offset only
    // (the scalar offset is excluded from it), and num_records = rows x pitch makes every row past M / N read as zeros
    // without touching memory -- no clamps, no reads outside the operands.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    i32x4_t rs_a, rs_b;
    {
#if GD_B_TILE_MAJOR
        // tile-major scratch: the tile column's blocks [K / 64][256 rows][64 k] follow each other; rows past N are zero in the scratch
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K + k_begin), pb = reinterpret_cast<uint64_t>(Wd + (n0 >> 8) * (K >> 6) * 16384 + (k_begin >> 6) * 16384);
        const int64_t rows_a = M - m0 < TM ? M - m0 : TM, rows_b = 256;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)((k_len >> 6) * 32768), 0x00020000};
        (void)rows_b;
#else
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K + k_begin), pb = reinterpret_cast<uint64_t>(Wd + n0 * ldw + k_begin);
        const int64_t rows_a = M - m0 < TM ? M - m0 : TM, rows_b = N - n0 < TN ? N - n0 : TN;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
        rs_b = i32x4_t{(int)(uint32_t)pb, (int)(uint32_t)(pb >> 32), (int)(rows_b * ldw * 2), 0x00020000};
#endif
    }
    int voff_a[FM], voff_b[FN];
#pragma unroll
    for (int pl = 0; pl < FN; pl++) {
        const int row = 8 * (FN * wave + pl) + (lane >> 3);
#if GD_B_TILE_MAJOR
        voff_b[pl] = row * 128 + 16 * ((lane & 7) ^ ((row >> 1) & 7));
#else
        voff_b[pl] = (int)(row * ldw * 2) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
#endif
    }
#pragma unroll
    for (int pl = 0; pl < FM; pl++) {
        const int row = 8 * (FM * wave + pl) + (lane >> 3);
        voff_a[pl] = (int)(row * K * 2) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
    }
#if GD_M0_GROUP
    // Pieces 4 g .. 4 g + 3 of an operand share ONE M0 value: the instruction's 12-bit offset is added to the LDS address and to the
    // global address alike, so piece 4 g + m is issued with offset 1024 m from a per-lane offset that is 1024 m smaller.
#pragma unroll
    for (int pl = 0; pl < FN; pl++) voff_b[pl] -= (pl & 3) * 1024;
#pragma unroll
    for (int pl = 0; pl < FM; pl++) voff_a[pl] -= (pl & 3) * 1024;
#endif
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    // The wave-uniform operands of the DMA travel in a DmaCtx made INSIDE the loop copy that uses them: defined there by
    // readfirstlane they are SGPRs for certain (across the per-wave branch the compiler otherwise re-derives them in
    // VGPRs, which the "s" operands of the instruction cannot take).
    struct DmaCtx { i32x4_t ra, rb; uint32_t lwa, lwb; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rb[e] = __builtin_amdgcn_readfirstlane(rs_b[e]);
        }
        c.lwa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(P_A + wave * FM * 1024)));
        c.lwb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(P_B + wave * FN * 1024)));
        return c;
    };
    // piece q of the wave's NP (0 .. FM-1: A, FM .. FM+FN-1: B) of the tile at byte position kb of the slice into stage `stage`
    auto issue_piece = [&](auto qq, int stage, int kb, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value, pl = q < FM ? q : q - FM;
        const uint32_t dst = (q < FM ? c.lwa : c.lwb) + (uint32_t)(stage * P_IMG + pl * 1024);
        const int vo = (q < FM) ? voff_a[pl] : voff_b[pl];
        const i32x4_t rs = (q < FM) ? c.ra : c.rb;
#if GD_B_TILE_MAJOR
        if constexpr (q >= FM) kb = __builtin_amdgcn_readfirstlane(kb << 8);     // k tile t: byte 128 t of a row-major row, block 32768 t of the tile-major scratch
#endif
#if GD_M0_GROUP
        if constexpr ((pl & 3) != 0) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rs), "s"(kb), "n"((pl & 3) * 1024) : "memory", "m0");
        } else {
            const uint32_t dstg = (q < FM ? c.lwa : c.lwb) + (uint32_t)(stage * P_IMG + pl * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dstg), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
        }
#else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
#endif
    };

    // ---- fragment reads (16 x 16 x 32): lane l = row l & 15 of the fragment's 16, k chunk 4 ks + (l >> 4), swizzled by the row
    const int r16 = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int f = r16 * ROW_BYTES + (((4 * ks + fq) ^ (r16 >> 1)) << 4);
        fw[ks] = P_B + wn * 16 * FN * ROW_BYTES + f;
        fx[ks] = P_A + wm * 16 * FM * ROW_BYTES + f;
    }
    Frag wf[2][FN], xf[2][FM];     // [k32 slice][16-row fragment]
    // read n of a slice, in the order the MFMAs want them: w0, x0 .. x(FM-1), w1 .. w(FN-1)
    auto read_one = [&](int stage, auto kk, auto nn) {
        constexpr int ks = decltype(kk)::value, n = decltype(nn)::value;
        if constexpr (n == 0) wf[ks][0] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG);
        else if constexpr (n <= FM) xf[ks][n - 1] = *reinterpret_cast<const Frag *>(smem + fx[ks] + stage * P_IMG + (n - 1) * 16 * ROW_BYTES);
        else wf[ks][n - FM] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG + (n - FM) * 16 * ROW_BYTES);
    };
    f32x4 acc[FN][FM];   // never zero-filled: the first k-step's slice-0 MFMAs take a literal-zero C operand

    auto kbytes = [&](int t) { return (t < nk ? t : nk - 1) << 7; };   // past the end: the last tile again (never used)

    // ---- prologue: tile 0 -> stage 0, tile 1 -> stage 1; slice 0 of tile 0 -> registers
    {
        const DmaCtx c0 = dma_ctx();
        gd_static_for<Plan::NP>([&](auto q) { issue_piece(q, 0, 0, c0); });
        gd_static_for<Plan::NP>([&](auto q) { issue_piece(q, 1, kbytes(1), c0); });
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Plan::NP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    gd_static_for<Plan::NR>([&](auto n) { read_one(0, std::integral_constant<int, 0>{}, n); });

#if GD_STAMPS
    uint64_t gd_sum = 0, gd_cnt = 0, gd_t0 = 0;      // diagnostic builds (tools/exp/dense_stamps.hip): cycles of one wait / barrier, summed over the k-steps
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gd_t0) :: "memory");
#endif
    // ---- one k-step = Plan::NS fenced slots.  Stage C holds tile j, stage Nn tile j+1 (landing); WO = the wave's slot offset.
    auto kstep = [&](auto cc, auto first, auto wo_, int j, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value;
        constexpr bool FIRST = decltype(first)::value;
        const int kb2 = __builtin_amdgcn_readfirstlane(kbytes(j + 2));
        gd_static_for<Plan::NS>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t / (FN * FM), f = (t % (FN * FM)) / FM, g = t % FM;
            if constexpr (t == Plan::B1) {
#if GD_STAMPS == 3
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory"); gd_sum += b_ - a_; gd_cnt++; }
#else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if GD_STAMPS == 4
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory"); gd_sum += b_ - a_; gd_cnt++; }
#else
                if constexpr (!(GD_ABL & 4)) __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
            }
            if constexpr (t == Plan::B2) {
#if GD_STAMPS == 1
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(%2)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) : "n"(Plan::NP) : "memory"); gd_sum += b_ - a_; gd_cnt++; }
#else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Plan::NP) : "memory");
#endif
#if GD_STAMPS == 2
                { uint64_t a_, b_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory"); gd_sum += b_ - a_; gd_cnt++; }
#else
                if constexpr (!(GD_ABL & 4)) __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
            }
            if constexpr (I8) {
                typedef int i32x4_t __attribute__((ext_vector_type(4)));
                i32x4_t c = {0, 0, 0, 0};
                if constexpr (!(FIRST && ks == 0)) c = __builtin_bit_cast(i32x4_t, acc[f][g]);
                acc[f][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, wf[ks][f]),
                                                                                        __builtin_bit_cast(i32x4_t, xf[ks][g]), c, 0, 0, 0));
            } else if constexpr (FIRST && ks == 0) {
                const f32x4 zero = {0, 0, 0, 0};
                acc[f][g] = Mfma16<T>::run(wf[ks][f], xf[ks][g], zero);
            } else {
                acc[f][g] = Mfma16<T>::run(wf[ks][f], xf[ks][g], acc[f][g]);
            }
            if constexpr (!(GD_ABL & 2) && (t % Plan::RS1) == 0 && t / Plan::RS1 < Plan::NR)
                read_one(C, std::integral_constant<int, 1>{}, std::integral_constant<int, (t / Plan::RS1) % Plan::NR>{});
            if constexpr (!(GD_ABL & 2) && t >= Plan::R0 && t < Plan::R0 + Plan::NR)
                read_one(Nn, std::integral_constant<int, 0>{}, std::integral_constant<int, (t - Plan::R0) % Plan::NR>{});
            if constexpr (!(GD_ABL & 1) && t >= Plan::D0 && t < Plan::D0 + Plan::NP * Plan::DS && ((t - Plan::D0) % Plan::DS) == WO)
                issue_piece(std::integral_constant<int, ((t - Plan::D0) / Plan::DS) % Plan::NP>{}, C, kb2, dc);
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
    if constexpr (Plan::DS == 4) {
        if (wave == 0) main_loop(std::integral_constant<int, 0>{});
        else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
        else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
        else main_loop(std::integral_constant<int, 3>{});
    } else {
        if ((wave & 1) == 0) main_loop(std::integral_constant<int, 0>{});
        else main_loop(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if GD_STAMPS
    {
        uint64_t gd_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gd_t1) :: "memory");
        if (blockIdx.x == 17 && (threadIdx.x & 63) == 0) {
            unsigned long long *o = g_gd_stamps + 4 * (threadIdx.x >> 6);
            o[0] = gd_sum; o[1] = gd_cnt; o[2] = gd_t1 - gd_t0;
        }
    }
#endif

    // ---- epilogue: acc[f][g][r] = out[m0 + 16 FM wm + 16 g + (lane & 15)][n0 + 16 FN wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));      // nothing of the epilogue's address arithmetic is hoisted above the loop
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 16 * FN;
    if constexpr (SPLITK) {
        float *o = partial + (int64_t)slice * M * N;
#pragma unroll
        for (int f = 0; f < FN; f++)
#pragma unroll
            for (int g = 0; g < FM; g++) {
                const int64_t m = m0 + wm * 16 * FM + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[e]) : "a"(acc[f][g][e]));
                if (m < M && nn < N) store4_partial(o + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < FN; f++)
#pragma unroll
            for (int g = 0; g < FM; g++) {
                const int64_t m = m0 + wm * 16 * FM + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][g][e]));
                    if constexpr (I8) {
                        v[e] = (float)__builtin_bit_cast(int, sv) * (sA[m < M ? m : M - 1] / 127.0f) * (sB[nn + e < N ? nn + e : N - 1] / 127.0f);
                    } else {
                        if (bias != nullptr && nn + e < N) sv += to_f32(bias[nn + e]);
                        v[e] = to_f32(from_f32<T>(sv));
                    }
                }
                if (m < M && nn < N) store4(o + m * N + nn, v, nn, N);
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
    const bool same_out = !I8 && out_dtype == (std::is_same_v<T, f16_t> ? MBNB_F16 : MBNB_BF16);
    const bool with_outl = OUTL != 0 && ep.x != nullptr && ep.n_out > 0;
    const bool ow_vec = OUTL != 0 && (ep.n_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(ep.ow) & 15) == 0);
    using OutlT = std::conditional_t<OUTL == 1, f16_t, bf16_t>;
    auto rne = [&](float x) { return to_f32(from_f32<OutlT>(x)); };   // round to the output dtype and back
    // OUTL operands, all requested here in one burst (inside the fragment loops each load would cost a full memory latency:
    // the accumulator reads below are ordered asm): per 16-column group f the outlier weights of row n = 16 f + er16 (8 per
    // lane at outlier index 8 efq) and the four bias values of the lane's columns; per 16-row group the compact activations
    u32x4 wfr_all[OUTL ? 8 * NCH : 1];      // [chunk][f]
    u32x2 bias_all[(OUTL || !I8) ? FN : 1];   // 16-bit forms: the kernel's own bias (in T) travels the same way
    u32x4 xfr_all[OUTL ? FM * NCH : 1];     // [chunk][g]
    if constexpr (!I8) {
        if (bias != nullptr) {
            const uint16_t *bp = reinterpret_cast<const uint16_t *>(bias);
#pragma unroll
            for (int f = 0; f < FN; f++) {
                const int64_t n = n_base + 16 * f + 4 * efq;
                if (n + 4 <= N && (reinterpret_cast<uintptr_t>(bp + n) & 7) == 0) bias_all[f] = *reinterpret_cast<const u32x2 *>(bp + n);
                else {
                    uint32_t t[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) t[e] = bp[n + e < N ? n + e : N - 1];
                    bias_all[f] = u32x2{t[0] | (t[1] << 16), t[2] | (t[3] << 16)};
                }
            }
        }
    }
    if constexpr (OUTL) {
#pragma unroll
        for (int f = 0; f < 8; f++) {
            bias_all[f] = u32x2{0u, 0u};
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                wfr_all[8 * c + f] = u32x4{0u, 0u, 0u, 0u};
                if (with_outl) {
                    const uint16_t *ow = static_cast<const uint16_t *>(ep.ow);
                    int64_t n = n_base + 16 * f + er16;
                    n = n < N ? n : N - 1;
                    const int64_t j0 = 32 * c + 8 * efq;
                    if (ow_vec && j0 + 8 <= ep.n_out) wfr_all[8 * c + f] = *reinterpret_cast<const u32x4 *>(ow + n * ep.n_out + j0);
                    else if (j0 < ep.n_out) {
                        uint32_t t[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) t[e] = (j0 + e < ep.n_out) ? (uint32_t)ow[n * ep.n_out + j0 + e] : 0u;
                        wfr_all[8 * c + f] = u32x4{t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16)};
                    }
                }
            }
            if (ep.bias != nullptr) {
                const uint16_t *bp = static_cast<const uint16_t *>(ep.bias);
                const int64_t n = n_base + 16 * f + 4 * efq;
                if (n + 4 <= N && (reinterpret_cast<uintptr_t>(bp + n) & 7) == 0) bias_all[f] = *reinterpret_cast<const u32x2 *>(bp + n);
                else {
                    uint32_t t[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) t[e] = bp[n + e < N ? n + e : N - 1];
                    bias_all[f] = u32x2{t[0] | (t[1] << 16), t[2] | (t[3] << 16)};
                }
            }
        }
#pragma unroll
        for (int g = 0; g < FM; g++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                xfr_all[FM * c + g] = u32x4{0u, 0u, 0u, 0u};
                if (with_outl) {
                    const uint16_t *xx = static_cast<const uint16_t *>(ep.x);
                    int64_t m = m0 + wm * 16 * FM + 16 * g + er16;
                    m = m < M ? m : M - 1;
                    const int64_t j0 = 32 * c + 8 * efq;
                    if (j0 < ep.ldx) xfr_all[FM * c + g] = *reinterpret_cast<const u32x4 *>(xx + m * ep.ldx + j0);
                }
            }
        }
    }
    // I8: the lane's 8 x 4 column scales sB / 127, requested once for both halves (in the fragment loop they cost a memory
    // latency and four IEEE divisions per column group and half)
    float bv_all[I8 ? FN : 1][4];
    if constexpr (I8) {
        const bool sb_vec = (reinterpret_cast<uintptr_t>(sB) & 15) == 0;
#pragma unroll
        for (int f = 0; f < FN; f++) {
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
    // WO / WB: outlier term / bias (OutlierEpilogue's, or the 16-bit kernel's own) present -- compile-time inside the fragment
    // loops (as run-time tests they became four branches per fragment; bias loads in the loops cost a memory latency each)
    auto epilogue16 = [&](auto wo_t, auto wb_t) {
    constexpr bool WO = decltype(wo_t)::value, WB = decltype(wb_t)::value;
    constexpr int GP = (FM % GD_EPI_GROUPS == 0) ? GD_EPI_GROUPS : 4;   // 16-row groups per staged part
    gd_static_for<FM / GP>([&](auto hh) {
        constexpr int H = decltype(hh)::value;
        const int64_t m_base = m0 + wm * 16 * FM + 16 * GP * H;
        float sa[GP] = {};   // I8: the row groups' scales
        if constexpr (I8) {
#pragma unroll
            for (int g = 0; g < GP; g++) {
                const int64_t m = m_base + 16 * g + er16;
                sa[g] = sA[m < M ? m : M - 1] / 127.0f;
            }
        }
#pragma unroll
        for (int f = 0; f < FN; f++) {
            const int nl = 16 * f + 4 * efq;
            float bb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (WB && I8) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t hb = bias_all[f][e >> 1] >> (16 * (e & 1));
                    bb[e] = unpack_lo<OutlT>(hb);
                }
            }
            float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // I8: the four column scales
            if constexpr (I8) {
#pragma unroll
                for (int e = 0; e < 4; e++) bv[e] = bv_all[f][e];
            } else if constexpr (WB) {
#pragma unroll
                for (int e = 0; e < 4; e++) bv[e] = unpack_lo<T>(bias_all[f][e >> 1] >> (16 * (e & 1)));
            }
#pragma unroll
            for (int g = 0; g < GP; g++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sv;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][GP * H + g][e]));
                    if constexpr (I8) v[e] = (float)__builtin_bit_cast(int, sv) * sa[g] * bv[e];
                    else v[e] = sv + bv[e];
                }
                if constexpr (!I8) {
                    // output type != weight type: round to the weight type first (the reference computes in it, then casts);
                    // same type: the pack below is this very rounding (every instruction here runs 256 x per lane)
                    if (!same_out) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = to_f32(from_f32<T>(v[e]));
                    }
                }
                if constexpr (OUTL != 0) {
                    if constexpr (WO) {
                        f32x4 o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        using OFrag = typename Mfma16<OutlT>::frag;
#pragma unroll
                        for (int c = 0; c < NCH; c++)
                            o = Mfma16<OutlT>::run(__builtin_bit_cast(OFrag, wfr_all[8 * c + f]), __builtin_bit_cast(OFrag, xfr_all[FM * c + GP * H + g]), o);
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = rne(rne(v[e]) + rne(o[e]));
                    }
                    if constexpr (WB) {   // OutlierEpilogue bias (v is in the output type's grid after the outlier term: rounded once here otherwise)
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = rne((WO ? v[e] : rne(v[e])) + bb[e]);
                    }
                }
                u32x2 pk;
                if constexpr (OUTL != 0) pk = u32x2{pack2<OutlT>(v[0], v[1]), pack2<OutlT>(v[2], v[3])};
                else if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + er16) * ROWB + nl * 2) = pk;
            }
        }
        const int ch = lane_e & 15;  // 4 rows x 16 chunks of 16 B per instruction
        u32x4 piece[4 * GP];
#pragma unroll
        for (int p = 0; p < 4 * GP; p++) {
            const char *srcp = wave_lds + (p * 4 + (lane_e >> 4)) * ROWB + ch * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
            piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        const int64_t n = n_base + ch * 8;
        if (n < N && ch < 2 * FN) {     // (chunks past the wave's 16 FN columns belong to the neighbouring wave or tile)
            if (vec_ok && n + 8 <= N) {
#pragma unroll
                for (int p = 0; p < 4 * GP; p++) {
                    const int64_t m = m_base + p * 4 + (lane_e >> 4);
                    // not read again by this launch, and 256 workgroups store 32 MB at once: write-through (common.h GD_EPI_STORE; round 2's
                    // nontemporal stores: 98.5 -> 95.6 us at 4096^3 against plain ones, tools/exp/ab_dense.py variants 1008 / 1040)
                    if (m < M) store_out16(reinterpret_cast<u32x4 *>(out + m * N + n), piece[p]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4 * GP; p++) {
                    const int64_t m = m_base + p * 4 + (lane_e >> 4);
                    if (m >= M) continue;
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this half's staging reads are done before the next half's writes
    });
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if constexpr (OUTL != 0) {
        if (with_outl) {
            if (ep.bias != nullptr) epilogue16(Yes{}, Yes{});
            else epilogue16(Yes{}, No{});
        } else {
            if (ep.bias != nullptr) epilogue16(No{}, Yes{});
            else epilogue16(No{}, No{});
        }
    } else if constexpr (!I8) {
        if (bias != nullptr) epilogue16(No{}, Yes{});
        else epilogue16(No{}, No{});
    } else {
        epilogue16(No{}, No{});
    }
}

// The kernel of uniform 256-column tiles (every form: 16-bit, split-K, 256 x 128 tiles, int8, int8 + outliers).
template <typename T, bool SPLITK, int FM = 8, bool I8 = false, int OUTL = 0, int NCH = 1>
__global__ __launch_bounds__(256, 1) void k_gemm_dense(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                       void *__restrict__ out_v, int out_dtype, float *__restrict__ partial,
                                                       int64_t M, int64_t N, int64_t K, int64_t ldw, int64_t k_per_slice,
                                                       const float *__restrict__ sA, const float *__restrict__ sB, OutlierEpilogue ep) {
    constexpr int TM = 32 * FM;
    constexpr int PM = FM == 8 ? 4 : 8, PN = 32 / PM;   // XCD patch of 32 tiles: the one with the smallest operand perimeter
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = (int)((N + 255) >> 8);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x, slice = 0;
    if constexpr (SPLITK) {
        slice = bid / nwg;
        bid -= slice * nwg;
    }
    int tm, tn;
    gd_patch_walk<PM, PN>(gd_xcd_major(bid, nwg), tiles_m, tiles_n, tm, tn);
    gemm_dense_body<T, SPLITK, FM, I8, OUTL, NCH, 8>(X, Wd, bias, out_v, out_dtype, partial, M, N, K, ldw, k_per_slice, sA, sB, ep,
                                                     (int64_t)tm * TM, (int64_t)tn << 8, slice);
}

// Column-balanced grid (round 4): the first `cols_a` tile columns are 32 FNA wide, the remaining tiles_n - cols_a are 32 FNB wide
// (FNB = FNA - 1), chosen by gemm_dense_nb_plan (gemm_dense.hip) so that tiles_m x tiles_n fills the 256 CUs in whole rounds where uniform
// 256-wide columns leave a partial last round (4096 x 11008: 688 tiles = 2.69 rounds run as 3; here 8 columns of 256 + 40 of 224 = 768
// tiles = 3 rounds of which two are 7/8 long).  A workgroup runs the body of its own column's width; a row's summation order does not
// depend on the tile it is computed in, so the bits equal k_gemm_dense's.  256-row tiles, unsplit, 16-bit.
template <typename T, int FNA, int FNB>
__global__ __launch_bounds__(256, 1) void k_gemm_dense_nb(const T *__restrict__ X, const T *__restrict__ Wd, const T *__restrict__ bias,
                                                          void *__restrict__ out_v, int out_dtype, int64_t M, int64_t N, int64_t K,
                                                          int64_t ldw, int tiles_n, int cols_a) {
    const int tiles_m = (int)((M + 255) >> 8);
    int tm, tn;
    gd_patch_walk<4, 8>(gd_xcd_major(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const OutlierEpilogue ep{};
    if (FNA == FNB || tn < cols_a)
        gemm_dense_body<T, false, 8, false, 0, 1, FNA>(X, Wd, bias, out_v, out_dtype, nullptr, M, N, K, ldw, K, nullptr, nullptr, ep,
                                                       (int64_t)tm << 8, (int64_t)tn * (32 * FNA), 0);
    else
        gemm_dense_body<T, false, 8, false, 0, 1, FNB>(X, Wd, bias, out_v, out_dtype, nullptr, M, N, K, ldw, K, nullptr, nullptr, ep,
                                                       (int64_t)tm << 8, (int64_t)cols_a * (32 * FNA) + (int64_t)(tn - cols_a) * (32 * FNB), 0);
}

}  // namespace mbnb
