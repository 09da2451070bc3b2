// gemm_fused4.h — k_gemm_fused4: out = X [M, K] * dequant(W4 [N, K_weight])^T (+ bias), the 4-bit decode INSIDE the four-wave
// MFMA pipeline of k_gemm_dense (gemm_dense.h) — one launch, no N x K scratch (reference: matmul_4bit, functional.py:680-773;
// its fused Metal kernel mm:567-696).  Blocksize 64 (= the k-step: one absmax per weight row and k-step).
//
// Same tile (256 x 256 x 64), wave layout (4 waves, 128 n x 128 m each on v_mfma_f32_16x16x32), LDS images, fragment reads,
// activation LDS-DMA and slot discipline (one 16-cycle MFMA per fenced slot, fillers behind it) as k_gemm_dense.  What
// changes is the weight operand: instead of 8 LDS-DMA pieces of a dequantised weight per wave and k-step, a wave
//   * fetches the PACKED bytes of its 64 weight rows (2 pieces of 32 rows x 32 B; a slot per tile parity, wave-private) and,
//     every fourth k-step, their absmax for four k-steps (1 piece of 64 rows x 16 B; NESTED: the int8 codes + their absmax2);
//   * decodes them — lane = one weight row, 32 packed bytes = 64 values per k-step — through the 256-entry byte table
//     (entry b = (code[b & 15], code[b >> 4]) as two f32: one ds_read_b64 per byte), two v_mul_f32 by the row's absmax, one
//     RNE pack to 16 bit (the bits dequantize_4bit writes), and stores 8 x 16 B into the NEXT tile's weight image.
// The 32 byte-units of a tile are spread over the ~115 slots between two B2 barriers (one unit per 3 - 3.3 slots: 4 VALU + 1
// LDS read, a ds_write_b128 every fourth unit; GfPlan), so the matrix pipe never waits for a decode burst:
//     tile j+2, units 0-6   in slots 106 .. 127 of k-step j   (into stage C, free since barrier 1)
//     tile j+1, units 7-31  in slots   0 ..  94 of k-step j   (into stage N)
//     barrier 2 (slot 100): vmcnt -> A(j+1), raw(j+2), absmax landed; lgkmcnt(0) -> this wave's image writes done
// Every wave decodes each weight element of its tile once per 256 activation rows (the price of fusing; at M = 4096 that is 16 x
// the work of decoding once) — the point of this kernel is that the work rides in the MFMA shadow.
// Output bits: identical to dequantize_4bit + k_gemm_dense (same B-operand bits, same MFMA order per accumulator).
// Requirements (launcher): blocksize 64, K % 64 == 0, K >= 128, K_weight % 256 == 0, 16-byte aligned X / packed / absmax rows.
#pragma once
#include "gemm_dense.h"

namespace mbnb {

constexpr int GF_RAW = 4 * P_IMG;                    // raw packed bytes: [tile parity][wave][2 KiB]
constexpr int GF_RAW_SLOT = 8192;
constexpr int GF_AM = GF_RAW + 2 * GF_RAW_SLOT;      // absmax-by-4: [block parity][wave][1 KiB]  (NESTED: codes 256 B + absmax2 256 B per wave)
constexpr int GF_AM_SLOT = 4096;
constexpr int GF_LDS = GF_AM + 2 * GF_AM_SLOT;       // 155 648 B dynamic + 2 KiB static byte table

// slot plan of the decode (relative to the start of the k-step in which the tile's raw bytes become readable at barrier 2)
struct GfPlan {
    static constexpr int B1 = 36, B2 = 100, NP = 8;            // barriers; A pieces per wave
    // Lookup of unit u, counted from the start of the k-step whose barrier 2 makes the tile's raw bytes readable (>= 128: the
    // next k-step).  No LDS instruction sits in the four slots in front of a barrier (its lgkmcnt(0) would wait for it):
    //   units 0-6    slots 106 .. 124 of k-step j     (tile j+2, behind barrier 2)
    //   units 7-16   slots   0 ..  27 of k-step j+1
    //   units 17-31  slots  38 ..  84 of k-step j+1   (behind barrier 1; last image write in slot 94)
    static constexpr int lookup_slot(int u) { return u < 7 ? 106 + 3 * u : (u < 17 ? 128 + 3 * (u - 7) : 128 + 38 + ((u - 17) * 10) / 3); }
    // products / pack relative to the unit's lookup (three lookups in flight: unit % 3); image write of chunk c behind unit 4c+3
    static constexpr int OM1 = 7, OM2 = 8, OC = 9, OW = 10;
    static constexpr int write_slot(int c) {
        const int t = lookup_slot(4 * c + 3) + OW, m = t % 128;
        return (m >= B1 - 4 && m < B1) ? t - m + B1 + 1 : t;
    }
    static constexpr int RWB = 18;                             // read of the tile's second 16 packed bytes (units 16-31; next k-step)
    // Fragment reads.  The activation fragments of a k32 slice (8 x 4 registers) serve all 64 MFMAs of the slice; a WEIGHT
    // fragment serves 8 consecutive MFMAs only, and the weight image of stage C is not written again before slot ~125 (the
    // decode of tile j+2 starts behind barrier 2), so weight fragments are read shortly before their use and 32 registers
    // hold them instead of 64: {w[0][0-3], w[1][0-3]} and {w[0][4-7], w[1][4-7]} share registers by liveness.
    //   x[1][g] (stage C)   slot 4 g                 before barrier 1: the A image of stage C is refilled behind it
    //   w[0][4+i] (stage C) slots 16, 24, 30, 40     used from slot 32 + 8 i
    //   w[1][i]   (stage C) slot 50 + 8 i            used from slot 64 + 8 i
    //   w[1][4+i] (stage C) slots 80, 85, 90, 94     used from slot 96 + 8 i; before barrier 2
    //   x[0][g], w[0][0-3] of tile j+1 (stage N)     slots 100 .. 111
    static constexpr int x1_slot(int g) { return 4 * g; }
    static constexpr int w0hi_slot(int i) { return i == 0 ? 16 : (i == 1 ? 24 : (i == 2 ? 30 : 40)); }
    static constexpr int w1lo_slot(int i) { return 50 + 8 * i; }
    static constexpr int w1hi_slot(int i) { return i == 3 ? 94 : 80 + 5 * i; }
    // the unit whose action `off` slots behind its lookup falls on slot t of THIS k-step: wrap 0 = the tile decoded behind
    // barrier 2 (tile j+2), wrap 1 = the tile that was begun one k-step ago (tile j+1); -1 = none
    static constexpr int unit_at(int t, int off, int wrap) {
        for (int u = 0; u < 32; u++)
            if (lookup_slot(u) + off - 128 * wrap == t) return u;
        return -1;
    }
    static constexpr int chunk_written_at(int t, int wrap) {
        for (int c = 0; c < 8; c++)
            if (write_slot(c) - 128 * wrap == t) return c;
        return -1;
    }
    // first unit whose lookup falls into the NEXT k-step
    static constexpr int split() {
        int u = 0;
        while (lookup_slot(u) < 128) u++;
        return u;
    }
};

// ABL (diagnostic builds under tools/exp only; the product instantiates 0): 1 no table lookups, 2 no products, 4 no image writes,
// 8 no byte extract / pack either (with 7: no decode work at all), 16 no raw / absmax LDS-DMA, 32 no raw / absmax LDS reads
template <typename T, bool NESTED, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_fused4(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                        const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                        int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma16<T>::frag;
    constexpr int FM = 8, TM = 256, PM = 4, PN = 8;
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;

    // ---- tile -> workgroup map (k_gemm_dense)
    const int64_t tiles_m = (M + TM - 1) / TM, tiles_n = (N + 255) >> 8;
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
    const int64_t m0 = tm * TM, n0 = tn << 8;
    const int nk = (int)(K >> 6);

    // ---- byte table
    {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = tid * 2 + h, b = e >> 1, nib = (e & 1) ? (b >> 4) : (b & 15);
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (nib == i) v = (wp.qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
            s_lut2[e] = v;
        }
    }

    // ---- LDS-DMA descriptors: activations (as k_gemm_dense), packed bytes, absmax.  Rows past M / N read as zeros through
    // the range check of the descriptor (the per-lane offset carries the row).
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const int64_t rowb = wp.K_weight >> 1;             // packed bytes per weight row
    i32x4_t rs_a, rs_p, rs_m, rs_m2;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(X + m0 * K), pp = reinterpret_cast<uint64_t>(wp.packed + n0 * rowb);
        const int64_t rows_a = M - m0 < TM ? M - m0 : TM, rows_b = N - n0 < 256 ? N - n0 : 256;
        rs_a = i32x4_t{(int)(uint32_t)pa, (int)(uint32_t)(pa >> 32), (int)(rows_a * K * 2), 0x00020000};
        rs_p = i32x4_t{(int)(uint32_t)pp, (int)(uint32_t)(pp >> 32), (int)(rows_b * rowb), 0x00020000};
        if constexpr (!NESTED) {
            const uint64_t pm = reinterpret_cast<uint64_t>(wp.am.f32 + n0 * wp.nblk);
            rs_m = i32x4_t{(int)(uint32_t)pm, (int)(uint32_t)(pm >> 32), (int)(rows_b * wp.nblk * 4), 0x00020000};
            rs_m2 = rs_m;
        } else {
            const uint64_t pm = reinterpret_cast<uint64_t>(wp.am.i8 + n0 * wp.nblk);
            rs_m = i32x4_t{(int)(uint32_t)pm, (int)(uint32_t)(pm >> 32), (int)(rows_b * wp.nblk), 0x00020000};
            const uint64_t p2 = reinterpret_cast<uint64_t>(wp.am.am2);
            const int64_t n2 = ((wp.N * wp.nblk - 1) >> wp.bs2_shift) + 1;
            rs_m2 = i32x4_t{(int)(uint32_t)p2, (int)(uint32_t)(p2 >> 32), (int)(n2 * 4), 0x00020000};
        }
    }
    // piece pl of the wave (8 rows x 128 B): row 8 (8 wave + pl) + (lane >> 3), source chunk (lane & 7) ^ ((row >> 1) & 7); the swizzle
    // term depends on the parity of pl only, the row term is linear in pl: voff(pl) = va[pl & 1] + (pl >> 1) * 32 K  (bytes)
    int va[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int row = 8 * (FM * wave + pl) + (lane >> 3);
        va[pl] = (int)(row * K * 2) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    const int a_step = __builtin_amdgcn_readfirstlane((int)(32 * K));
    // raw piece p (0, 1): lane l fetches 16 bytes of row 64 wave + 32 p + (l >> 1), half l & 1 -> LDS slot + 32 row + 16 half
    const int voff_p0 = (int)((64 * wave + (lane >> 1)) * rowb) + 16 * (lane & 1);
    const int voff_p1 = voff_p0 + (int)(32 * rowb);
    // absmax piece: lane l fetches the four k-steps' absmax of row 64 wave + l (16 B; NESTED: 4 codes) -> LDS slot + 16 l (4 l)
    const int am_row = 64 * wave + lane;
    const int voff_m = NESTED ? (int)(am_row * wp.nblk) : (int)(am_row * wp.nblk * 4);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    struct DmaCtx { i32x4_t ra, rp, rm, rm2; uint32_t lwa, lwp, lwm; };
    auto dma_ctx = [&]() {
        DmaCtx c;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c.ra[e] = __builtin_amdgcn_readfirstlane(rs_a[e]);
            c.rp[e] = __builtin_amdgcn_readfirstlane(rs_p[e]);
            c.rm[e] = __builtin_amdgcn_readfirstlane(rs_m[e]);
            c.rm2[e] = __builtin_amdgcn_readfirstlane(rs_m2[e]);
        }
        c.lwa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(P_A + wave * FM * 1024)));
        c.lwp = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(GF_RAW + wave * 2048)));
        c.lwm = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)(GF_AM + wave * 1024)));
        return c;
    };
    int va_k[2] = {va[0], va[1]};      // re-made opaque in every k-step: nothing derived from them is hoisted into loop-long registers
    auto issue_a = [&](auto qq, int stage, int kb, const DmaCtx &c) {
        constexpr int q = decltype(qq)::value;
        const uint32_t dst = c.lwa + (uint32_t)(stage * P_IMG + q * 1024);
        const int vo = va_k[q & 1] + (q >> 1) * a_step;
        const i32x4_t rs = c.ra;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
    };
    // packed bytes of tile t (32 B per row and k-step) into the raw slot `par`
    auto issue_raw = [&](auto pp, int par, int kb, const DmaCtx &c) {
        constexpr int p = decltype(pp)::value;
        const uint32_t dst = c.lwp + (uint32_t)(par * GF_RAW_SLOT + p * 1024);
        const int vo = p == 0 ? voff_p0 : voff_p1;
        const i32x4_t rs = c.rp;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(kb) : "memory", "m0");
    };
    // absmax of the four tiles 4 b .. 4 b + 3 into slot b & 1
    const int nb4 = (int)(wp.nblk >> 2);
    // `go` == 0 skips the fetch with a scalar branch INSIDE the asm statement: a branch the compiler can see splits the k-step
    // into basic blocks, and its register allocation across them falls apart (hundreds of spills, accumulators moved between
    // AGPRs at every loop boundary)
    auto issue_am = [&](int b, int go_, const DmaCtx &c) {
        const int bc = b < nb4 ? b : nb4 - 1;
        const int go = __builtin_amdgcn_readfirstlane(go_);
        if constexpr (!NESTED) {
            const uint32_t dst = c.lwm + (uint32_t)((b & 1) * GF_AM_SLOT);
            const int so = __builtin_amdgcn_readfirstlane(bc << 4);
            const int vo = voff_m;
            const i32x4_t rs = c.rm;
            asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 .Lgf_am_skip%=\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n.Lgf_am_skip%=:"
                         ::"s"(dst), "v"(vo), "s"(rs), "s"(so), "s"(go) : "memory", "m0", "scc");
        } else {
            // codes: 4 bytes per lane (256 B per wave); absmax2: the f32 of the 256-block the four codes lie in (one block: blocksize2 % 4 == 0)
            const uint32_t dst = c.lwm + (uint32_t)((b & 1) * GF_AM_SLOT);
            const int so = __builtin_amdgcn_readfirstlane(bc << 2);
            const int vo = voff_m;
            const i32x4_t rs = c.rm, rs2 = c.rm2;
            int64_t row = n0 + am_row;
            row = row < wp.N ? row : wp.N - 1;
            const int vo2 = (int)(((row * wp.nblk + 4 * (int64_t)bc) >> wp.bs2_shift) << 2);
            const int zero = 0;
            asm volatile("s_cmp_eq_u32 %7, 0\n\ts_cbranch_scc1 .Lgf_am_skip%=\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\t"
                         "s_add_u32 m0, %0, 512\n\ts_nop 0\n\tbuffer_load_dword %4, %5, %6 offen lds\n.Lgf_am_skip%=:"
                         ::"s"(dst), "v"(vo), "s"(rs), "s"(so), "v"(vo2), "s"(rs2), "s"(zero), "s"(go) : "memory", "m0", "scc");
        }
    };
    constexpr int AMN = NESTED ? 2 : 1;     // vector-memory operations of one absmax fetch

    // ---- decode role: lane l owns weight row 64 wave + rl, rl = 2 (l & 7) + ((l >> 3) & 1) + 16 (l >> 4): the 8 lanes of a
    // ds_write_b128 group then differ in row bits 1-3, i.e. hit 8 different swizzled chunks
    const int rl = 2 * (lane & 7) + ((lane >> 3) & 1) + 16 * (lane >> 4);
    const int b_row = 64 * wave + rl;
    const int raw_rd = GF_RAW + wave * 2048 + 32 * rl;                       // + par * GF_RAW_SLOT (+ 16)
    const int am_rd = GF_AM + wave * 1024 + (NESTED ? 4 : 16) * rl;         // + (b & 1) * GF_AM_SLOT (+ 4 (t & 3))
    const int wbase = P_B + b_row * ROW_BYTES + (((b_row >> 1) & 7) << 4);  // image chunk c of the row: wbase ^ (c << 4)
    uint32_t rw[2][8];      // [tile parity][dword]: the lane's 32 packed bytes
    float am[2];            // [tile parity]
    float Lr[3][2];         // lookups in flight (unit % 3)
    float P0 = 0.0f, P1 = 0.0f;
    u32x4 ob[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};     // packed chunk being assembled, by chunk parity (a write may slip behind a barrier)
    int wb_k = wbase;       // re-made opaque in every k-step (see va_k)
    const char *lut2 = reinterpret_cast<const char *>(s_lut2);

    // first half of tile t's bytes + its absmax (slot B2: the DMA of both has been waited for)
    auto read_rwA = [&](auto pp, int t) {
        constexpr int P = decltype(pp)::value;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + raw_rd + P * GF_RAW_SLOT);
        rw[P][0] = v[0]; rw[P][1] = v[1]; rw[P][2] = v[2]; rw[P][3] = v[3];
        const int so = __builtin_amdgcn_readfirstlane(((t >> 2) & 1) * GF_AM_SLOT);
        if constexpr (!NESTED) {
            am[P] = *reinterpret_cast<const float *>(smem + am_rd + so + ((t & 3) << 2));
        } else {
            const uint32_t word = *reinterpret_cast<const uint32_t *>(smem + am_rd + so);
            const float a2 = *reinterpret_cast<const float *>(smem + am_rd + so + 512);
            const float q = (float)(int)(int8_t)(word >> (8 * (t & 3)));
            am[P] = q * (a2 / 127.0f);       // dequantize_blockwise's arithmetic (functional.py:592-594)
        }
    };
    auto read_rwB = [&](auto pp) {
        constexpr int P = decltype(pp)::value;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + raw_rd + P * GF_RAW_SLOT + 16);
        rw[P][4] = v[0]; rw[P][5] = v[1]; rw[P][6] = v[2]; rw[P][7] = v[3];
    };
    // unit u of the tile of parity P: byte u of the lane's 32
    auto dec_lookup = [&](auto pp, auto uu) {
        constexpr int P = decltype(pp)::value, u = decltype(uu)::value;
        if constexpr (ABL & 8) return;
        uint32_t off;
        const uint32_t w = rw[P][u >> 2];
        if constexpr ((u & 3) == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(w));
        else if constexpr ((u & 3) == 1) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(w));
        else if constexpr ((u & 3) == 2) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(w));
        else asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(w));
        if constexpr (ABL & 1) {
            Lr[u % 3][0] = __builtin_bit_cast(float, off);
            Lr[u % 3][1] = __builtin_bit_cast(float, off + 1u);
            return;
        }
        const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + off);
        Lr[u % 3][0] = v[0];
        Lr[u % 3][1] = v[1];
    };
    // scalar multiplies from assembly: kept away from the SLP vectoriser (a packed-f32 VALU op beside MFMAs costs far more
    // issue time than the two scalar ones, MI355X_MICROARCH.md "price of one filler beside MFMAs")
    auto dec_mul1 = [&](auto pp, auto uu) {
        constexpr int P = decltype(pp)::value, u = decltype(uu)::value;
        const float l = Lr[u % 3][0], a = am[P];
        float r = l;
        if constexpr (!(ABL & 2)) asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(l), "v"(a));
        P0 = r;
    };
    auto dec_mul2 = [&](auto pp, auto uu) {
        constexpr int P = decltype(pp)::value, u = decltype(uu)::value;
        const float l = Lr[u % 3][1], a = am[P];
        float r = l;
        if constexpr (!(ABL & 2)) asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(l), "v"(a));
        P1 = r;
    };
    auto dec_pack = [&](auto uu) {
        constexpr int u = decltype(uu)::value;
        if constexpr (ABL & 8) return;
        ob[(u >> 2) & 1][u & 3] = pack2<T>(P0, P1);
    };
    auto dec_write = [&](int stage, auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (ABL & 4) {
            if constexpr (!(ABL & 8)) asm volatile("" ::"v"(ob[c & 1]));     // keeps the producing instructions alive
            return;
        }
        *reinterpret_cast<u32x4 *>(smem + (wb_k ^ (c << 4)) + stage * P_IMG) = ob[c & 1];
    };

    // ---- fragment reads (k_gemm_dense)
    const int r16 = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int f = r16 * ROW_BYTES + (((4 * ks + fq) ^ (r16 >> 1)) << 4);
        fw[ks] = P_B + wn * 128 * ROW_BYTES + f;
        fx[ks] = P_A + wm * 16 * FM * ROW_BYTES + f;
    }
    Frag wf[2][8], xf[2][FM];
    auto read_w = [&](int stage, auto kk, auto ff) {
        constexpr int ks = decltype(kk)::value, f = decltype(ff)::value;
        wf[ks][f] = *reinterpret_cast<const Frag *>(smem + fw[ks] + stage * P_IMG + f * 16 * ROW_BYTES);
    };
    auto read_x = [&](int stage, auto kk, auto gg) {
        constexpr int ks = decltype(kk)::value, g = decltype(gg)::value;
        xf[ks][g] = *reinterpret_cast<const Frag *>(smem + fx[ks] + stage * P_IMG + g * 16 * ROW_BYTES);
    };
    f32x4 acc[8][FM];
    auto mfma_acc = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    auto mfma_zero = [&](f32x4 &c, const Frag &a, const Frag &b) {
        if constexpr (std::is_same_v<T, bf16_t>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    };

    auto tcl = [&](int t) { return t < nk ? t : nk - 1; };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr int USPLIT = GfPlan::split();

    // ---- prologue: absmax block 0, raw of tiles 0 and 1, A(0), A(1); tile 0 decoded into stage 0, the first USPLIT units of
    // tile 1 (what steady state does behind barrier 2 of "k-step -1") into stage 1; raw of tile 2 requested.
    {
        const DmaCtx c0 = dma_ctx();
        issue_am(0, 1, c0);
        issue_raw(I0{}, 0, 0, c0);
        issue_raw(I1{}, 0, 0, c0);
        issue_raw(I0{}, 1, tcl(1) << 5, c0);
        issue_raw(I1{}, 1, tcl(1) << 5, c0);
        gd_static_for<FM>([&](auto q) { issue_a(q, 0, 0, c0); });
        gd_static_for<FM>([&](auto q) { issue_a(q, 1, tcl(1) << 7, c0); });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * FM) : "memory");     // absmax + raw landed; the activations stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // byte table written
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_rwA(I0{}, 0);
        read_rwB(I0{});
        gd_static_for<32>([&](auto u) {
            dec_lookup(I0{}, u);
            dec_mul1(I0{}, u);
            dec_mul2(I0{}, u);
            dec_pack(u);
            if constexpr ((decltype(u)::value & 3) == 3) dec_write(0, std::integral_constant<int, decltype(u)::value / 4>{});
        });
        read_rwA(I1{}, tcl(1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // raw slot 0 read: refill it with tile 2
        issue_raw(I0{}, 0, tcl(2) << 5, c0);
        issue_raw(I1{}, 0, tcl(2) << 5, c0);
        // tile 1, units 0 .. USPLIT-1, in the order and with the register roles of the steady state
        gd_static_for<USPLIT + 4>([&](auto s) {
            constexpr int u = decltype(s)::value, v = u - 3;    // finished three units behind its lookup (register roles of the loop)
            if constexpr (v >= 0 && v < 32) {
                constexpr int r = GfPlan::lookup_slot(v);
                if constexpr (r + GfPlan::OM1 < 128) dec_mul1(I1{}, std::integral_constant<int, v>{});
                if constexpr (r + GfPlan::OM2 < 128) dec_mul2(I1{}, std::integral_constant<int, v>{});
                if constexpr (r + GfPlan::OC < 128) dec_pack(std::integral_constant<int, v>{});
                if constexpr ((v & 3) == 3 && GfPlan::write_slot(v / 4) < 128) dec_write(1, std::integral_constant<int, v / 4>{});
            }
            if constexpr (u < USPLIT) dec_lookup(I1{}, std::integral_constant<int, u>{});
        });
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                   // A(0), A(1) landed (raw of tile 2 may still fly)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    gd_static_for<8>([&](auto g) { read_x(0, I0{}, g); });
    gd_static_for<4>([&](auto f) { read_w(0, I0{}, f); });

    // ---- one k-step: stage C holds tile j (both operands), stage Nn tile j+1 (activations landing, weights being decoded)
    auto kstep = [&](auto cc, auto first, auto wo_, int j, const DmaCtx &dc) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, WO = decltype(wo_)::value;
        constexpr bool FIRST = decltype(first)::value;
        using PC = std::integral_constant<int, C>;
        using PN_ = std::integral_constant<int, Nn>;
        asm volatile("" : "+v"(wb_k), "+v"(va_k[0]), "+v"(va_k[1]));
        const int kb2 = __builtin_amdgcn_readfirstlane(tcl(j + 2) << 7);
        const int rb3 = __builtin_amdgcn_readfirstlane(tcl(j + 3) << 5);
        const int am_now = (((j + 3) & 3) == 0) ? 1 : 0;
        gd_static_for<128>([&](auto tt) {
            constexpr int t = decltype(tt)::value, ks = t / 64, f = (t % 64) / FM, g = t % FM;
            if constexpr (t == GfPlan::B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (t == GfPlan::B2) {
                // everything issued one k-step ago has landed: A(j+1), raw(j+2), (absmax); own image writes are done
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ABL & 16) ? GfPlan::NP : GfPlan::NP + 2) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // MFMAs from assembly with the accumulator pinned to an AGPR tuple ("+a"): with the builtin the allocator is free to
            // keep accumulators in VGPRs or to give an MFMA a destination other than its source, and with the decode's live ranges
            // in the loop it does both (accumulators copied between files at every loop boundary, hundreds of spills)
            if constexpr (FIRST && ks == 0) mfma_zero(acc[f][g], wf[ks][f], xf[ks][g]);
            else mfma_acc(acc[f][g], wf[ks][f], xf[ks][g]);
            // fragment reads (GfPlan)
            if constexpr ((t % 4) == 0 && t / 4 < 8) read_x(C, I1{}, std::integral_constant<int, (t / 4) % 8>{});
            gd_static_for<4>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                if constexpr (t == GfPlan::w0hi_slot(i)) read_w(C, I0{}, std::integral_constant<int, 4 + i>{});
                if constexpr (t == GfPlan::w1lo_slot(i)) read_w(C, I1{}, std::integral_constant<int, i>{});
                if constexpr (t == GfPlan::w1hi_slot(i)) read_w(C, I1{}, std::integral_constant<int, 4 + i>{});
            });
            if constexpr (t >= GfPlan::B2 && t < GfPlan::B2 + 8) read_x(Nn, I0{}, std::integral_constant<int, (t - GfPlan::B2) % 8>{});
            if constexpr (t >= GfPlan::B2 + 8 && t < GfPlan::B2 + 12) read_w(Nn, I0{}, std::integral_constant<int, (t - GfPlan::B2 - 8) % 4>{});
            // LDS-DMA: slot 36 + 4 i + w: i = 0 absmax (every fourth k-step), 1, 2 raw of tile j+3, 3 .. 10 A pieces of tile j+2
            if constexpr (t >= GfPlan::B1 && t < GfPlan::B1 + 4 * 11 && ((t - GfPlan::B1) % 4) == WO) {
                constexpr int i = (t - GfPlan::B1) / 4;
                if constexpr (i == 0) {
                    if constexpr (!(ABL & 16)) issue_am((j + 3) >> 2, am_now, dc);
                } else if constexpr (i == 1) {
                    if constexpr (!(ABL & 16)) issue_raw(I0{}, Nn, rb3, dc);
                } else if constexpr (i == 2) {
                    if constexpr (!(ABL & 16)) issue_raw(I1{}, Nn, rb3, dc);
                } else {
                    issue_a(std::integral_constant<int, (i - 3) % FM>{}, C, kb2, dc);
                }
            }
            // decode: raw reads
            if constexpr (!(ABL & 32)) {
                if constexpr (t == GfPlan::B2) read_rwA(PC{}, tcl(j + 2));
                if constexpr (t == GfPlan::RWB) read_rwB(PN_{});
            }
            // decode: units.  Slots >= R0: tile j+2 (parity C, stage C); slots below: tile j+1 (parity Nn, stage Nn), whose
            // unit u sits at lookup_slot(u) - 128.
            {
                using G = GfPlan;
                constexpr int l0 = G::unit_at(t, 0, 0), l1 = G::unit_at(t, 0, 1);
                constexpr int a0 = G::unit_at(t, G::OM1, 0), a1 = G::unit_at(t, G::OM1, 1);
                constexpr int b0 = G::unit_at(t, G::OM2, 0), b1 = G::unit_at(t, G::OM2, 1);
                constexpr int c0 = G::unit_at(t, G::OC, 0), c1 = G::unit_at(t, G::OC, 1);
                constexpr int w0 = G::chunk_written_at(t, 0), w1 = G::chunk_written_at(t, 1);
                if constexpr (w0 >= 0) dec_write(C, std::integral_constant<int, w0>{});
                if constexpr (w1 >= 0) dec_write(Nn, std::integral_constant<int, w1>{});
                if constexpr (c0 >= 0) dec_pack(std::integral_constant<int, c0>{});
                if constexpr (c1 >= 0) dec_pack(std::integral_constant<int, c1>{});
                if constexpr (a0 >= 0) dec_mul1(PC{}, std::integral_constant<int, a0>{});
                if constexpr (a1 >= 0) dec_mul1(PN_{}, std::integral_constant<int, a1>{});
                if constexpr (b0 >= 0) dec_mul2(PC{}, std::integral_constant<int, b0>{});
                if constexpr (b1 >= 0) dec_mul2(PN_{}, std::integral_constant<int, b1>{});
                if constexpr (l0 >= 0) dec_lookup(PC{}, std::integral_constant<int, l0>{});
                if constexpr (l1 >= 0) dec_lookup(PN_{}, std::integral_constant<int, l1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto main_loop = [&](auto wo) {
        const DmaCtx dc = dma_ctx();
        kstep(I0{}, std::true_type{}, wo, 0, dc);
        int j = 1;
        for (; j + 1 < nk; j += 2) {
            kstep(I1{}, std::false_type{}, wo, j, dc);
            kstep(I0{}, std::false_type{}, wo, j + 1, dc);
        }
        if (j < nk) kstep(I1{}, std::false_type{}, wo, j, dc);
    };
    if (wave == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue (k_gemm_dense, 16-bit weights): acc[f][g][r] = out[m0 + 128 wm + 16 g + (lane & 15)][n0 + 128 wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63, er16 = lane_e & 15, efq = lane_e >> 4;
    const int64_t n_base = n0 + wn * 128;
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int g = 0; g < FM; g++) {
                const int64_t m = m0 + wm * 16 * FM + 16 * g + er16, nn = n_base + 16 * f + 4 * efq;
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
    constexpr int ROWB = 264;
    char *wave_lds = smem + wave * 64 * ROWB;
    uint16_t *out = static_cast<uint16_t *>(out_v);
    const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
    const bool same_out = out_dtype == (std::is_same_v<T, f16_t> ? MBNB_F16 : MBNB_BF16);
    u32x2 bias_all[8];
    if (bias != nullptr) {
        const uint16_t *bp = reinterpret_cast<const uint16_t *>(bias);
#pragma unroll
        for (int f = 0; f < 8; f++) {
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
    auto epilogue16 = [&](auto wb_t) {
        constexpr bool WB = decltype(wb_t)::value;
        gd_static_for<FM / 4>([&](auto hh) {
            constexpr int H = decltype(hh)::value;
            const int64_t m_base = m0 + wm * 16 * FM + 64 * H;
#pragma unroll
            for (int f = 0; f < 8; f++) {
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
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(acc[f][4 * H + g][e]));
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
            const int ch = lane_e & 15;
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
                        if (m < M) store_out16_nt(reinterpret_cast<u32x4 *>(out + m * N + n), piece[p]);
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
    };
    if (bias != nullptr) epilogue16(std::true_type{});
    else epilogue16(std::false_type{});
}

}  // namespace mbnb
