// gemm256t.h — k_gemm256t (diagnostic, not shipped): k_gemm256s (RAW2) with the contraction on v_mfma_f32_16x16x32 instead of
// 32x32x16.  Measured 121.3 vs 120.3 us at 4096^3 (profiles/r02_gemm256_ab_16x16x32.txt) and bit-identical outputs: no gain.
//
// Why: where the chip holds its clock down under MFMA load, the 16x16x32 shape sustains a higher clock at equal cycles per
// flop (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x in bare loops on random data); the fragment bytes read from
// LDS per k-step are the same (24 ds_read_b128 per wave).  Same tile (256 x 256 x 64), wave grid (2 n x 4 m, wave tile
// 128 n x 64 m = 8 x 4 accumulators of 16 x 16), LDS images, swizzle, byte-table decode, absmax-by-4, raw-by-2, barrier
// placement and epilogue staging as k_gemm256s.  An MFMA group = 16 MFMAs = one k32 slice x four of the eight n-fragments:
//   G0 (slice 0, n 0-3) G1 (slice 0, n 4-7) G2 (slice 1, n 0-3) | barrier | G3 (slice 1, n 4-7)
// with the weight fragments of the next group and (every other group) the four activation fragments of the next slice read
// one group ahead: 2 x 16 + 2 x 16 fragment registers (k_gemm256s: 2 x 16 + 2 x 8), paid for by a decode pipeline with two
// quarters in flight instead of three and 32-bit activation offsets.  Sums differ from k_gemm256s in the last bits (k32
// instead of two k16 per MFMA); parity is against the oracle, as for every kernel.
#pragma once

#include "parked/gemm256s.h"

namespace mbnb {

template <typename T, bool NESTED>
__global__ __launch_bounds__(512, 2) void k_gemm256t(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma16<T>::frag;
    constexpr bool RAW2 = true, STAG = false, ADMA = false;
    constexpr int AMN = NESTED ? 2 : 1;                     // LDS-DMA instructions of one absmax-by-4 fetch
    constexpr int P_AM4 = P_RAW + 16384;
    constexpr int AM4_SLOT = NESTED ? 2048 : 4096;
    // byte table: entry b = (code[b & 15], code[b >> 4]) as two f32 -> one ds_read_b64 per packed byte.  STATIC LDS
    // object: its address is a compile-time constant, so a lookup needs no address add.
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

    // ---- tile -> workgroup map (as k_gemm256p): blocks b, b+8, ... share an XCD's L2 -> compact 4 (m) x 8 (n) patches
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

    {
        const int b = tid >> 1, nib = (tid & 1) ? (b >> 4) : (b & 15);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nib == i) v = (wp.qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        s_lut2[tid] = v;
    }

    // LDS-DMA: `off` = wave-uniform byte offset of the wave's 1 KiB (or 256 B) landing zone inside the dynamic region
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    auto dma16 = [&](const void *g, int off) {
        if constexpr (ADMA) lds_dma<16>(g, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off)));
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                              (__attribute__((address_space(3))) void *)(smem + off), 16, 0, 0);
    };
    auto dma4 = [&](const void *g, int off) {
        if constexpr (ADMA) lds_dma<4>(g, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off)));
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                              (__attribute__((address_space(3))) void *)(smem + off), 4, 0, 0);
    };

    // ---- activation pieces: wave w moves pieces 4w..4w+3 (8 rows x 128 B each), bank swizzle on the source address
    // (byte offsets from the tile's first row as 32-bit values: 256 rows x K x 2 B < 2^32 is checked by the launcher)
    uint32_t a_off[4];
    const T *x_tile = X + (m0 < M ? m0 : M - 1) * K;     // uniform
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_off[i] = (uint32_t)(((m - (m0 < M ? m0 : M - 1)) * K + 8 * c) * (int64_t)sizeof(T));
    }
    auto issue_a = [&](int stage, int64_t k0, int first, int count) {
        const char *xb = reinterpret_cast<const char *>(x_tile + k0);   // uniform
#pragma unroll
        for (int i = first; i < first + count; i++) dma16(xb + a_off[i], P_A + stage * P_IMG + (wave * 4 + i) * 1024);
    };

    // ---- weight decode role of this thread: rows 32*wave .. 32*wave+31 belong to this wave; lane -> (row, k-half) so
    // that the 8 lanes of a ds_write_b128 group hit 8 different swizzled chunks (as k_gemm256p)
    const int l32 = lane & 31;
    const int b_rloc = 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);   // row inside the wave's 32
    const int b_row = 32 * wave + b_rloc;
    const int b_half = l32 >> 4;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    const int64_t row_bytes = wp.K_weight >> 1;
    const uint8_t *p_src = wp.packed + bn * row_bytes + 16 * b_half;            // per-step form
    // by-2 form: instruction i, lane l -> LDS row R = 16 i + (l >> 2), chunk position l & 3 holds source chunk
    // (l & 3) ^ ((R >> 2) & 3) of the row's 64 bytes (two k-steps)
    const uint8_t *p2_src[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int R = 16 * i + (lane >> 2);
        int64_t n = n0 + 32 * wave + R;
        n = n < N ? n : N - 1;
        p2_src[i] = wp.packed + n * row_bytes + 16 * ((lane & 3) ^ ((R >> 2) & 3));
    }
    const int64_t nblk2 = wp.K_weight >> 7;                                      // 128-k blocks per row
    auto issue_raw = [&](int rs, int64_t k0) { dma16(p_src + (k0 >> 1), P_RAW + rs * 8192 + wave * 1024); };
    auto issue_raw2 = [&](int64_t blk) {
        const int64_t b = blk < nblk2 ? blk : nblk2 - 1;
#pragma unroll
        for (int i = 0; i < 2; i++) dma16(p2_src[i] + b * 64, P_RAW + wave * 2048 + i * 1024);
    };
    // absmax-by-4 (k_gemm256p): lanes 0-31 fetch the absmax of FOUR consecutive k-steps of their row every fourth step
    int64_t am4_src_row = n0 + 32 * wave + (lane & 31);
    am4_src_row = am4_src_row < N ? am4_src_row : N - 1;
    auto issue_am4 = [&](int64_t blk) {
        const int64_t nb4 = wp.nblk >> 2;
        const int64_t b = blk < nb4 ? blk : nb4 - 1;
        if (lane < 32) {
            if constexpr (!NESTED) {
                dma16(wp.am.f32 + am4_src_row * wp.nblk + 4 * b, P_AM4 + (int)(blk & 1) * AM4_SLOT + wave * 512);
            } else {
                const int64_t ai = am4_src_row * wp.nblk + 4 * b;
                dma4(wp.am.i8 + ai, P_AM4 + (int)(blk & 1) * AM4_SLOT + wave * 128);
                dma4(wp.am.am2 + (ai >> wp.bs2_shift), P_AM4 + (int)(blk & 1) * AM4_SLOT + 1024 + wave * 128);
            }
        }
    };
    u32x4 rw[2];     // raw packed bytes of the tiles being / about to be decoded, by tile parity
    float ram[2];    // their absmax
    auto load_am = [&](auto pp, int64_t t) {       // t: (clamped) tile index
        constexpr int P = decltype(pp)::value;
        if constexpr (!NESTED) {
            ram[P] = *reinterpret_cast<const float *>(smem + P_AM4 + (int)((t >> 2) & 1) * AM4_SLOT + wave * 512 + b_rloc * 16 +
                                                      (int)(t & 3) * 4);
        } else {
            const char *slot = smem + P_AM4 + (int)((t >> 2) & 1) * AM4_SLOT + wave * 128 + b_rloc * 4;
            const uint32_t word = *reinterpret_cast<const uint32_t *>(slot);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(t & 3)));
            const float a2 = *reinterpret_cast<const float *>(slot + 1024);
            ram[P] = q * (a2 / 127.0f);  // dequantize_blockwise arithmetic (functional.py:592-594)
        }
    };
    auto load_rw = [&](auto pp, int rs) {          // per-step form: the lane's own 16 bytes of raw slot rs
        constexpr int P = decltype(pp)::value;
        rw[P] = *reinterpret_cast<const u32x4 *>(smem + P_RAW + rs * 8192 + wave * 1024 + lane * 16);
    };
    auto load_rw2 = [&]() {                        // by-2 form: both tiles of the block that has landed
#pragma unroll
        for (int t = 0; t < 2; t++)
            rw[t] = *reinterpret_cast<const u32x4 *>(smem + P_RAW + wave * 2048 + b_rloc * 64 +
                                                     (((2 * t + b_half) ^ ((b_rloc >> 2) & 3)) << 4));
    };
    int bw_off[4];   // byte offsets of this thread's 4 output chunks inside stage 0 of the B image
#pragma unroll
    for (int d = 0; d < 4; d++) bw_off[d] = P_B + swz_off(b_row, 4 * b_half + d);
    // decode of a quarter (8 k of the thread's 32): lookup = 4 x ds_read_b64 of the byte table; finish = code * absmax
    // in f32 -> RNE 16 bit (the reference's dequantize_4bit bits) -> ds_write_b128 into the weight image
    auto lookup_q = [&](uint32_t w, float (&L)[8]) {
        const char *lut2 = reinterpret_cast<const char *>(s_lut2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * j)) & 0xFFu) << 3));
            L[2 * j] = v[0];
            L[2 * j + 1] = v[1];
        }
    };
    auto finish_q = [&](const float (&L)[8], float am, int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // two scalar v_mul_f32, kept away from the SLP vectoriser: beside MFMAs a packed-f32 VALU op costs far more
            // issue time than the two scalar ops it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
            float p0, p1;
            asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(L[2 * j]), "v"(am));
            asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(L[2 * j + 1]), "v"(am));
            o[j] = pack2<T>(p0, p1);
        }
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = o;
    };
    float La[8], Lb[8];   // two quarters in flight

    // ---- fragment read offsets (16 x 16 x 32 operands: lane l = row l & 15 of the fragment, k-chunk 4 s + (l >> 4) of the
    // 128-byte row; rows 16 apart share (row >> 1) & 7, so one base per slice + immediate offsets)
    const int f16r = lane & 15, fq = lane >> 4;
    int fw[2], fx[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int f = f16r * ROW_BYTES + (((4 * s + fq) ^ ((f16r >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 64 * ROW_BYTES + f;
    }
    auto read_w = [&](int stage, int s, int half, Frag (&wf)[4]) {      // n-fragments 4 half .. 4 half + 3 of slice s
#pragma unroll
        for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const Frag *>(smem + fw[s] + stage * P_IMG + (4 * half + i) * 16 * ROW_BYTES);
    };
    auto read_x = [&](int stage, int s, Frag (&xf)[4]) {                // the four m-fragments of slice s
#pragma unroll
        for (int g = 0; g < 4; g++) xf[g] = *reinterpret_cast<const Frag *>(smem + fx[s] + stage * P_IMG + g * 16 * ROW_BYTES);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[f][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto mfma_group = [&](auto half_, const Frag (&wf)[4], const Frag (&xf)[4]) {
        constexpr int H = decltype(half_)::value;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) acc[4 * H + i][g] = Mfma16<T>::run(wf[i], xf[g], acc[4 * H + i][g]);
    };

    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;
    auto tclamp = [&](int64_t t) { return t < nk ? t : nk - 1; };
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    // ---- prologue (k_gemm256s): A(0), raw block 0, absmax block 0, A(1) -> LDS; tile 0 decoded; quarter 0 of tile 1 looked up
    issue_a(0, 0, 0, 4);
    issue_raw2(0);
    issue_am4(0);
    issue_a(1, kclamp(1), 0, 4);
    MBNB_VMCNT(4);                                      // everything but A(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // byte-table writes
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_rw2();
    load_am(P0{}, 0);
    load_am(P1{}, tclamp(1));
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float L[8];
        lookup_q(rw[0][d], L);
        finish_q(L, ram[0], d, 0);
    }
    lookup_q(rw[1][0], La);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own decode writes done, raw slot read
    issue_raw2(1);      // tiles 2, 3
    __builtin_amdgcn_s_barrier();  // decoded B(0) visible
    asm volatile("" ::: "memory");
    Frag wfA[4], wfB[4], xf0[4], xf1[4];
    read_w(0, 0, 0, wfA);
    read_x(0, 0, xf0);

    // One k-step, stage parity C (tile j in stage C, tile T = j+1 decoded into stage Nn, raw of tiles j+2/j+3 by-2):
    //   G0: MFMA(slice 0, n 0-3: wfA, xf0)  reads: wfB <- (s0, n 4-7)            decode: F(T,q0) L(T,q1)   DMA a2
    //   G1: MFMA(slice 0, n 4-7: wfB, xf0)  reads: wfA <- (s1, n 0-3), xf1 <- s1   decode: F(T,q1) L(T,q2)   DMA a3
    //   G2: MFMA(slice 1, n 0-3: wfA, xf1)  reads: wfB <- (s1, n 4-7)            decode: L(T,q3) F(T,q2) F(T,q3)  [odd j: raw2] [am4]
    //   wait, barrier j
    //   G3: MFMA(slice 1, n 4-7: wfB, xf1)  reads: wfA <- next (s0, n 0-3), xf0 <- next s0; raw / absmax of tile j+2;
    //                                        decode: L(T2,q0)   DMA a0 a1 of A(j+2)
    // Two lookup buffers: q0 -> La, q1 -> Lb, q2 -> La (free after F(q0)), q3 -> Lb (free after F(q1)).
    // VMEM order and waits are k_gemm256s's by-2 form: odd j vmcnt(2 [+ AMN]), even j vmcnt(0).
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        using PC = std::integral_constant<int, C>;
        // ---- group 0
        read_w(C, 0, 1, wfB);
        mfma_group(H0{}, wfA, xf0);
        finish_q(La, ram[Nn], 0, Nn);
        lookup_q(rw[Nn][1], Lb);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1
        read_w(C, 1, 0, wfA);
        read_x(C, 1, xf1);
        mfma_group(H1{}, wfB, xf0);
        finish_q(Lb, ram[Nn], 1, Nn);
        lookup_q(rw[Nn][2], La);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 2
        const bool am_now = ((j + 3) & 3) == 0;
        if constexpr (C == 1) {   // odd step: refill the by-2 slot (its last readers: the load_rw2 of group 3, one step ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_raw2((j + 3) >> 1);
        }
        read_w(C, 1, 1, wfB);
        lookup_q(rw[Nn][3], Lb);
        mfma_group(H0{}, wfA, xf1);
        finish_q(La, ram[Nn], 2, Nn);
        finish_q(Lb, ram[Nn], 3, Nn);
        if (am_now) issue_am4((j + 3) >> 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C == 1) {
            if (am_now) { if constexpr (NESTED) MBNB_VMCNT(4); else MBNB_VMCNT(3); } else { MBNB_VMCNT(2); }
        } else {
            MBNB_VMCNT(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own decode writes + fragment reads done
        __builtin_amdgcn_s_barrier();                         // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 3
        read_w(Nn, 0, 0, wfA);
        read_x(Nn, 0, xf0);
        if constexpr (C == 0) load_rw2();            // even step: tiles j+2 (parity 0) and j+3 (parity 1)
        load_am(PC{}, tclamp(j + 2));
        mfma_group(H1{}, wfB, xf1);
        lookup_q(rw[C][0], La);
        issue_a(C, kclamp(j + 2), 0, 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);

    // ---- epilogue: acc[f][g][r] = out[m0 + 64 wm + 16 g + (lane & 15)][n0 + 128 wn + 16 f + 4 (lane >> 4) + r]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int64_t m_base = m0 + wm * 64, n_base = n0 + wn * 128;
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int f = 0; f < 8; f++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t m = m_base + 16 * g + f16r, nn = n_base + 16 * f + 4 * fq;
                if (m >= M || nn >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[f][g][e];
                    if (bias != nullptr && nn + e < N) s += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(s));
                }
                store4(o + m * N + nn, v, nn, N);
            }
        return;
    }
    // 16-bit outputs: the wave's [64 m][128 n] tile is staged in its private 16 KiB of LDS (264-byte row pitch) and leaves as
    // 16-byte stores of whole 256-byte row segments (k_gemm256p's epilogue_staged, with this accumulator layout)
    {
        constexpr int ROWB = 264;
        char *wave_lds = smem + wave * 64 * ROWB;
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const int nl = 16 * f + 4 * fq;
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
                for (int e = 0; e < 4; e++) v[e] = to_f32(from_f32<T>(acc[f][g][e] + bv[e]));
                u32x2 pk;
                if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(v[0], v[1]), pack2<f16_t>(v[2], v[3])};
                else pk = u32x2{pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3])};
                *reinterpret_cast<u32x2 *>(wave_lds + (16 * g + f16r) * ROWB + nl * 2) = pk;
            }
        }
        uint16_t *out = static_cast<uint16_t *>(out_v);
        const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
        const int ch = lane & 15;  // 4 rows x 16 chunks of 16 B per instruction
        u32x4 piece[16];
#pragma unroll
        for (int p = 0; p < 16; p++) {
            const char *srcp = wave_lds + (p * 4 + (lane >> 4)) * ROWB + ch * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(srcp), hi = *reinterpret_cast<const u32x2 *>(srcp + 8);
            piece[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        const int64_t n = n_base + ch * 8;
        if (n >= N) return;
        if (vec_ok && n + 8 <= N) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const int64_t m = m_base + p * 4 + (lane >> 4);
                if (m < M) *reinterpret_cast<u32x4 *>(out + m * N + n) = piece[p];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const int64_t m = m_base + p * 4 + (lane >> 4);
                if (m >= M) continue;
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (n + e < N) out[m * N + n + e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
            }
        }
    }
}

}  // namespace mbnb
