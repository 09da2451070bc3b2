// gemm256s.h — k_gemm256s: the 256 x 256 x 64 fused 4-bit decode + MFMA GEMM of gemm256.h (k_gemm256p), restated without
// the ablation scaffolding and with three schedule options (template VAR, bit set):
//
//   1  RAW2   packed weights are fetched two k-steps at a time.  In k_gemm256p every lane DMAs its own 16 bytes each
//             k-step: one wave-instruction touches 32 rows x 32 B = 32 different 128-byte lines for 1 KiB of payload (the
//             activation pieces touch 8 full lines per KiB), and ablating that one instruction per k-step was worth 10 us
//             of 126.  Here a wave fetches 16 rows x 64 B per instruction (whole 64-byte sectors, half as many lines per
//             byte) every other k-step into a single 2 KiB-per-wave slot; the lanes pick their 16 bytes of both k-steps
//             out of it at once (ds_read_b128, source-side XOR swizzle -> conflict free) and the slot is refilled one
//             k-step later.  LDS: 16 KiB for the slot instead of 2 x 8 KiB.
//   2  STAG   the two waves of a SIMD (w, w + 4) run the same program in lockstep in k_gemm256p, so both want the matrix
//             pipe, then both want the VALU / LDS for the decode (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
//             Stage C of the weight image is free as soon as barrier j has been passed, so the decode of tile j+2 may sit
//             anywhere in [group 3 of step j, group 2 of step j+1]: waves 0-3 ("early") decode in groups 3 and 0, waves
//             4-7 ("late") in groups 1 and 2.  While one wave of a SIMD decodes, its partner issues bare MFMAs.
//             No extra barrier, no extra LDS; MFMA order per accumulator is unchanged -> outputs bit-identical.
//   4  ADMA   LDS-DMA issued from inline assembly (gemm_tile.h lds_dma): the compiler's LDS waits become exact
//             lgkmcnt(n) instead of lgkmcnt(0) after a "pending FLAT" access.
//
// Everything else is k_gemm256p<T, NESTED, 0, AM4 = true, BLUT = true>: same tile -> workgroup map, LDS images and swizzle,
// byte-table decode (B operand = the bits dequantize_4bit produces), absmax-by-4 fetch, fragment double buffering, one
// barrier per k-step between MFMA groups 2 and 3, LDS-staged epilogue.  Requirements (checked by the launcher):
// blocksize 64, K % 64 == 0, K_weight % 256 == 0, 16-byte aligned X / packed rows, (NESTED: blocksize2 % 4 == 0 and a
// 4-byte aligned code pointer).
#pragma once

#include "gemm256.h"

namespace mbnb {

template <bool NESTED, int VAR> constexpr int gemm256s_lds_bytes() {
    // A0 A1 B0 B1 | raw (16 KiB either way: 2 slots x 8 KiB, or one by-2 slot) | absmax-by-4 slots
    return P_RAW + 16384 + (NESTED ? 4096 : 8192);
}

template <typename T, bool NESTED, int VAR>
__global__ __launch_bounds__(512, 2) void k_gemm256s(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    constexpr bool RAW2 = (VAR & 1) != 0, STAG = (VAR & 2) != 0, ADMA = (VAR & 4) != 0, RFIRST = (VAR & 8) != 0;
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
    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0, int first, int count) {
#pragma unroll
        for (int i = first; i < first + count; i++) dma16(a_src[i] + k0, P_A + stage * P_IMG + (wave * 4 + i) * 1024);
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
    float La[8], Lb[8], Lc[8];

    // ---- fragment read offsets: per MFMA group s (chunk 2s + fh, swizzled by the row)
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
    auto tclamp = [&](int64_t t) { return t < nk ? t : nk - 1; };
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };
    const bool early = STAG && wave < 4;      // wave-uniform

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // ---- prologue: A(0), raw of tiles 0 and 1, absmax block 0 -> LDS; tile 0 decoded into stage 0; A(1) and the next
    // raw in flight.  What of tile 1 is decoded here depends on the role: see the decode plan above `kstep`.
    issue_a(0, 0, 0, 4);
    if constexpr (RAW2) {
        issue_raw2(0);
    } else {
        issue_raw(0, 0);
        issue_raw(1, kclamp(1));
    }
    issue_am4(0);
    issue_a(1, kclamp(1), 0, 4);                        // stage 1 is free: its HBM/L2 latency runs under the decode of tile 0
    MBNB_VMCNT(4);                                      // everything but A(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // byte-table writes
    __builtin_amdgcn_s_barrier();                       // raw barrier: __syncthreads() would drain A(1) as well
    asm volatile("" ::: "memory");
    if constexpr (RAW2) {
        load_rw2();
    } else {
        load_rw(P0{}, 0);
        load_rw(P1{}, 1);
    }
    load_am(P0{}, 0);
    load_am(P1{}, tclamp(1));
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float L[8];
        lookup_q(rw[0][d], L);
        finish_q(L, ram[0], d, 0);
    }
    if constexpr (STAG) {
        if (early) {   // early role: quarters 0, 1 of tile 1 are done before step 0
            lookup_q(rw[1][0], La);
            lookup_q(rw[1][1], Lb);
            finish_q(La, ram[1], 0, 1);
            finish_q(Lb, ram[1], 1, 1);
        }
    } else {
        lookup_q(rw[1][0], La);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own decode writes done, raw slots read
    if constexpr (RAW2) issue_raw2(1);      // tiles 2, 3
    else issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();  // decoded B(0) visible (each wave waited for its own LDS writes)
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    // One k-step with compile-time stage parity C: stage C holds tile j, tile T = j+1 (parity Nn) goes to stage Nn,
    // tile T2 = j+2 (parity C) to stage C once barrier j has been passed.  Groups G0 G1 G2 | barrier j | G3; a group =
    // 8 MFMAs on fragments read one group earlier.  Decode plan (L = byte-table lookups, F = products + image write):
    //   ROLE 0 (k_gemm256p):  G0 F(T,q0) L(T,q1) | G1 F(T,q1) L(T,q2) L(T,q3) | G2 F(T,q2) F(T,q3) | G3 L(T2,q0)
    //   ROLE 1 (early):       G0 L,F(T,q2) L,F(T,q3) | G1 - | G2 - | G3 L,F(T2,q0) L,F(T2,q1)
    //   ROLE 2 (late):        G0 - | G1 L,F(T,q0) L,F(T,q1) | G2 L,F(T,q2) L,F(T,q3) | G3 -
    // VMEM program order per wave and the waits before barrier j (A(j+1) and the raw of tile j+2 must have landed):
    //   per-step raw:  G3(j-1) a0 a1 | G0 a2 | G1 a3 | G2 raw(j+3) [am4]          -> vmcnt(1 [+ AMN])
    //   by-2 raw:      G3(j-1) a0 a1 | G0 a2 | G1 a3 | G2 [odd j: raw2 x2] [am4]  -> odd j: vmcnt(2 [+ AMN]), even j: vmcnt(0)
    // (am4 is issued when (j + 3) % 4 == 0, i.e. on odd steps only.)  By-2: the block of tiles j+2, j+3 is read into rw[]
    // in G3 of EVEN steps; its slot is refilled in G2 of the next (odd) step, behind an lgkmcnt(0) that the group's MFMAs
    // need anyway, and has a whole k-step to land.
    auto kstep = [&](auto cc, auto role_, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1, ROLE = decltype(role_)::value;
        using PC = std::integral_constant<int, C>;
        // ---- group 0
        read_frags(C, 1, wfB, xfB);
        if constexpr (ROLE == 1) {
            lookup_q(rw[Nn][2], La);
            lookup_q(rw[Nn][3], Lb);
        }
        if constexpr (RFIRST && ROLE == 0) {
            lookup_q(rw[Nn][1], Lb);
            __builtin_amdgcn_sched_barrier(0);   // every LDS read of the group is issued before its first MFMA
        }
        mfma_group(wfA, xfA);
        if constexpr (ROLE == 0) {
            finish_q(La, ram[Nn], 0, Nn);
            if constexpr (!RFIRST) lookup_q(rw[Nn][1], Lb);
        } else if constexpr (ROLE == 1) {
            finish_q(La, ram[Nn], 2, Nn);
            finish_q(Lb, ram[Nn], 3, Nn);
        }
        if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1
        read_frags(C, 2, wfA, xfA);
        if constexpr (ROLE == 2) {
            lookup_q(rw[Nn][0], La);
            lookup_q(rw[Nn][1], Lb);
        }
        if constexpr (RFIRST && ROLE == 0) {
            lookup_q(rw[Nn][2], La);
            lookup_q(rw[Nn][3], Lc);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_group(wfB, xfB);
        if constexpr (ROLE == 0) {
            finish_q(Lb, ram[Nn], 1, Nn);
            if constexpr (!RFIRST) {
                lookup_q(rw[Nn][2], La);
                lookup_q(rw[Nn][3], Lc);
            }
        } else if constexpr (ROLE == 2) {
            finish_q(La, ram[Nn], 0, Nn);
            finish_q(Lb, ram[Nn], 1, Nn);
        }
        if (j > 0) issue_a(Nn, kclamp(j + 1), 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 2
        const bool am_now = ((j + 3) & 3) == 0;
        if constexpr (RAW2) {
            if constexpr (C == 1) {   // odd step: refill the by-2 slot (its last readers: the load_rw2 of group 3, one step ago)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_raw2((j + 3) >> 1);      // tiles j+3, j+4: read into rw[] in group 3 of step j+1
            }
        }
        read_frags(C, 3, wfB, xfB);
        if constexpr (ROLE == 2) {
            lookup_q(rw[Nn][2], La);
            lookup_q(rw[Nn][3], Lb);
        }
        if constexpr (RFIRST) __builtin_amdgcn_sched_barrier(0);
        mfma_group(wfA, xfA);
        if constexpr (ROLE == 0) {
            finish_q(La, ram[Nn], 2, Nn);
            finish_q(Lc, ram[Nn], 3, Nn);
        } else if constexpr (ROLE == 2) {
            finish_q(La, ram[Nn], 2, Nn);
            finish_q(Lb, ram[Nn], 3, Nn);
        }
        if constexpr (!RAW2) issue_raw(Nn, kclamp(j + 3));
        if (am_now) issue_am4((j + 3) >> 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RAW2) {
            if constexpr (C == 1) {
                if (am_now) { if constexpr (NESTED) MBNB_VMCNT(4); else MBNB_VMCNT(3); } else { MBNB_VMCNT(2); }
            } else {
                MBNB_VMCNT(0);
            }
        } else {
            if (am_now) { if constexpr (NESTED) MBNB_VMCNT(3); else MBNB_VMCNT(2); } else { MBNB_VMCNT(1); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own decode writes + fragment reads done
        __builtin_amdgcn_s_barrier();                         // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 3: first fragments of stage Nn; refill stage C; raw of tile j+2 -> registers
        read_frags(Nn, 0, wfA, xfA);
        if constexpr (RAW2) {
            if constexpr (C == 0) load_rw2();            // even step: tiles j+2 (parity 0) and j+3 (parity 1)
        } else {
            load_rw(PC{}, C);
        }
        load_am(PC{}, tclamp(j + 2));
        if constexpr (ROLE == 1) {
            lookup_q(rw[C][0], La);
            lookup_q(rw[C][1], Lb);
        }
        if constexpr (RFIRST && ROLE == 0) {
            lookup_q(rw[C][0], La);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_group(wfB, xfB);
        if constexpr (ROLE == 0) {
            if constexpr (!RFIRST) lookup_q(rw[C][0], La);
        } else if constexpr (ROLE == 1) {
            finish_q(La, ram[C], 0, C);
            finish_q(Lb, ram[C], 1, C);
        }
        issue_a(C, kclamp(j + 2), 0, 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto main_loop = [&](auto role_) {
        for (int64_t j = 0; j < nk; j += 2) {
            kstep(std::integral_constant<int, 0>{}, role_, j);
            if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, role_, j + 1);
        }
    };
    if constexpr (STAG) {
        if (early) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 2>{});
    } else {
        main_loop(std::integral_constant<int, 0>{});
    }
    MBNB_VMCNT(0);

    // ---- epilogue (k_gemm256p): stage memory reused as store staging once every wave has drained its LDS reads
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
