// gemm256q.h — k_gemm256q: the 256 x 256 x 64 fused 4-bit GEMM of gemm256.h (k_gemm256p) re-cut for FOUR waves,
// one per SIMD, each owning a 128 (n) x 128 (m) block of the tile.
//
// Why: with two waves per SIMD the k-step of k_gemm256p needs about as many vector-issue cycles for its fillers
// (decode VALU, LDS reads/writes, LDS-DMA) as the MFMAs leave free, and the two waves' streams
// collide.  One wave per SIMD with 128 x 128 per wave reads a third fewer fragments from LDS per MFMA
// (8 ds_read_b128 per 16 MFMAs instead of 6 per 8), keeps all fillers in the shadow of the wave's own MFMAs, and
// has 512 registers (256 accumulators + 256).  Same LDS images, swizzle, LDS-DMA staging, byte-table decode
// (bit-identical B operand) and absmax-by-4 fetch as k_gemm256p; requires blocksize 64 and K_weight % 256 == 0.
//
// Per wave and k-step: 64 MFMAs (4 groups of 16), 32 fragment reads, 8 activation DMA pieces, 2 raw pieces,
// 8 decode quarters (2 passes of 32 rows x 2 halves).  Pipeline of tile j+1's decode inside step j:
//   group 3 of step j-1: raw(j+1) -> registers, lookups of quarters 0,1
//   group 0: products of 0,1 | lookups 2,3,4      group 1: products 2,3,4 | lookups 5,6,7      group 2: products 5,6,7
// and one barrier per k-step between groups 2 and 3, as in k_gemm256p.
//
// STATUS (round 1): correct (parity tests pass with MBNB_Q4W=1 in a -DMBNB_ABLATION build) but not yet usable: with all
// 256 AGPRs holding accumulators hipcc spills ~280 loop-spanning VGPRs (addresses, offsets) to scratch and reloads
// them inside the k-step (50 scratch loads per two k-steps, each behind a vmcnt wait): 579 us vs 123 us for
// k_gemm256p.  The loop itself peaks at only 168 VGPRs; the spills come from live-range splitting around the
// prologue / epilogue peaks.  Next: per-phase kernels-within-a-kernel (noinline prologue / epilogue) or inline-asm
// pinned address registers.
#pragma once
#include "gemm256.h"

namespace mbnb {

template <typename T, bool NESTED>
__global__ __launch_bounds__(256, 1) void k_gemm256q(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    constexpr int RAWQ = 8192;                                   // one raw slot: 4 waves x 2 passes x 1 KiB
    constexpr int P_AM4 = P_RAW + 2 * RAWQ;
    constexpr int AM4_BLK = NESTED ? 2048 : 4096;                // one absmax block of four k-steps for 256 rows
    constexpr int AMN = NESTED ? 2 : 1;                          // LDS-DMA instructions per absmax fetch
    __shared__ __attribute__((aligned(2048))) float s_lut2[512]; // byte table: (code[b & 15], code[b >> 4])
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

    {   // byte table, 2 entries per thread
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

    // ---- activation pieces: wave w moves pieces 8w .. 8w+7 (8 rows x 128 B each), swizzle applied to the source
    // (32-bit byte offsets from the uniform base: the DMA address is SGPR base + VGPR offset, no 64-bit VALU add
    //  and half the registers; the dispatcher guarantees M * K * 2 < 2^32)
    uint32_t a_off[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = 8 * (wave * 8 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_off[i] = (uint32_t)((m * K + 8 * c) * (int64_t)sizeof(T));
    }
    auto issue_a = [&](int stage, int64_t k0, int first, int count) {
        const char *xb = reinterpret_cast<const char *>(X) + k0 * (int64_t)sizeof(T);   // uniform
#pragma unroll
        for (int i = first; i < first + count; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(xb + a_off[i]);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_A + stage * P_IMG + (wave * 8 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    // ---- weight decode role: pass p of wave w is "virtual wave" 2w + p of k_gemm256p (32 rows x 2 k-halves)
    const int l32 = lane & 31;
    const int b_half = l32 >> 4;
    int b_row[2];
    uint32_t p_off[2];   // byte offsets into the packed weight (N * K_weight / 2 < 2^32, checked by the dispatcher)
#pragma unroll
    for (int p = 0; p < 2; p++) {
        b_row[p] = 32 * (2 * wave + p) + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
        int64_t bn = n0 + b_row[p];
        bn = bn < N ? bn : N - 1;
        p_off[p] = (uint32_t)(((bn * wp.K_weight) >> 1) + 16 * b_half);
    }
    auto issue_raw = [&](int rs, int64_t k0) {
        const uint8_t *pb = wp.packed + (k0 >> 1);   // uniform
#pragma unroll
        for (int p = 0; p < 2; p++) {
            auto g = (const __attribute__((address_space(1))) void *)(pb + p_off[p]);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_RAW + rs * RAWQ + wave * 2048 + p * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };
    // absmax of four consecutive k-steps (block b = tiles 4b .. 4b+3 -> slot b & 1): lane l fetches row 64w + l
    int64_t am4_row = n0 + 64 * wave + lane;
    am4_row = am4_row < N ? am4_row : N - 1;
    const uint32_t am4_idx = (uint32_t)(am4_row * wp.nblk);   // N * nblk < 2^32
    auto issue_am4 = [&](int64_t blk) {
        const int64_t nb4 = wp.nblk >> 2;
        const int64_t b = blk < nb4 ? blk : nb4 - 1;
        if constexpr (!NESTED) {
            auto g = (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(wp.am.f32 + 4 * b) + am4_idx * 4u);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * AM4_BLK + wave * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        } else {
            const int64_t ai = (int64_t)am4_idx + 4 * b;
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.i8 + ai);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * AM4_BLK + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
            auto g2 = (const __attribute__((address_space(1))) void *)(wp.am.am2 + (ai >> wp.bs2_shift));
            auto l2 = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * AM4_BLK + 1024 + wave * 256);
            __builtin_amdgcn_global_load_lds(g2, l2, 4, 0, 0);
        }
    };
    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    // raw registers of the tile being decoded, by tile parity and pass
    u32x4 rw[2][2];
    float ram[2][2];
    auto load_raw = [&](auto pp, int64_t t_in) {
        constexpr int P = decltype(pp)::value;
        const int64_t t = t_in < nk ? t_in : nk - 1;
        const int rs = (int)(t_in & 1);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            rw[P][p] = *reinterpret_cast<const u32x4 *>(smem + P_RAW + rs * RAWQ + wave * 2048 + p * 1024 + lane * 16);
            const int row_local = b_row[p] - 64 * wave;
            if constexpr (!NESTED) {
                ram[P][p] = *reinterpret_cast<const float *>(smem + P_AM4 + (int)((t >> 2) & 1) * AM4_BLK + wave * 1024 +
                                                             row_local * 16 + (int)(t & 3) * 4);
            } else {
                const char *slot = smem + P_AM4 + (int)((t >> 2) & 1) * AM4_BLK + wave * 256 + row_local * 4;
                const uint32_t word = *reinterpret_cast<const uint32_t *>(slot);
                const float q = (float)(int)(int8_t)(word >> (8 * (int)(t & 3)));
                const float a2 = *reinterpret_cast<const float *>(slot + 1024);
                ram[P][p] = q * (a2 / 127.0f);  // dequantize_blockwise arithmetic (functional.py:592-594)
            }
        }
    };
    int bw_off[2][4];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int d = 0; d < 4; d++) bw_off[p][d] = P_B + swz_off(b_row[p], 4 * b_half + d);

    // quarter Q = 4 * pass + d: lookups (4 x ds_read_b64 from the byte table) and products
    // (code * absmax in f32 -> RNE 16 bit: the reference's dequantize_4bit bits) -> ds_write_b128
    auto lookup_q = [&](uint32_t w, float (&L)[8]) {
        const char *lut2 = reinterpret_cast<const char *>(s_lut2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * j)) & 0xFFu) << 3));
            L[2 * j] = v[0];
            L[2 * j + 1] = v[1];
        }
    };
    auto finish_q = [&](const float (&L)[8], float am, int p, int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float p0, p1;   // scalar multiplies (a packed-f32 op beside MFMAs costs more issue time than two scalar ones)
            asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(L[2 * j]), "v"(am));
            asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(L[2 * j + 1]), "v"(am));
            o[j] = pack2<T>(p0, p1);
        }
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[p][d]) = o;
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
    auto read_frags = [&](int stage, int s, Frag (&wf)[4], Frag (&xf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) wf[i] = *reinterpret_cast<const Frag *>(smem + fw[s] + stage * P_IMG + i * 32 * ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 4; j++) xf[j] = *reinterpret_cast<const Frag *>(smem + fx[s] + stage * P_IMG + j * 32 * ROW_BYTES);
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    auto mfma_group = [&](const Frag (&wf)[4], const Frag (&xf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = Mfma<T>::run(wf[i], xf[j], acc[i][j]);
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    float L[8][8];   // looked-up code pairs of the 8 quarters in flight
    // ---- prologue: stage 0 <- tile 0 (decoded in place); raw(1) in registers with quarters 0,1 looked up;
    //      A(1), raw(2) in flight
    issue_a(0, 0, 0, 8);
    issue_raw(0, 0);
    issue_raw(1, kclamp(1));
    issue_am4(0);
    MBNB_VMCNT(0);
    __syncthreads();  // byte table, A(0), raw(0), raw(1), absmax block 0 visible
    load_raw(P0{}, 0);
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
            float Lt[8];
            lookup_q(rw[0][p][d], Lt);
            finish_q(Lt, ram[0][p], p, d, 0);
            __builtin_amdgcn_sched_barrier(0);   // one quarter at a time: keeps the prologue's register peak low
        }
    load_raw(P1{}, 1);
    lookup_q(rw[1][0][0], L[0]);
    lookup_q(rw[1][0][1], L[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_a(1, kclamp(1), 0, 8);
    issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();  // decoded B(0) visible (each wave waited for its own LDS writes)
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[4], wfB[4], xfB[4];
    read_frags(0, 0, wfA, xfA);

    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        using PC = std::integral_constant<int, C>;
        // group 0
        read_frags(C, 1, wfB, xfB);
        mfma_group(wfA, xfA);
        finish_q(L[0], ram[Nn][0], 0, 0, Nn);
        finish_q(L[1], ram[Nn][0], 0, 1, Nn);
        lookup_q(rw[Nn][0][2], L[2]);
        lookup_q(rw[Nn][0][3], L[3]);
        lookup_q(rw[Nn][1][0], L[4]);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 4, 2);
        __builtin_amdgcn_sched_barrier(0);
        // group 1
        read_frags(C, 2, wfA, xfA);
        mfma_group(wfB, xfB);
        finish_q(L[2], ram[Nn][0], 0, 2, Nn);
        finish_q(L[3], ram[Nn][0], 0, 3, Nn);
        finish_q(L[4], ram[Nn][1], 1, 0, Nn);
        lookup_q(rw[Nn][1][1], L[5]);
        lookup_q(rw[Nn][1][2], L[6]);
        lookup_q(rw[Nn][1][3], L[7]);
        if (j > 0) issue_a(Nn, kclamp(j + 1), 6, 2);
        __builtin_amdgcn_sched_barrier(0);
        // group 2
        read_frags(C, 3, wfB, xfB);
        mfma_group(wfA, xfA);
        finish_q(L[5], ram[Nn][1], 1, 1, Nn);
        finish_q(L[6], ram[Nn][1], 1, 2, Nn);
        finish_q(L[7], ram[Nn][1], 1, 3, Nn);
        issue_raw(Nn, kclamp(j + 3));
        const bool am_now = ((j + 3) & 3) == 0;
        if (am_now) issue_am4((j + 3) >> 2);
        __builtin_amdgcn_sched_barrier(0);
        // everything but what this group just issued has landed: A(j+1), raw(j+2), older absmax blocks
        if (am_now) { if constexpr (NESTED) MBNB_VMCNT(4); else MBNB_VMCNT(3); } else { MBNB_VMCNT(2); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own decode writes + fragment reads done
        __builtin_amdgcn_s_barrier();                         // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // group 3: first fragments of stage Nn; refill stage C; raw(j+2) -> registers; look up its quarters 0,1
        read_frags(Nn, 0, wfA, xfA);
        load_raw(PC{}, j + 2);
        mfma_group(wfB, xfB);
        lookup_q(rw[C][0][0], L[0]);
        lookup_q(rw[C][0][1], L[1]);
        issue_a(C, kclamp(j + 2), 0, 4);
        __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(AMN >= 1, "");
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);

    // ---- epilogue: all waves are past their last barrier-protected LDS read; one more barrier makes the stage
    // memory reusable as store staging (16.5 KiB per wave, two 64-row halves one after the other)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // lane-derived values recomputed from an opaque copy of the thread index, so that nothing of the epilogue's
    // address arithmetic is hoisted above (and kept alive across) the main loop
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane_e = tid2 & 63;
    if (out_dtype != MBNB_F32) {
        char *wave_lds = smem + wave * 64 * 264;
        if (out_dtype == MBNB_F16) {
            epilogue_staged<T, f16_t, 4, 0>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 128, n0 + wn * 128, lane_e);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads of the first half done before it is rewritten
            epilogue_staged<T, f16_t, 4, 2>(acc, wave_lds, bias, static_cast<f16_t *>(out_v), M, N, m0 + wm * 128 + 64, n0 + wn * 128, lane_e);
        } else {
            epilogue_staged<T, bf16_t, 4, 0>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 128, n0 + wn * 128, lane_e);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            epilogue_staged<T, bf16_t, 4, 2>(acc, wave_lds, bias, static_cast<bf16_t *>(out_v), M, N, m0 + wm * 128 + 64, n0 + wn * 128, lane_e);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t m = m0 + wm * 128 + j * 32 + (lane_e & 31);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t nn = n0 + wn * 128 + i * 32 + 8 * g + 4 * (lane_e >> 5);
                if (m >= M || nn >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[i][j][4 * g + e];
                    if (bias != nullptr && nn + e < N) s += to_f32(bias[nn + e]);
                    v[e] = to_f32(from_f32<T>(s));
                }
                store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
}

template <bool NESTED> constexpr int gemm256q_lds_bytes() { return P_RAW + 2 * 8192 + 2 * (NESTED ? 2048 : 4096); }

}  // namespace mbnb
