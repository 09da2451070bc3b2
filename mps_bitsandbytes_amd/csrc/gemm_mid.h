// gemm_mid.h — k_gemm_mid: fused 4-bit decode + MFMA GEMM for MID-SIZED batches (too many rows for the weight-streaming
// kernels, too few output tiles of 256 x 256 to fill 256 CUs) — the reference's own native range, M <= 512
// (functional.py:714-717, mm:1987-1993; kernel role: nf4_matmul_simd / _large, mm:440-696).
//
//   out[M,N] = X[M,K] · decode(W)[N,K]^T (+ bias)        blocksize 64, K % 64 == 0, K_weight % 256 == 0
//
// Tile 128 (m) x 64 (n) x 64 (k), 8 waves (512 threads) as 2 (n) x 4 (m), wave tile 32 n x 32 m = one 32 x 32
// accumulator (four MFMAs per wave and k-step, two waves per SIMD); 74 KiB of LDS and < 128 VGPRs: two workgroups per CU
// when the grid has them.  Optional split-K (grid = tiles x slices): a slice writes its f32 partial tile through LDS as whole
// 256-byte rows into a caller workspace [slice][M][N]; k_splitk_reduce_rm adds the slices in index order
// (deterministic), then bias, one rounding to the weight dtype, cast.
//
// The k_gemm256s pipeline at a smaller scale, every global access in the loop an LDS-DMA with hand-counted vmcnt:
//   * activations: 16 pieces of 8 rows x 128 B per k-step (2 per wave) into a ring of THREE 16 KiB stages, issued two
//     k-steps ahead (a ring of five, issued four ahead, was slower: the k-step is bound by its LDS reads and its barrier,
//     not by the L2 round trip -- tools/exp/abl_mid.py); the bank swizzle chunk ^ ((row >> 1) & 7) is applied to the
//     source address;
//   * packed weights two k-steps at a time: half a piece per wave = its own 8 rows x 64 B (whole 64-byte sectors) into one
//     of two 4 KiB slots; a thread later picks the dword (8 k) of its (row, eighth) with one ds_read_b32;
//   * absmax four k-steps at a time (lanes 0-7 of a wave: 16 B of their row; double quant: the dword of four int8 codes
//     and their absmax2) into one of two 1 KiB slots;
//   * decode of tile j+1 (byte table x absmax -> RNE 16 bit: the bits dequantize_4bit produces) into weight stage
//     (j+1) & 1 while the MFMAs of tile j run; one barrier per k-step.
// All issues of a k-step stand together, so the wait before its barrier is vmcnt(number issued in this step); all LDS reads
// of a step are issued before them and the LDS-DMA goes out from inline assembly, so the compiler's LDS waits are exact
// counts and one LDS latency is paid per k-step, not one per MFMA (first version: 0.78 us per k-step).
#pragma once

#include "gemm256.h"

namespace mbnb {

constexpr int MID_NA = 3;                                // activation stages: A(j + MID_NA - 1) is issued in k-step j
constexpr int MID_A = 0, MID_A_STAGE = 16384;            // activation stages (128 rows x 128 B)
constexpr int MID_B = MID_NA * MID_A_STAGE, MID_B_STAGE = 8192;   // 2 decoded-weight stages (64 rows x 128 B)
constexpr int MID_RAW = MID_B + 2 * MID_B_STAGE, MID_RAW_SLOT = 4096;      // 2 raw slots (64 rows x 64 B = two k-steps)
constexpr int MID_AM = MID_RAW + 2 * MID_RAW_SLOT, MID_AM_SLOT = 1024;     // 2 absmax-by-4 slots
constexpr int MID_LDS = MID_AM + 2 * MID_AM_SLOT;

template <typename T, bool NESTED, int ABL = 0>   // ABL: timing-only diagnostic variants (tools/exp), 0 in the product
__global__ __launch_bounds__(512, 4) void k_gemm_mid(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     float *__restrict__ partial, int64_t M, int64_t N, int64_t K,
                                                     int slices, int64_t k_per_slice) {
    using Frag = typename Mfma<T>::frag;
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];   // byte table: (code[b & 15], code[b >> 4])
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
    const int wn = wave >> 2, wm = wave & 3;                       // wave tile: n rows [32 wn, +32), m rows [32 wm, +32)

    // ---- block -> (tile, slice): blocks b, b + 8, ... share an XCD (speed only); consecutive remapped ids walk the n-tiles
    // of one m-strip and one K slice, so an XCD's workgroups share activation rows in L2
    const int64_t tiles_m = (M + 127) >> 7, tiles_n = (N + 63) >> 6;
    const int64_t tiles = tiles_m * tiles_n, nwg = tiles * slices;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int64_t slice = bid / tiles, tile = bid % tiles;
    const int64_t tm = tile / tiles_n, tn = tile % tiles_n;
    const int64_t m0 = tm << 7, n0 = tn << 6;
    const int64_t k_begin = slice * k_per_slice;
    const int64_t k_end = (k_begin + k_per_slice < K) ? k_begin + k_per_slice : K;
    const int64_t nk = (k_end - k_begin) >> 6;     // >= 1 (the launcher never creates empty slices)

    {
        const int b = tid >> 1, nib = (tid & 1) ? (b >> 4) : (b & 15);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nib == i) v = (wp.qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        s_lut2[tid] = v;
    }

    // LDS-DMA from inline assembly (gemm_tile.h lds_dma): invisible to the compiler's wait-count pass, so its LDS waits
    // stay exact lgkmcnt(n) and the fragment reads / table lookups issued at the top of a k-step really run ahead
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    auto dma16 = [&](const void *g, int off) {
        lds_dma<16>(g, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off)));
    };
    auto dma4 = [&](const void *g, int off) {
        lds_dma<4>(g, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)off)));
    };

    // ---- activation pieces: wave w moves pieces 2w, 2w+1 (8 rows x 128 B each)
    const T *a_src[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (wave * 2 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + k_begin + 8 * c;
    }
    auto kclamp = [&](int64_t t) { return (t < nk ? t : nk - 1) << 6; };
    auto issue_a = [&](int64_t t, int i) {     // piece i of this wave's two, tile t -> stage t % MID_NA
        dma16(a_src[i] + kclamp(t), MID_A + (int)(t % MID_NA) * MID_A_STAGE + (wave * 2 + i) * 1024);
    };
    // ---- packed weights, two k-steps per piece: wave w fetches its own 8 rows x 64 B (lanes 0-31: row l >> 2, 16-byte
    // chunk l & 3) into its 512 bytes of the slot; no swizzle needed (a ds_read_b32 pass covers 4 rows x 32 B)
    const int64_t row_bytes = wp.K_weight >> 1;
    int64_t wrow = n0 + 8 * wave + ((lane & 31) >> 2);
    wrow = wrow < N ? wrow : N - 1;
    const uint8_t *p2_src = wp.packed + wrow * row_bytes + (k_begin >> 1) + 16 * (lane & 3);
    const int64_t nblk2 = (nk + 1) >> 1;     // 128-k blocks of this slice (K_weight % 256 == 0 keeps a half-used last one in bounds)
    auto issue_raw2 = [&](int64_t b) {
        const int64_t bb = b < nblk2 ? b : nblk2 - 1;
        if (lane < 32) dma16(p2_src + bb * 64, MID_RAW + (int)(b & 1) * MID_RAW_SLOT + wave * 512);
    };
    // ---- absmax, four k-steps per piece: lanes 0-7 of a wave fetch the 16 B of their row
    int64_t arow = n0 + 8 * wave + (lane & 7);
    arow = arow < N ? arow : N - 1;
    const int64_t am_idx0 = arow * wp.nblk + (k_begin >> 6);
    const int64_t nblk4 = (nk + 3) >> 2;
    auto issue_am4 = [&](int64_t b) {
        const int64_t bb = b < nblk4 ? b : nblk4 - 1;
        if (lane < 8) {
            if constexpr (!NESTED) {
                dma16(wp.am.f32 + am_idx0 + 4 * bb, MID_AM + (int)(b & 1) * MID_AM_SLOT + wave * 128);
            } else {
                const int64_t ai = am_idx0 + 4 * bb;
                dma4(wp.am.i8 + ai, MID_AM + (int)(b & 1) * MID_AM_SLOT + wave * 32);
                dma4(wp.am.am2 + (ai >> wp.bs2_shift), MID_AM + (int)(b & 1) * MID_AM_SLOT + 512 + wave * 32);
            }
        }
    };
    // ---- decode role: thread = (row 8 w + R, dword e of the k-step's 32 packed bytes: 8 k -> one 16-byte image chunk)
    const int R = lane >> 3, e8 = lane & 7;
    const int b_row = 8 * wave + R;
    const int bw_off = MID_B + swz_off(b_row, e8);
    auto load_raw = [&](int64_t t) -> uint32_t {      // tile t (relative to the slice)
        return *reinterpret_cast<const uint32_t *>(smem + MID_RAW + (int)((t >> 1) & 1) * MID_RAW_SLOT + wave * 512 + R * 64 +
                                                   (int)(t & 1) * 32 + e8 * 4);
    };
    auto load_am = [&](int64_t t) -> float {
        const int slot = MID_AM + (int)((t >> 2) & 1) * MID_AM_SLOT;
        if constexpr (!NESTED) {
            return *reinterpret_cast<const float *>(smem + slot + wave * 128 + R * 16 + (int)(t & 3) * 4);
        } else {
            const uint32_t word = *reinterpret_cast<const uint32_t *>(smem + slot + wave * 32 + R * 4);
            const float qv = (float)(int)(int8_t)(word >> (8 * (int)(t & 3)));
            const float a2 = *reinterpret_cast<const float *>(smem + slot + 512 + wave * 32 + R * 4);
            return qv * (a2 / 127.0f);   // dequantize_blockwise arithmetic (functional.py:592-594)
        }
    };
    auto lookup4 = [&](uint32_t w, float (&L)[8]) {                  // 4 packed bytes -> 8 code values
        const char *lut2 = reinterpret_cast<const char *>(s_lut2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * j)) & 0xFFu) << 3));
            L[2 * j] = v[0];
            L[2 * j + 1] = v[1];
        }
    };
    auto finish4 = [&](const float (&L)[8], float am, int stage) {   // x absmax -> RNE 16 bit -> one image chunk
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float p0, p1;
            asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(L[2 * j]), "v"(am));
            asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(L[2 * j + 1]), "v"(am));
            o[j] = pack2<T>(p0, p1);
        }
        *reinterpret_cast<u32x4 *>(smem + stage * MID_B_STAGE + bw_off) = o;
    };

    // ---- fragments (gemm_tile.h layout): MFMA group s reads chunk 2s + fh of rows fr
    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = MID_B + wn * 32 * ROW_BYTES + f;
        fx[s] = MID_A + wm * 32 * ROW_BYTES + f;
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;

    // ---- prologue: A(0), A(1), raw blocks 0 and 1 (tiles 0..3), absmax block 0 (tiles 0..3); tile 0 decoded; the raw
    // dword / absmax of tile 1 in registers
    issue_a(0, 0);
    issue_a(0, 1);
    issue_raw2(0);
    issue_raw2(1);
    issue_am4(0);
    issue_a(1, 0);
    issue_a(1, 1);
    static_assert(MID_NA == 3, "prologue / loop waits are written for two tiles of activations in flight");
    MBNB_VMCNT(2);                                      // everything but A(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // byte-table writes
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint32_t w_next;
    float am_next;
    {
        const uint32_t w = load_raw(0);
        const float am = load_am(0);
        float L0[8];
        lookup4(w, L0);
        finish4(L0, am, 0);
        const int64_t t1 = nk > 1 ? 1 : 0;
        w_next = load_raw(t1);
        am_next = load_am(t1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // decoded tile 0 visible
    asm volatile("" ::: "memory");

    // k-step j: stage j & 1 of the weight image and stage j % 3 of the activations hold tile j.  Per wave: every LDS read
    // of the step first (8 fragments of tile j, 4 byte-table lookups of tile j+1 -- its raw dword came into a register one
    // step ago -- and the raw dword / absmax of tile j+2), then 4 MFMAs with the LDS-DMA issues (A(j+2) x2, raw block
    // (j+4)/2 on even j, absmax block (j+4)/4 when j % 4 == 0) and the products / image write of tile j+1 between them;
    // vmcnt(issued in this step), lgkmcnt(0), barrier.  Raw block b (tiles 2b, 2b+1) is first read at step 2b-2 and
    // issued at step 2b-4; absmax block q (tiles 4q..4q+3) is first read at step 4q-2 and issued at step 4q-4.
    for (int64_t j = 0; j < nk; j++) {
        const int C = (int)(j & 1), Nn = C ^ 1;
        const int ast = (int)(j % MID_NA);
        Frag wf[4], xf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            wf[s] = *reinterpret_cast<const Frag *>(smem + fw[s] + C * MID_B_STAGE);
            xf[s] = *reinterpret_cast<const Frag *>(smem + fx[s] + ast * MID_A_STAGE);
        }
        float L0[8];
        lookup4(w_next, L0);
        const float am_cur = am_next;
        const int64_t t2 = j + 2 < nk ? j + 2 : nk - 1;
        w_next = load_raw(t2);
        am_next = load_am(t2);
        const bool raw_now = (j & 1) == 0, am_now = (j & 3) == 0;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) issue_a(j + MID_NA - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 2)) {
            acc = Mfma<T>::run(wf[0], xf[0], acc);
            acc = Mfma<T>::run(wf[1], xf[1], acc);
        } else {
            asm volatile("" ::"v"(wf[0]), "v"(xf[0]), "v"(wf[1]), "v"(xf[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) issue_a(j + MID_NA - 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 4)) finish4(L0, am_cur, Nn);
        else asm volatile("" ::"v"(L0[0]), "v"(L0[7]), "v"(am_cur));
        if constexpr (!(ABL & 2)) acc = Mfma<T>::run(wf[2], xf[2], acc);
        else asm volatile("" ::"v"(wf[2]), "v"(xf[2]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 8)) {
            if (raw_now) issue_raw2((j + 4) >> 1);
            if (am_now) issue_am4((j + 4) >> 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 2)) acc = Mfma<T>::run(wf[3], xf[3], acc);
        else asm volatile("" ::"v"(wf[3]), "v"(xf[3]));
        __builtin_amdgcn_sched_barrier(0);
        // A(j+1) and everything else issued in earlier steps has landed once only this step's issues are outstanding
        if constexpr (ABL & 9) { MBNB_VMCNT(0); }
        else if (am_now) { if constexpr (NESTED) MBNB_VMCNT(5); else MBNB_VMCNT(4); }
        else if (raw_now) MBNB_VMCNT(3);
        else MBNB_VMCNT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    MBNB_VMCNT(0);
    __builtin_amdgcn_s_barrier();     // LDS reusable as store staging (every wave is past its last fragment read)
    asm volatile("" ::: "memory");

    // ---- epilogue: acc[4g+e] = out[m0 + 32 wm + fr][n0 + 32 wn + 8g + 4 fh + e]
    if (partial != nullptr) {
        // f32 partial tile through LDS ([128][64] f32, 264-byte row pitch) -> whole 256-byte rows of partial[slice][M][N]
        constexpr int PITCH = 264;
        float *pout = partial + slice * M * N;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int ml = wm * 32 + fr, nl = wn * 32 + 8 * g + 4 * fh;
            *reinterpret_cast<f32x4 *>(smem + ml * PITCH + nl * 4) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
        __syncthreads();
        const int ch = tid & 15;          // 16 chunks of 16 B per row
        const int64_t n = n0 + ch * 4;
        const bool vec_ok = (N % 4 == 0);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int ml = p * 32 + (tid >> 4);
            const int64_t m = m0 + ml;
            if (m >= M || n >= N) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + ml * PITCH + ch * 16);
            if (vec_ok && n + 4 <= N) store_f32x4_wt(pout + m * N + n, v);   // write-through: gemm256.h store4_partial
            else
                for (int e = 0; e < 4; e++)
                    if (n + e < N) pout[m * N + n + e] = v[e];
        }
        return;
    }
    // direct output: bias, one rounding to the weight dtype, cast; 16-bit outputs leave through LDS as whole 128-byte rows
    float vals[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int64_t n = n0 + wn * 32 + 8 * g + 4 * fh + e;
            float sv = acc[4 * g + e];
            if (bias != nullptr) sv += to_f32(bias[n < N ? n : N - 1]);
            vals[g][e] = to_f32(from_f32<T>(sv));
        }
    if (out_dtype == MBNB_F32) {
        float *o = static_cast<float *>(out_v);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int64_t m = m0 + wm * 32 + fr, n = n0 + wn * 32 + 8 * g + 4 * fh;
            if (m >= M || n >= N) continue;
            store4(o + m * N + n, vals[g], n, N);
        }
        return;
    }
    constexpr int PITCH16 = 136;   // 128 B of outputs + 8 B pad
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int ml = wm * 32 + fr, nl = wn * 32 + 8 * g + 4 * fh;
        u32x2 pk;
        if (out_dtype == MBNB_F16) pk = u32x2{pack2<f16_t>(vals[g][0], vals[g][1]), pack2<f16_t>(vals[g][2], vals[g][3])};
        else pk = u32x2{pack2<bf16_t>(vals[g][0], vals[g][1]), pack2<bf16_t>(vals[g][2], vals[g][3])};
        *reinterpret_cast<u32x2 *>(smem + ml * PITCH16 + nl * 2) = pk;
    }
    __syncthreads();
    {
        uint16_t *o = static_cast<uint16_t *>(out_v);
        const int ch = tid & 7;           // 8 chunks of 16 B per row
        const int64_t n = n0 + ch * 8;
        const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out_v) & 15) == 0);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int ml = p * 64 + (tid >> 3);
            const int64_t m = m0 + ml;
            if (m >= M || n >= N) continue;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(smem + ml * PITCH16 + ch * 16);
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(smem + ml * PITCH16 + ch * 16 + 8);
            if (vec_ok && n + 8 <= N) {
                *reinterpret_cast<u32x4 *>(o + m * N + n) = u32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
                const uint32_t wv[4] = {lo[0], lo[1], hi[0], hi[1]};
                for (int e = 0; e < 8; e++)
                    if (n + e < N) o[m * N + n + e] = (uint16_t)(wv[e >> 1] >> (16 * (e & 1)));
            }
        }
    }
}

// out[m, n] = cast_out(round_T(sum_s partial[s][m][n] + bias[n])): slices added in index order (deterministic); row-major
// partials, four consecutive n per thread (16-byte loads when N % 4 == 0).
template <typename T, typename OutT>
__global__ __launch_bounds__(256) void k_splitk_reduce_rm(const float *__restrict__ partial, int slices, const T *__restrict__ bias,
                                                         OutT *__restrict__ out, int64_t M, int64_t N) {
    const int64_t groups_n = (N + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * groups_n) return;
    const int64_t m = t / groups_n, n = (t - m * groups_n) << 2;
    const int64_t stride = M * N;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((N & 3) == 0) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(partial + m * N + n);
        for (int sl = 1; sl < slices; sl++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(partial + (int64_t)sl * stride + m * N + n);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] = v[e];
    } else {
        for (int e = 0; e < 4; e++)
            if (n + e < N) {
                float s = partial[m * N + n + e];
                for (int sl = 1; sl < slices; sl++) s += partial[(int64_t)sl * stride + m * N + n + e];
                a[e] = s;
            }
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float s = a[e];
        if (bias != nullptr && n + e < N) s += to_f32(bias[n + e]);
        v[e] = to_f32(from_f32<T>(s));
    }
    store4(out + m * N + n, v, n, N);
}

}  // namespace mbnb
