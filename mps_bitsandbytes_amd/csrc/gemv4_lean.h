// gemv4_lean.h — k_gemv4_lean: the M = 1 decode shape of matmul_4bit (BASELINE configs[1]; reference Metal GEMV mm:440-560) with the
// per-wave fixed cost of k_gemv4 (gemv4.h) cut away.
//
// k_gemv4 at 4096 x 4096 is not HBM-bound and not latency-bound: it is INSTRUCTION-bound.  One wave per weight row = 16 waves per
// CU, 627 instructions per wave (llvm-objdump of the M = 1 instantiation), of which only 225 are the decode of the lane's 64
// weights (64 v_bfe + 64 ds_read_b32 + 32 v_pk_mul + 32 v_cvt_pk + 32 v_dot2); 16 x 627 / 4 SIMDs x ~4.5 cycles = 11 k cycles
// = the 4.6 us the kernel takes on a cache-resident layer.  The other 400: a 16-way select chain that builds the
// code table (70), 64-bit address arithmetic per load (40), a next-trip register ring that a 4096-wide row never uses (47
// moves), bpermute-based reduction, bookkeeping.  Here, for K = 2048 KU (KU = 1, 2, 4: every chunk of the row is requested up
// front, no ring):
//   * the weight row and its absmax are addressed through per-wave buffer descriptors (scalar base = the wave's row, per-lane
//     32-bit offset = 16 lane, chunk in the instruction offset): no 64-bit VALU;
//   * the code table comes from constant memory (one load + one LDS write by 16 lanes);
//   * the activations go to LDS by LDS-DMA (as k_gemv4), everything is requested before the first wait;
//   * the wave reduction runs on DPP row shifts / broadcasts (no LDS round trips).
// Same arithmetic per weight as k_gemv4 (table x absmax in f32, RNE to 16 bit, v_dot2 f32 accumulation; lane-order of the partial
// sums differs in the reduction tree only).  Requirements: M = 1, blocksize 64 (plain or double-quantised absmax), K % 64 == 0,
// K <= 2048 KU (KU = 1, 2, 3, 4, 6, 8: a row whose last chunk is partial reads zeros past its end -- the descriptors' range checks --
// and the activations past K land in LDS as zeros, by the same check), K_weight == K, 16-bit weights, 16-byte aligned rows; the
// double-quantised form also nblk % 4 == 0 (the codes are fetched as aligned dwords).
#pragma once
#include "gemv4.h"

namespace mbnb {

#ifndef GV_ABL
#define GV_ABL 0    // diagnostic builds (tools/exp/abl_gemv.py; timing only, results are wrong): 1 no table lookups / products (the packed dword is the
                    // operand), 2 no activation reads from LDS, 4 no decode loop at all (loads, wait, reduce), 8 no weight / absmax loads either
#endif

__device__ __forceinline__ float dpp_wave_sum(float v) {
    // row_shr 1, 2, 3 -> each lane holds the sum of up to 4 predecessors in its row of 16; then the classic row reduction
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));   // row_shr:8
    // lane 15 of each row now holds the row's sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast:15 -> rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));   // row_bcast:31 -> rows 2, 3
    return v;   // lane 63 holds the wave's sum
}

template <typename T, typename OutT, int QT, bool NESTED, int KU>
__global__ __launch_bounds__(256) void k_gemv4_lean(const T *__restrict__ X, const uint8_t *__restrict__ packed, AbsmaxView am,
                                                   const T *__restrict__ bias, OutT *__restrict__ out, int64_t N, int64_t K) {
    __shared__ float lut[16];
    extern __shared__ __attribute__((aligned(16))) char xs[];      // the activation row: K * 2 bytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const bool live = n < N;
    n = live ? n : N - 1;
    const int64_t nblk = K >> 6, row_bytes = K >> 1;

    // ---- activations -> LDS (1 KiB per wave-instruction), then every weight / absmax request of the row, then the table
    {
        // buffer form: bytes past the K activations read as zeros without touching memory (the row's last chunk may be partial)
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        const uint64_t px = reinterpret_cast<uint64_t>(X);
        i32x4_t rs_x = i32x4_t{(int)(uint32_t)px, (int)(uint32_t)(px >> 32), (int)(K * 2), 0x00020000};
#pragma unroll
        for (int e = 0; e < 4; e++) rs_x[e] = __builtin_amdgcn_readfirstlane(rs_x[e]);
        const uint32_t xs_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)xs;
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(xs_base + 4096u * u + 1024u * wave));
            const int vo = 4096 * u + tid * 16;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(vo), "s"(rs_x) : "memory", "m0");
        }
    }
    // per-wave buffer descriptors: the row's packed bytes, its absmax (NESTED: its codes, and the absmax2 array); n is wave-uniform
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(packed + n * row_bytes), 0, (int)row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = NESTED ? __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(am.i8 + n * nblk), 0, (int)nblk, 0x00020000)
                                               : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(am.f32 + n * nblk), 0, (int)(nblk * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(NESTED ? am.am2 : am.f32), 0, 0x7FFFFFFC, 0x00020000);
    const int bs2_shift = NESTED ? __builtin_ctz((unsigned)am.bs2) : 0;
    u32x4 wq[KU];
    float a[KU];
    int aq[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) {
        // lane l: packed bytes 16 l .. + 15 of chunk u (k = 2048 u + 32 l .. + 31), one absmax (block 32 u + l / 2)
#if GV_ABL & 8
        wq[u] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
        a[u] = 1.0f;
        continue;
#endif
        wq[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, u * 1024, 2));    // aux 2: nt
        if constexpr (!NESTED) {
            a[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, (lane >> 1) * 4, u * 128, 0));
        } else {
            const int bi = 32 * u + (lane >> 1);
            aq[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, bi & ~3, 0, 0);        // the aligned dword that holds code bi
            const int64_t gi = n * nblk + (bi < (int)nblk ? bi : (int)nblk - 1);     // blocks past the row: code 0 (range check), any valid absmax2
            a[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m2, (int)((gi >> bs2_shift) << 2), 0, 0));
        }
    }
    if (tid < 16) lut[tid] = (QT == MBNB_NF4 ? g_nf4_tab : g_fp4_tab)[tid];
    if constexpr (NESTED) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int bi = 32 * u + (lane >> 1);
            const float q = (float)(int)(int8_t)(aq[u] >> (8 * (bi & 3)));
            a[u] = q * (a[u] / 127.0f);          // dequantize_blockwise's arithmetic (functional.py:592-594)
        }
    }
    __syncthreads();     // table and activations in LDS (the barrier's fence waits for this wave's loads: all of them are needed now anyway)

    float acc = 0.0f;
    const char *lutb = reinterpret_cast<const char *>(lut);
#if GV_ABL & 4
#pragma unroll
    for (int u = 0; u < KU; u++) acc += __builtin_bit_cast(float, wq[u][0] ^ wq[u][1] ^ wq[u][2] ^ wq[u][3]) * a[u];
#else
#pragma unroll
    for (int u = 0; u < KU; u++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t w = wq[u][c];
#if GV_ABL & 2
            const u32x4 xq = u32x4{w, w + 1, w + 2, w + 3};
#else
            const u32x4 xq = *reinterpret_cast<const u32x4 *>(xs + (2048 * u + 32 * lane + 8 * c) * 2);
#endif
#if GV_ABL & 1
#pragma unroll
            for (int j = 0; j < 4; j++) acc = Dot2<T>::run(w + j, xq[j], acc);
            continue;
#endif
            const uint32_t wo = w & 0xF0F0F0F0u;
            const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float lo = *reinterpret_cast<const float *>(lutb + bfe_u32(we, 8 * j, 8));
                const float hi = *reinterpret_cast<const float *>(lutb + bfe_u32(wo, 8 * j + 2, 6));
                const f32x2 pr = f32x2{lo, hi} * f32x2{a[u], a[u]};      // two IEEE f32 products
                acc = Dot2<T>::run(pack2<T>(pr[0], pr[1]), xq[j], acc);
            }
        }
    }
#endif
    const float s = dpp_wave_sum(acc);
    if (lane == 63 && live) {
        const float v = s + (bias ? to_f32(bias[n]) : 0.0f);
        out[n] = from_f32<OutT>(to_f32(from_f32<T>(v)));
    }
}

}  // namespace mbnb
