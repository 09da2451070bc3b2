// gemm256.h — 256 x 256 x 64 fused decode + MFMA GEMM for large M (the MFMA-bound regime).
//
//   out[M,N] = X[M,K] · decode(W)[N,K]^T (+ bias)
//
// One workgroup = 8 waves (512 threads) = one 256(m) x 256(n) output tile; one workgroup per CU
// (128 KiB of LDS: two stages of {activation tile 32 KiB, decoded weight tile 32 KiB}).
//   * activations: global -> LDS directly (global_load_lds_dwordx4, 4 x 1 KiB pieces per wave and
//     k-step); the LDS image is lane-linear, so the bank swizzle is applied to the SOURCE address;
//   * weights: each thread loads the 16 packed bytes (32 k) + absmax of one weight row one k-step
//     ahead into VGPRs, decodes them (LDS code table x absmax -> RNE 16-bit: the reference's
//     dequantize_4bit bits) and writes 4 x 16 B into the next stage's weight image, one quarter
//     between each group of MFMAs of the current stage, so VALU decode and MFMA overlap;
//   * one barrier per k-step; tiles at the ragged M / N edges clamp their row index (no
//     predicated loads), the epilogue masks the stores.
// LDS images and fragment reads are those of gemm_tile.h (128-byte rows, chunk ^ ((row>>1)&7)).
// Wave grid 2 (n) x 4 (m): each wave owns 128 (n) x 64 (m) of out^T as 4 x 2 accumulators of
// v_mfma_f32_32x32x16 (weight tile = MFMA A operand, activation tile = B operand).
#pragma once

#include <type_traits>

#include "gemm_tile.h"

namespace mbnb {
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void fill_code_lut_rt(float *lut, int tid, int qt) {
    if (tid < 16) {
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (tid == i) v = (qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        lut[tid] = v;
    }
}

// 4-bit producer with the code table chosen at run time (one kernel for NF4 and FP4)
template <typename T, bool NESTED> struct Q4ProducerRT {
    struct Params {
        const uint8_t *packed;
        AbsmaxView am;
        int64_t N, K_weight;
        int64_t nblk;
        int bs_shift;
        int qt;
        int bs2_shift;  // log2(absmax blocksize2) for the nested form (k_gemm256p only)
        int w8, w6;     // the constants 8 and 6, passed at run time so hipcc keeps v_bfe_u32 (it folds a
                        // constant-width field extract into shift + and, one VALU instruction more)
    };
    struct Regs {
        u32x4 w;
        float am;
    };
    static __device__ __forceinline__ void init_lut(float *lut, int tid, const Params &p) { fill_code_lut_rt(lut, tid, p.qt); }
    // unconditional loads: n and k are already clamped in range by the caller
    static __device__ __forceinline__ void fetch(const Params &p, int64_t n, int64_t k, Regs &r) {
        r.w = *reinterpret_cast<const u32x4 *>(p.packed + ((n * p.K_weight + k) >> 1));
        r.am = load_absmax<NESTED>(p.am, n * p.nblk + (k >> p.bs_shift));
    }
    // quarter d (8 of the thread's 32 k) -> one 16-byte chunk
    static __device__ __forceinline__ void emit_quarter(const Regs &r, int d, const float *lut, char *tile, int row, int chunk) {
        const uint32_t w = r.w[d];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float lo = lut[(w >> (8 * j)) & 15] * r.am;
            const float hi = lut[(w >> (8 * j + 4)) & 15] * r.am;
            o[j] = pack2<T>(lo, hi);
        }
        *reinterpret_cast<u32x4 *>(tile + swz_off(row, chunk)) = o;
    }
};

// int8 rowwise weights (Linear8bit)
template <typename T, int WF = W8_INT8> struct I8ProducerRT {
    struct Params {
        const int8_t *w;
        const float *scales;
        int64_t N, K_weight;
    };
    struct Regs {
        u32x4 w[2];
        float s;
    };
    static __device__ __forceinline__ void init_lut(float *, int, const Params &) {}
    static __device__ __forceinline__ void fetch(const Params &p, int64_t n, int64_t k, Regs &r) {
        r.s = w8_row_scale<WF>(p.scales[n]);
        r.w[0] = *reinterpret_cast<const u32x4 *>(p.w + n * p.K_weight + k);
        r.w[1] = *reinterpret_cast<const u32x4 *>(p.w + n * p.K_weight + k + 16);
    }
    static __device__ __forceinline__ void emit_quarter(const Regs &r, int d, const float *, char *tile, int row, int chunk) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = r.w[d >> 1][2 * (d & 1) + (j >> 1)];
            const float q0 = w8_decode_sel<WF>(w, 2 * (j & 1));
            const float q1 = w8_decode_sel<WF>(w, 2 * (j & 1) + 1);
            o[j] = pack2<T>(q0 * r.s, q1 * r.s);
        }
        *reinterpret_cast<u32x4 *>(tile + swz_off(row, chunk)) = o;
    }
};

template <typename OutT>
__device__ __forceinline__ void store4(OutT *o, const float (&v)[4], int64_t n, int64_t N) {
    if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(o) & (4 * sizeof(OutT) - 1)) == 0)) {
        if constexpr (sizeof(OutT) == 2)
            *reinterpret_cast<u32x2 *>(o) = u32x2{pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3])};
        else
            *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (n + e < N) o[e] = from_f32<OutT>(v[e]);
    }
}

// Split-K partials (f32): write-through stores -- the next launch (the slice reduction) reads them on other XCDs, and nothing is left for the
// end of the launch to drain: 0.2-1.3 us per call on the split shapes of k_gemm_small (96 x 4096^2 15.5 -> 14.8 us, 256 x 4096 x 11008
// 45.1 -> 43.8; two library builds on one box, profiles/r03_small_partials_write_through_ab.txt).
__device__ __forceinline__ void store4_partial(float *o, const float (&v)[4], int64_t n, int64_t N) {
    if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
        store_f32x4_wt(o, f32x4{v[0], v[1], v[2], v[3]});
    } else {
        store4(o, v, n, N);
    }
}

constexpr int G256_A_BYTES = 256 * ROW_BYTES;
constexpr int G256_STAGE = 2 * G256_A_BYTES;
constexpr int G256_LDS = 2 * G256_STAGE + 64;

// Requirements (checked by the launcher): K % 64 == 0, K <= K_weight, X and the weight rows
// 16-byte aligned, M, N >= 1.
template <typename T, typename Producer>
__global__ __launch_bounds__(512, 2) void k_gemm256(const T *__restrict__ X, typename Producer::Params wp,
                                                    const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                    int64_t M, int64_t N, int64_t K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lut = reinterpret_cast<float *>(smem + 2 * G256_STAGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

    // ---- tile -> workgroup map.  Blocks are dealt round-robin over the 8 XCDs (speed only), so
    // blocks b, b+8, b+16, ... share an L2: give each such group a compact 4 (m) x 8 (n) patch
    // of tiles when the grid allows, so a patch's activation strips and weight strips are L2 hits.
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

    Producer::init_lut(lut, tid, wp);

    // ---- activation tile: 32 pieces of 1 KiB (8 rows x 128 B); wave w moves pieces 4w..4w+3
    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);  // inverse swizzle on the source
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(a_src[i] + k0);
            auto l = (__attribute__((address_space(3))) void *)(smem + stage * G256_STAGE + (wave * 4 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    // ---- weight tile: thread t decodes row t/2, k-half t%2
    const int b_row = tid >> 1, b_half = tid & 1;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    typename Producer::Regs breg, breg_next;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;

    // ---- prologue: stage 0 <- tile 0; registers <- packed tile 1
    issue_a(0, 0);
    Producer::fetch(wp, bn, 32 * b_half, breg);
    {
        const int64_t k1 = nk > 1 ? 64 : 0;
        Producer::fetch(wp, bn, k1 + 32 * b_half, breg_next);
    }
    __syncthreads();  // code table visible
#pragma unroll
    for (int d = 0; d < 4; d++) Producer::emit_quarter(breg, d, lut, smem + G256_A_BYTES, b_row, 4 * b_half + d);

    const int fr = lane & 31, fh = lane >> 5;
    for (int64_t kt = 0; kt < nk; kt++) {
        const int cur = (int)(kt & 1), nxt = cur ^ 1;
        __syncthreads();  // stage `cur` complete (LDS-DMA + decoded weights); stage `nxt` free
        // prefetch: activations of tile kt+1 -> stage nxt; packed weights of tile kt+2 -> registers.
        // Past the end the index is clamped: the redundant work lands in a stage nobody reads.
        const int64_t k1 = (kt + 1 < nk) ? (kt + 1) << 6 : k_last;
        const int64_t k2 = (kt + 2 < nk) ? (kt + 2) << 6 : k_last;
        issue_a(nxt, k1);
        breg = breg_next;
        Producer::fetch(wp, bn, k2 + 32 * b_half, breg_next);

        const char *As = smem + cur * G256_STAGE;
        const char *Bs = As + G256_A_BYTES;
        char *Bn = smem + nxt * G256_STAGE + G256_A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            typename Mfma<T>::frag wf[4], xf[2];
#pragma unroll
            for (int i = 0; i < 4; i++)
                wf[i] = *reinterpret_cast<const typename Mfma<T>::frag *>(Bs + swz_off(wn * 128 + i * 32 + fr, 2 * s + fh));
#pragma unroll
            for (int j = 0; j < 2; j++)
                xf[j] = *reinterpret_cast<const typename Mfma<T>::frag *>(As + swz_off(wm * 64 + j * 32 + fr, 2 * s + fh));
            Producer::emit_quarter(breg, s, lut, Bn, b_row, 4 * b_half + s);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[i], xf[j], acc[i][j]);
        }
    }

    // ---- epilogue: acc[i][j][4g+e] = out[m0 + wm*64 + j*32 + fr][n0 + wn*128 + i*32 + 8g + 4fh + e]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t n = n0 + wn * 128 + i * 32 + 8 * g + 4 * fh;
                if (m >= M || n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[i][j][4 * g + e];
                    if (bias != nullptr && n + e < N) s += to_f32(bias[n + e]);
                    v[e] = to_f32(from_f32<T>(s));  // one rounding to the compute dtype
                }
                if (out_dtype == MBNB_F16) store4(static_cast<f16_t *>(out_v) + m * N + n, v, n, N);
                else if (out_dtype == MBNB_BF16) store4(static_cast<bf16_t *>(out_v) + m * N + n, v, n, N);
                else store4(static_cast<float *>(out_v) + m * N + n, v, n, N);
            }
        }
}


// =====================================================================================
// k_gemm256p — software-pipelined 4-bit variant of k_gemm256 (same tile and wave grid).
//
// LDS map: static [code table 1K]; dynamic [A0 32K][A1 32K][B0 32K][B1 32K][raw0][raw1]  (A = activation image,
// B = decoded weight image; every image keeps the gemm_tile.h row swizzle).  A and B halves are
// 64 KiB apart so that one base VGPR + a 16-bit immediate reaches both stages of an image.
//
// Every global access in the main loop is an LDS-DMA (global_load_lds), so all vector-memory
// waits are hand-counted `s_waitcnt vmcnt(N)` and loads stay in flight across the barrier:
//   * activations A(j+2): 4 x 1 KiB pieces per wave -> stage j&1, issued right after barrier j;
//   * the thread's own packed weights + absmax for tile j+3 ("raw"): 16 B + 4 B per lane into a
//     per-wave raw slot (double-buffered), read back by the SAME lane two k-steps later -- no
//     VGPR-destination loads, hence no compiler-inserted vmcnt(0).
// One barrier per k-step, placed between MFMA groups 2 and 3 (a group = 8 MFMAs = one k16 slice):
// groups 0..2 run on fragments of stage c, the decoded weights of tile j+1 go to stage c^1 during
// groups 0/1, and group 3 runs while the first fragments of stage c^1 are already being read.
// Inside a group the MFMAs are issued FIRST, then the decode and the next group's fragment
// reads, so LDS latency and decode VALU sit in the shadow of the matrix pipe.
//
// VMEM program order per wave: ... [A(j+1) x4, raw(j+2) x R] [A(j+2) x4, raw(j+3) x R] ...
//   before decoding tile j+1 (group 0 of step j): raw(j+1) done  <=> vmcnt(4 + R)
//   before barrier j:                            A(j+1)  done  <=> vmcnt(R)
// =====================================================================================
// one v_bfe_u32 (hipcc otherwise emits shift + and for a constant-position field)
__device__ __forceinline__ uint32_t bfe_u32(uint32_t x, int off, int width) {
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(off), "n"(width));
    return r;
}

constexpr int P_A = 0, P_B = 65536, P_RAW = 131072;  // offsets inside the dynamic LDS region
constexpr int P_IMG = 32768;  // bytes per image stage

// ---------------------------------------------------------------------------------------------
// Epilogue of the 256 x 256 kernels: accumulators -> (+bias, one rounding to the compute dtype, cast)
// -> LDS -> global.  Each wave owns 64 (m) x 128 (n) outputs; in the accumulator a lane holds 4
// consecutive n of one m per register group, i.e. 8-byte pieces of 64 different rows, which as
// direct global stores touch 64 cache lines per instruction.  Staging the wave's [64][128] tile in
// its private 16 KiB of (now idle) LDS turns that into 16-byte stores of whole 256-byte row segments.
// 16-bit outputs only; fp32 output keeps the direct stores.
// ---------------------------------------------------------------------------------------------
template <typename T, typename OutT, int NJ = 2, int J0 = 0>
__device__ __forceinline__ void epilogue_staged(const f32x16 (&acc)[4][NJ], char *wave_lds, const T *__restrict__ bias,
                                                OutT *__restrict__ out, int64_t M, int64_t N, int64_t m_base,
                                                int64_t n_base, int lane) {
    static_assert(sizeof(OutT) == 2, "staged epilogue is for 16-bit outputs");
    constexpr int ROWB = 264;  // 256 B of outputs + 8 B pad: ds_write_b64 of 32 rows -> 2-way conflicts at most
    const int fr = lane & 31, fh = lane >> 5;
    // phase 1, in two straight-line versions (with / without bias: a per-element test of the pointer is a taken branch
    // per accumulator).  The four bias values of a lane's column group depend on (i, g) only: loaded once, index clamped
    // instead of guarded (the out-of-range columns are never stored).
    auto stage_tiles = [&](auto with_bias) {
        constexpr bool WB = decltype(with_bias)::value;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int nl = i * 32 + 8 * g + 4 * fh;  // local column of the first of 4 outputs
                float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (WB) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int64_t n = n_base + nl + e;
                        bv[e] = to_f32(bias[n < N ? n : N - 1]);
                    }
                }
                if constexpr (NJ > 2) __builtin_amdgcn_sched_barrier(0);   // bound the live range of the accumulator reads
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float s;
                        if constexpr (NJ > 2) {
                            // 256-accumulator kernels: pull each value out of its AGPR where it is used (otherwise the
                            // allocator copies all 256 to VGPRs at the loop exit and spills what does not fit)
                            asm("v_accvgpr_read_b32 %0, %1" : "=v"(s) : "a"(acc[i][J0 + j][4 * g + e]));
                        } else {
                            s = acc[i][J0 + j][4 * g + e];
                        }
                        if constexpr (WB) s += bv[e];
                        v[e] = to_f32(from_f32<T>(s));  // one rounding to the compute dtype
                    }
                    *reinterpret_cast<u32x2 *>(wave_lds + (j * 32 + fr) * ROWB + nl * 2) =
                        u32x2{pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3])};
                }
            }
    };
    if (bias != nullptr) stage_tiles(std::true_type{});
    else stage_tiles(std::false_type{});
    // wave-private tile: no barrier, and no wait either -- one wave's LDS instructions execute in issue order, so the
    // reads below see the writes above (the compiler waits for each read's data before the store that uses it)
    const bool vec_ok = (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // all staging reads first (unconditional, pipelined), then the guarded stores: with the bounds test in front of
    // each read every iteration would expose one LDS round trip
    const int ch = lane & 15;  // 4 rows x 16 chunks of 16 B per instruction
    u32x4 piece[16];
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const char *src = wave_lds + (p * 4 + (lane >> 4)) * ROWB + ch * 16;
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(src), hi = *reinterpret_cast<const u32x2 *>(src + 8);
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
                if (n + e < N) reinterpret_cast<uint16_t *>(out + m * N + n)[e] = (uint16_t)(piece[p][e >> 1] >> (16 * (e & 1)));
        }
    }
}

#define MBNB_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

#ifdef MBNB_ABLATION
// debug stamps (diagnostic builds only): [set][event] shader-clock values of workgroup 0
__device__ unsigned long long g_dbg_stamps[2][1024];
#endif
#define MBNB_NOP4() do {} while (0)
#define MBNB_NOP2() do {} while (0)

template <typename T, bool NESTED, int ablate = 0, bool AM4 = false, bool BLUT = false>
__global__ __launch_bounds__(512, 2) void k_gemm256p(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    // `ablate` (compile-time, debug only; -DMBNB_ABLATION builds the variants and MBNB_ABLATE selects one):
    // timing-only kernels that skip 1 = activation DMA, 2 = raw DMA, 4 = decode, 8 = MFMAs,
    // 16 = fragment reads inside the main loop.  Production kernels are ablate = 0.
    // LDS-DMA instructions per wave and k-step for the raw slot: packed + absmax (+ absmax2 when nested)
    constexpr int RAW_BYTES = 8192 + 2048 + (NESTED ? 2048 : 0);
    // the code table is a STATIC LDS object: its address is a compile-time constant, so a lookup is
    // `ds_read_b32 v, v_idx4 offset:<table>` with no address add (a table inside the dynamic region
    // costs one v_add per lookup).  256 floats keep the dynamic region 1 KiB aligned.
    __shared__ __attribute__((aligned(1024))) float s_lut[256];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

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

    fill_code_lut_rt(s_lut, tid, wp.qt);
    // byte table: entry b = (code[b & 15], code[b >> 4]) as two f32 -> one ds_read_b64 per packed byte
    __shared__ __attribute__((aligned(2048))) float s_lut2[512];
    if constexpr (BLUT) {
        const int b = tid >> 1, nib = (tid & 1) ? (b >> 4) : (b & 15);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nib == i) v = (wp.qt == MBNB_NF4) ? nf4_code(i) : fp4_code(i);
        s_lut2[tid] = v;
    }

    // ---- activation pieces: wave w moves pieces 4w..4w+3 (8 rows x 128 B each), swizzle on the source
    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0, int first = 0, int count = 4) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(a_src[i] + k0);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_A + stage * P_IMG + (wave * 4 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    // ---- weight decode role of this thread: rows 32*wave .. 32*wave+31 belong to this wave.
    // lane -> (row, k-half) chosen so that the 8 lanes of a ds_write_b128 group hit 8 different
    // swizzled chunks: lanes 0-7 even rows, 8-15 odd rows (half 0); 16-31 the same for half 1.
    const int l32 = lane & 31;
    const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
    const int b_half = l32 >> 4;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    const uint8_t *p_src = wp.packed + ((bn * wp.K_weight) >> 1) + 16 * b_half;
    const int64_t am_row = bn * wp.nblk;
    const int raw_lane = P_RAW + wave * 1024 + lane * 16;        // this lane's packed 16 B
    const int raw_am = P_RAW + 8192 + wave * 256 + lane * 4;     // absmax f32 (or the dword holding the int8 code)
    const int raw_am2 = P_RAW + 8192 + 2048 + wave * 256 + lane * 4;
    auto issue_raw = [&](int rs, int64_t k0) {
        char *base = smem + P_RAW + rs * RAW_BYTES;
        {
            auto g = (const __attribute__((address_space(1))) void *)(p_src + (k0 >> 1));
            auto l = (__attribute__((address_space(3))) void *)(base + wave * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
        const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
        if constexpr (AM4) {
            // absmax arrives separately, four k-steps at a time (issue_am4)
        } else if constexpr (!NESTED) {
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.f32 + ai);
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
        } else {
            // the aligned dword that contains int8 code `ai`, and its absmax2 (one per 2^bs2_shift codes)
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.i8 + (ai & ~(int64_t)3));
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
            auto g2 = (const __attribute__((address_space(1))) void *)(wp.am.am2 + (ai >> wp.bs2_shift));
            auto l2 = (__attribute__((address_space(3))) void *)(base + 8192 + 2048 + wave * 256);
            __builtin_amdgcn_global_load_lds(g2, l2, 4, 0, 0);
        }
    };
    // AM4 (blocksize 64, plain f32 absmax, K_weight % 256 == 0): a per-lane 4-byte absmax DMA every k-step pulls
    // one 64-byte sector per lane -- 512 sector requests per CU and k-step for 2 KiB of data, as many as the
    // whole activation tile.  Instead lanes 0-31 of a wave fetch 16 B = the absmax of FOUR consecutive
    // k-steps of their row every fourth step (block b = tiles 4b..4b+3 -> slot b & 1).
    constexpr int P_AM4 = P_RAW + 2 * RAW_BYTES;
    int64_t am4_src_row = n0 + 32 * wave + (lane & 31);  // the row whose absmax this lane fetches
    am4_src_row = am4_src_row < N ? am4_src_row : N - 1;
    auto issue_am4 = [&](int64_t blk) {
        const int64_t nb4 = wp.nblk >> 2;
        const int64_t b = blk < nb4 ? blk : nb4 - 1;
        if (lane < 32) {
            if constexpr (!NESTED) {
                auto g = (const __attribute__((address_space(1))) void *)(wp.am.f32 + am4_src_row * wp.nblk + 4 * b);
                auto l = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * 4096 + wave * 512);
                __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
            } else {
                // double-quantised absmax: the dword holding the row's four int8 codes of this block (nblk % 4 == 0
                // keeps it aligned) and their absmax2 (one value: 4 | blocksize2, so the four codes share it)
                const int64_t ai = am4_src_row * wp.nblk + 4 * b;
                auto g = (const __attribute__((address_space(1))) void *)(wp.am.i8 + ai);
                auto l = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * 2048 + wave * 128);
                __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
                auto g2 = (const __attribute__((address_space(1))) void *)(wp.am.am2 + (ai >> wp.bs2_shift));
                auto l2 = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * 2048 + 1024 + wave * 128);
                __builtin_amdgcn_global_load_lds(g2, l2, 4, 0, 0);
            }
        }
    };
    // raw registers of the tile being decoded, by tile parity
    u32x4 rw[2];
    float ram[2];
    auto load_raw = [&](auto pp, int rs, int64_t k0) {
        constexpr int P = decltype(pp)::value;
        const char *base = smem + rs * RAW_BYTES;
        rw[P] = *reinterpret_cast<const u32x4 *>(base + raw_lane);
        if constexpr (AM4 && !NESTED) {
            const int64_t t = k0 >> 6;  // (clamped) tile index
            ram[P] = *reinterpret_cast<const float *>(smem + P_AM4 + (int)((t >> 2) & 1) * 4096 + wave * 512 +
                                                      (b_row - 32 * wave) * 16 + (int)(t & 3) * 4);
        } else if constexpr (AM4) {
            const int64_t t = k0 >> 6;
            const char *slot = smem + P_AM4 + (int)((t >> 2) & 1) * 2048 + wave * 128 + (b_row - 32 * wave) * 4;
            const uint32_t word = *reinterpret_cast<const uint32_t *>(slot);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(t & 3)));
            const float a2 = *reinterpret_cast<const float *>(slot + 1024);
            ram[P] = q * (a2 / 127.0f);  // dequantize_blockwise arithmetic (functional.py:592-594)
        } else if constexpr (!NESTED) {
            ram[P] = *reinterpret_cast<const float *>(base + raw_am);
        } else {
            const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
            const uint32_t word = *reinterpret_cast<const uint32_t *>(base + raw_am);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(ai & 3)));
            const float a2 = *reinterpret_cast<const float *>(base + raw_am2);
            ram[P] = q * (a2 / 127.0f);  // dequantize_blockwise arithmetic (functional.py:592-594)
        }
    };
    int bw_off[4];  // byte offsets of this thread's 4 output chunks inside stage 0 of the B image
#pragma unroll
    for (int d = 0; d < 4; d++) bw_off[d] = P_B + swz_off(b_row, 4 * b_half + d);
    // decode of a quarter (8 k) is split in two halves issued one MFMA group apart, so the table
    // lookups' LDS latency is covered by 8 MFMAs instead of being waited for in place:
    //   lookup_q: byte offsets 4*idx with one v_bfe_u32 per nibble (odd nibbles: 6-bit field at 8j+2 of
    //             w & 0xF0F0F0F0; even: byte j of (w << 2) & 0x3C3C3C3C), then 8 ds_read_b32
    //   finish_q: value = code * absmax in f32 -> RNE 16-bit (the reference's dequantize_4bit bits) -> ds_write_b128
    uint32_t dbg_sink = 0;
    auto lookup_q = [&](uint32_t w, float (&L)[8]) {
        if constexpr (ablate & 65536) {
#pragma unroll
            for (int j = 0; j < 8; j++) L[j] = __builtin_bit_cast(float, w + (uint32_t)j);
            return;
        }
        if constexpr (BLUT) {
            const char *lut2 = reinterpret_cast<const char *>(s_lut2);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(lut2 + (((w >> (8 * j)) & 0xFFu) << 3));
                L[2 * j] = v[0];
                L[2 * j + 1] = v[1];
            }
            return;
        }
        const uint32_t wo = w & 0xF0F0F0F0u;
        const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
        const char *lutb = reinterpret_cast<const char *>(s_lut);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            L[2 * j] = *reinterpret_cast<const float *>(lutb + __builtin_amdgcn_ubfe(we, 8 * j, wp.w8));
            L[2 * j + 1] = *reinterpret_cast<const float *>(lutb + __builtin_amdgcn_ubfe(wo, 8 * j + 2, wp.w6));
        }
    };
    auto finish_q = [&](const float (&L)[8], float am, int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (ablate & 65536) {
                o[j] = __builtin_bit_cast(uint32_t, L[2 * j]) ^ __builtin_bit_cast(uint32_t, am);
            } else if constexpr (BLUT) {
                // two scalar v_mul_f32, kept out of the SLP vectoriser's hands: beside MFMAs a packed-f32 VALU
                // op costs far more issue time than the two scalar ops it replaces (MI355X_MICROARCH.md,
                // "price of one filler beside MFMAs")
                float p0, p1;
                asm("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(L[2 * j]), "v"(am));
                asm("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(L[2 * j + 1]), "v"(am));
                o[j] = pack2<T>(p0, p1);
            } else {
                o[j] = pack2<T>(L[2 * j] * am, L[2 * j + 1] * am);
            }
        }
        if constexpr (ablate & 32768) { dbg_sink ^= o[0] ^ o[1] ^ o[2] ^ o[3]; return; }
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = o;
    };
    float La[8], Lb[8], Lc[8];   // looked-up code values in flight: quarters (0 then 2), 1, 3
    float L01[2][8], L23[2][8];  // (VALU-decode debug variant only)
    // interleave directive for one MFMA group: per MFMA `nv` VALU and `nr` LDS reads (+ `nw` LDS writes on
    // the last MFMAs, `nm` LDS-DMA issues in the middle).  Within one wave non-MFMA instructions issue in
    // the shadow of the wave's own MFMAs only when they sit between them; clustered runs serialise with
    // the SIMD partner's MFMAs (tools/coexec_probe.hip).
    auto interleave = [&](auto nv_, auto nr_, auto nw_, auto nm_) {
        constexpr int nv = decltype(nv_)::value, nr = decltype(nr_)::value, nw = decltype(nw_)::value, nm = decltype(nm_)::value;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
            if (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
            if (nm > 0 && r >= 2 && r < 2 + nm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (nw > 0 && r >= 8 - nw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    // ---- ablate & 2048: table-free decode on the VALU (no LDS lookups).  Per thread and tile the 16
    // scaled code values RNE16(code[i] * absmax) are built once (16 v_mul + 8 cvt_pk) and split into
    // byte planes TL/TH (low / high bytes of entries 4q..4q+3); a nibble is then looked up with
    // v_perm_b32: entries 0-7 and 8-15 by its low 3 bits, merged by bit 3, planes re-interleaved.
    uint32_t TL[4], TH[4];
    auto build_table = [&](float am) {
        uint32_t Tp[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float c0 = (wp.qt == MBNB_NF4) ? nf4_code(2 * r) : fp4_code(2 * r);
            const float c1 = (wp.qt == MBNB_NF4) ? nf4_code(2 * r + 1) : fp4_code(2 * r + 1);
            Tp[r] = pack2<T>(c0 * am, c1 * am);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            TL[q] = __builtin_amdgcn_perm(Tp[2 * q + 1], Tp[2 * q], 0x06040200u);
            TH[q] = __builtin_amdgcn_perm(Tp[2 * q + 1], Tp[2 * q], 0x07050301u);
        }
    };
    auto decode_q_valu = [&](uint32_t w, int d, int stage) {
        const uint32_t o = w >> 4;
        const uint32_t selLE = w & 0x07070707u, selLO = o & 0x07070707u;
        const uint32_t sel3E = ((w >> 1) & 0x04040404u) | 0x03020100u;
        const uint32_t sel3O = ((o >> 1) & 0x04040404u) | 0x03020100u;
        auto plane = [&](const uint32_t (&P)[4], uint32_t selL, uint32_t sel3) {
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(P[3], P[2], selL), __builtin_amdgcn_perm(P[1], P[0], selL), sel3);
        };
        const uint32_t LE = plane(TL, selLE, sel3E), HE = plane(TH, selLE, sel3E);
        const uint32_t LO = plane(TL, selLO, sel3O), HO = plane(TH, selLO, sel3O);
        const uint32_t E0 = __builtin_amdgcn_perm(HE, LE, 0x05010400u), E1 = __builtin_amdgcn_perm(HE, LE, 0x07030602u);
        const uint32_t O0 = __builtin_amdgcn_perm(HO, LO, 0x05010400u), O1 = __builtin_amdgcn_perm(HO, LO, 0x07030602u);
        u32x4 out;
        out[0] = __builtin_amdgcn_perm(O0, E0, 0x05040100u);
        out[1] = __builtin_amdgcn_perm(O0, E0, 0x07060302u);
        out[2] = __builtin_amdgcn_perm(O1, E1, 0x05040100u);
        out[3] = __builtin_amdgcn_perm(O1, E1, 0x07060302u);
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = out;
    };

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
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // ---- prologue: stage 0 <- tile 0; raw(1) in registers with quarters 0,1 looked up; A(1), raw(2) in flight
    issue_a(0, 0);
    issue_raw(0, 0);
    issue_raw(1, kclamp(1));
    if constexpr (AM4) issue_am4(0);
    MBNB_VMCNT(0);
    __syncthreads();  // code table, A(0) and this wave's raw(0), raw(1) visible
    load_raw(P0{}, 0, 0);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float L[8];
        lookup_q(rw[0][d], L);
        finish_q(L, ram[0], d, 0);
    }
    load_raw(P1{}, 1, kclamp(1));
    lookup_q(rw[1][0], La);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_a(1, kclamp(1));
    issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();  // decoded B(0) visible (each wave waited for its own LDS writes)
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    int dbg_n = 0;
    auto stamp = [&]() {
#ifdef MBNB_ABLATION
        if constexpr (ablate & 512) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && dbg_n < 1024) g_dbg_stamps[wave >> 2][dbg_n] = t;
            dbg_n++;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    // one k-step with compile-time stage parity C (stage C holds tile j; tile j+1, parity Nn, is decoded
    // into stage Nn: its raw registers and quarters 0,1 lookups were issued in group 3 of the previous step)
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        using PC = std::integral_constant<int, C>;
        // Decode pipeline of tile j+1 (raw registers parity Nn), one quarter = 8 k:
        //   lookup:  q0 @ group 3 of step j-1 | q1 @ group 0 | q2, q3 @ group 1
        //   finish:  q0 @ group 0            | q1 @ group 1 | q2, q3 @ group 2   (all before barrier j)
        // group 0
        if constexpr (!(ablate & 16)) read_frags(C, 1, wfB, xfB);
        if constexpr (!(ablate & 8)) mfma_group(wfA, xfA);
        if constexpr (ablate & 2048) {
            build_table(ram[Nn]);
            decode_q_valu(rw[Nn][0], 0, Nn);
            decode_q_valu(rw[Nn][1], 1, Nn);
        } else if constexpr (!(ablate & 4)) {
            finish_q(La, ram[Nn], 0, Nn);
            lookup_q(rw[Nn][1], Lb);
        }
        if constexpr (!(ablate & 1) && !(ablate & 8192)) { if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 1); }
        if constexpr (ablate & 4096) interleave(I3{}, I2{}, I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        // group 1
        if constexpr (!(ablate & 16)) read_frags(C, 2, wfA, xfA);
        if constexpr (!(ablate & 8)) mfma_group(wfB, xfB);
        if constexpr (ablate & 2048) {
            decode_q_valu(rw[Nn][2], 2, Nn);
            decode_q_valu(rw[Nn][3], 3, Nn);
        } else if constexpr (!(ablate & 4)) {
            finish_q(Lb, ram[Nn], 1, Nn);
            lookup_q(rw[Nn][2], La);
            lookup_q(rw[Nn][3], Lc);
        }
        if constexpr (!(ablate & 1) && !(ablate & 8192)) { if (j > 0) issue_a(Nn, kclamp(j + 1), 3, 1); }
        if constexpr (ablate & 4096) interleave(I4{}, I3{}, I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        // group 2
        if constexpr (!(ablate & 16)) read_frags(C, 3, wfB, xfB);
        if constexpr (!(ablate & 8)) mfma_group(wfA, xfA);
        if constexpr (!(ablate & 4) && !(ablate & 2048)) {
            finish_q(La, ram[Nn], 2, Nn);
            finish_q(Lc, ram[Nn], 3, Nn);
        }
        if constexpr (!(ablate & 2)) issue_raw(Nn, kclamp(j + 3));
        const bool am_now = AM4 && (((j + 3) & 3) == 0);
        if constexpr (AM4) { if (am_now) issue_am4((j + 3) >> 2); }
        if constexpr (ablate & 4096) interleave(I4{}, I1{}, I2{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        // all but what this group just issued has landed: A(j+1), raw(j+2) (and older absmax blocks)
        if constexpr (AM4) { if (am_now) { if constexpr (NESTED) MBNB_VMCNT(3); else MBNB_VMCNT(2); } else { MBNB_VMCNT(1); } }
        else if constexpr (NESTED) MBNB_VMCNT(3); else MBNB_VMCNT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own decode writes + fragment reads done
        __builtin_amdgcn_s_barrier();                         // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        // group 3: first fragments of stage Nn; refill stage C; raw(j+2) -> registers; look up its quarter 0
        if constexpr (!(ablate & 16)) read_frags(Nn, 0, wfA, xfA);
        load_raw(PC{}, C, kclamp(j + 2));  // raw(j+2) landed before the barrier above
        if constexpr (!(ablate & 8)) mfma_group(wfB, xfB);
        if constexpr (!(ablate & 4) && !(ablate & 2048)) lookup_q(rw[C][0], La);
        if constexpr (!(ablate & 1)) issue_a(C, kclamp(j + 2), 0, (ablate & 8192) ? 4 : 2);
        if constexpr (ablate & 4096) interleave(I2{}, I2{}, I0{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
        stamp();
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);
    if constexpr (ablate & 32768) { if (dbg_sink == 0x12345u) acc[0][0][0] += 1.0f; }

    // ---- epilogue: every wave is past the last barrier-protected LDS read once all waves drained their
    // fragment reads; the extra barrier makes the stage memory reusable as store staging
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

#ifdef MBNB_ABLATION   // measured schedule alternatives: diagnostic builds only (make EXTRA=-DMBNB_ABLATION)
// =====================================================================================
// k_gemm256v — slot-pinned k-step with the decode done entirely on the VALU (gen_kstep.py valu):
// per thread and tile a 16-entry table of RNE16(code[i] * absmax) is built (16 v_mul + 8 cvt_pk) and split
// into byte planes; nibbles are looked up with v_perm_b32.  No LDS table lookups: the LDS carries only
// fragment reads, image writes and the LDS-DMA (measured LDS time of k_gemm256p's mix: 2140 cycles per
// k-step vs 2048 of MFMA -- tools/coexec_probe.hip -- so the table reads had to leave the LDS).
// =====================================================================================
template <typename T, bool NESTED, int ablate = 0, bool AM4 = false>
__global__ __launch_bounds__(512, 2) void k_gemm256v(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                     const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                     int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    // `ablate` (compile-time, debug only; -DMBNB_ABLATION builds the variants and MBNB_ABLATE selects one):
    // timing-only kernels that skip 1 = activation DMA, 2 = raw DMA, 4 = decode, 8 = MFMAs,
    // 16 = fragment reads inside the main loop.  Production kernels are ablate = 0.
    // LDS-DMA instructions per wave and k-step for the raw slot: packed + absmax (+ absmax2 when nested)
    constexpr int RAW_BYTES = 8192 + 2048 + (NESTED ? 2048 : 0);
    // the code table is a STATIC LDS object: its address is a compile-time constant, so a lookup is
    // `ds_read_b32 v, v_idx4 offset:<table>` with no address add (a table inside the dynamic region
    // costs one v_add per lookup).  256 floats keep the dynamic region 1 KiB aligned.
    __shared__ __attribute__((aligned(1024))) float s_lut[256];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

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

    fill_code_lut_rt(s_lut, tid, wp.qt);

    // ---- activation pieces: wave w moves pieces 4w..4w+3 (8 rows x 128 B each), swizzle on the source
    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0, int first = 0, int count = 4) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(a_src[i] + k0);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_A + stage * P_IMG + (wave * 4 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    // ---- weight decode role of this thread: rows 32*wave .. 32*wave+31 belong to this wave.
    // lane -> (row, k-half) chosen so that the 8 lanes of a ds_write_b128 group hit 8 different
    // swizzled chunks: lanes 0-7 even rows, 8-15 odd rows (half 0); 16-31 the same for half 1.
    const int l32 = lane & 31;
    const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
    const int b_half = l32 >> 4;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    const uint8_t *p_src = wp.packed + ((bn * wp.K_weight) >> 1) + 16 * b_half;
    const int64_t am_row = bn * wp.nblk;
    const int raw_lane = P_RAW + wave * 1024 + lane * 16;        // this lane's packed 16 B
    const int raw_am = P_RAW + 8192 + wave * 256 + lane * 4;     // absmax f32 (or the dword holding the int8 code)
    const int raw_am2 = P_RAW + 8192 + 2048 + wave * 256 + lane * 4;
    auto issue_raw = [&](int rs, int64_t k0) {
        char *base = smem + P_RAW + rs * RAW_BYTES;
        {
            auto g = (const __attribute__((address_space(1))) void *)(p_src + (k0 >> 1));
            auto l = (__attribute__((address_space(3))) void *)(base + wave * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
        const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
        if constexpr (AM4) {
        } else if constexpr (!NESTED) {
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.f32 + ai);
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
        } else {
            // the aligned dword that contains int8 code `ai`, and its absmax2 (one per 2^bs2_shift codes)
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.i8 + (ai & ~(int64_t)3));
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
            auto g2 = (const __attribute__((address_space(1))) void *)(wp.am.am2 + (ai >> wp.bs2_shift));
            auto l2 = (__attribute__((address_space(3))) void *)(base + 8192 + 2048 + wave * 256);
            __builtin_amdgcn_global_load_lds(g2, l2, 4, 0, 0);
        }
    };
    constexpr int P_AM4 = P_RAW + 2 * RAW_BYTES;
    int64_t am4_src_row = n0 + 32 * wave + (lane & 31);
    am4_src_row = am4_src_row < N ? am4_src_row : N - 1;
    auto issue_am4 = [&](int64_t blk) {
        const int64_t nb4 = wp.nblk >> 2;
        const int64_t b = blk < nb4 ? blk : nb4 - 1;
        if (lane < 32) {
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.f32 + am4_src_row * wp.nblk + 4 * b);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_AM4 + (int)(blk & 1) * 4096 + wave * 512);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };
    // raw registers of the tile being decoded, by tile parity
    u32x4 rw[2];
    float ram[2];
    auto load_raw = [&](auto pp, int rs, int64_t k0) {
        constexpr int P = decltype(pp)::value;
        const char *base = smem + rs * RAW_BYTES;
        rw[P] = *reinterpret_cast<const u32x4 *>(base + raw_lane);
        if constexpr (AM4) {  // prologue only: tiles 0 and 1, both in absmax block 0
            ram[P] = *reinterpret_cast<const float *>(smem + P_AM4 + wave * 512 + (b_row - 32 * wave) * 16 + (int)((k0 >> 6) & 3) * 4);
        } else if constexpr (!NESTED) {
            ram[P] = *reinterpret_cast<const float *>(base + raw_am);
        } else {
            const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
            const uint32_t word = *reinterpret_cast<const uint32_t *>(base + raw_am);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(ai & 3)));
            const float a2 = *reinterpret_cast<const float *>(base + raw_am2);
            ram[P] = q * (a2 / 127.0f);  // dequantize_blockwise arithmetic (functional.py:592-594)
        }
    };
    int bw_off[4];  // byte offsets of this thread's 4 output chunks inside stage 0 of the B image
#pragma unroll
    for (int d = 0; d < 4; d++) bw_off[d] = P_B + swz_off(b_row, 4 * b_half + d);
    // decode of a quarter (8 k) is split in two halves issued one MFMA group apart, so the table
    // lookups' LDS latency is covered by 8 MFMAs instead of being waited for in place:
    //   lookup_q: byte offsets 4*idx with one v_bfe_u32 per nibble (odd nibbles: 6-bit field at 8j+2 of
    //             w & 0xF0F0F0F0; even: byte j of (w << 2) & 0x3C3C3C3C), then 8 ds_read_b32
    //   finish_q: value = code * absmax in f32 -> RNE 16-bit (the reference's dequantize_4bit bits) -> ds_write_b128
    auto lookup_q = [&](uint32_t w, float (&L)[8]) {
        const uint32_t wo = w & 0xF0F0F0F0u;
        const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
        const char *lutb = reinterpret_cast<const char *>(s_lut);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            L[2 * j] = *reinterpret_cast<const float *>(lutb + __builtin_amdgcn_ubfe(we, 8 * j, wp.w8));
            L[2 * j + 1] = *reinterpret_cast<const float *>(lutb + __builtin_amdgcn_ubfe(wo, 8 * j + 2, wp.w6));
        }
    };
    auto finish_q = [&](const float (&L)[8], float am, int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = pack2<T>(L[2 * j] * am, L[2 * j + 1] * am);
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = o;
    };
    float La[8], Lb[8], Lc[8];   // looked-up code values in flight: quarters (0 then 2), 1, 3
    float L01[2][8], L23[2][8];  // (VALU-decode debug variant only)
    // interleave directive for one MFMA group: per MFMA `nv` VALU and `nr` LDS reads (+ `nw` LDS writes on
    // the last MFMAs, `nm` LDS-DMA issues in the middle).  Within one wave non-MFMA instructions issue in
    // the shadow of the wave's own MFMAs only when they sit between them; clustered runs serialise with
    // the SIMD partner's MFMAs (tools/coexec_probe.hip).
    auto interleave = [&](auto nv_, auto nr_, auto nw_, auto nm_) {
        constexpr int nv = decltype(nv_)::value, nr = decltype(nr_)::value, nw = decltype(nw_)::value, nm = decltype(nm_)::value;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
            if (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
            if (nm > 0 && r >= 2 && r < 2 + nm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (nw > 0 && r >= 8 - nw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    // ---- ablate & 2048: table-free decode on the VALU (no LDS lookups).  Per thread and tile the 16
    // scaled code values RNE16(code[i] * absmax) are built once (16 v_mul + 8 cvt_pk) and split into
    // byte planes TL/TH (low / high bytes of entries 4q..4q+3); a nibble is then looked up with
    // v_perm_b32: entries 0-7 and 8-15 by its low 3 bits, merged by bit 3, planes re-interleaved.
    uint32_t TL[4], TH[4];
    auto build_table = [&](float am) {
        uint32_t Tp[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float c0 = (wp.qt == MBNB_NF4) ? nf4_code(2 * r) : fp4_code(2 * r);
            const float c1 = (wp.qt == MBNB_NF4) ? nf4_code(2 * r + 1) : fp4_code(2 * r + 1);
            Tp[r] = pack2<T>(c0 * am, c1 * am);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            TL[q] = __builtin_amdgcn_perm(Tp[2 * q + 1], Tp[2 * q], 0x06040200u);
            TH[q] = __builtin_amdgcn_perm(Tp[2 * q + 1], Tp[2 * q], 0x07050301u);
        }
    };
    auto decode_q_valu = [&](uint32_t w, int d, int stage) {
        const uint32_t o = w >> 4;
        const uint32_t selLE = w & 0x07070707u, selLO = o & 0x07070707u;
        const uint32_t sel3E = ((w >> 1) & 0x04040404u) | 0x03020100u;
        const uint32_t sel3O = ((o >> 1) & 0x04040404u) | 0x03020100u;
        auto plane = [&](const uint32_t (&P)[4], uint32_t selL, uint32_t sel3) {
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(P[3], P[2], selL), __builtin_amdgcn_perm(P[1], P[0], selL), sel3);
        };
        const uint32_t LE = plane(TL, selLE, sel3E), HE = plane(TH, selLE, sel3E);
        const uint32_t LO = plane(TL, selLO, sel3O), HO = plane(TH, selLO, sel3O);
        const uint32_t E0 = __builtin_amdgcn_perm(HE, LE, 0x05010400u), E1 = __builtin_amdgcn_perm(HE, LE, 0x07030602u);
        const uint32_t O0 = __builtin_amdgcn_perm(HO, LO, 0x05010400u), O1 = __builtin_amdgcn_perm(HO, LO, 0x07030602u);
        u32x4 out;
        out[0] = __builtin_amdgcn_perm(O0, E0, 0x05040100u);
        out[1] = __builtin_amdgcn_perm(O0, E0, 0x07060302u);
        out[2] = __builtin_amdgcn_perm(O1, E1, 0x05040100u);
        out[3] = __builtin_amdgcn_perm(O1, E1, 0x07060302u);
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = out;
    };

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
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // ---- prologue: stage 0 <- tile 0; raw(1) in registers with quarters 0,1 looked up; A(1), raw(2) in flight
    issue_a(0, 0);
    issue_raw(0, 0);
    issue_raw(1, kclamp(1));
    if constexpr (AM4) issue_am4(0);
    MBNB_VMCNT(0);
    __syncthreads();  // code table, A(0) and this wave's raw(0), raw(1) visible
    load_raw(P0{}, 0, 0);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float L[8];
        lookup_q(rw[0][d], L);
        finish_q(L, ram[0], d, 0);
    }
    load_raw(P1{}, 1, kclamp(1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_a(1, kclamp(1));
    issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();  // decoded B(0) visible (each wave waited for its own LDS writes)
    asm volatile("" ::: "memory");
    Frag wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    // ---- slot-pinned k-step with VALU decode (gemm256_kstep_valu.inc, generated by gen_kstep.py valu)
    u32x4 rw1;   // raw registers of the tile being decoded (single set: a tile's decode ends before the next load)
    float ram1;
    uint32_t Tp[8], vo, selLE, selLO, sel3E, sel3O, pa, pb, LE, HE, LO, HO, E0, E1, O0, O1, ov[4];
    auto load_raw1 = [&](int64_t t) {
        const int rs = (int)(t & 1);
        const char *base = smem + rs * RAW_BYTES;
        rw1 = *reinterpret_cast<const u32x4 *>(base + raw_lane);
        if constexpr (AM4) {
            const int64_t tc = t < nk ? t : nk - 1;
            ram1 = *reinterpret_cast<const float *>(smem + P_AM4 + (int)((tc >> 2) & 1) * 4096 + wave * 512 +
                                                    (b_row - 32 * wave) * 16 + (int)(tc & 3) * 4);
        } else if constexpr (!NESTED) {
            ram1 = *reinterpret_cast<const float *>(base + raw_am);
        } else {
            const int64_t ai = am_row + ((kclamp(t) + 32 * b_half) >> wp.bs_shift);
            const uint32_t word = *reinterpret_cast<const uint32_t *>(base + raw_am);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(ai & 3)));
            const float a2 = *reinterpret_cast<const float *>(base + raw_am2);
            ram1 = q * (a2 / 127.0f);
        }
    };
    const bool is_nf4 = wp.qt == MBNB_NF4;
#define KS_CODE(i) (is_nf4 ? nf4_code(i) : fp4_code(i))
#define KS_FRAG(dst, base, stage, t) dst = *reinterpret_cast<const Frag *>(smem + (base) + (stage) * P_IMG + (t) * 32 * ROW_BYTES)
#define KS_LOAD_RAW1(tile) load_raw1(tile)
#define KS_WRITEV(par, q) *reinterpret_cast<u32x4 *>(smem + (par) * P_IMG + bw_off[q]) = u32x4{ov[0], ov[1], ov[2], ov[3]}
#define KS_DMA_A(stage, tile, piece) issue_a(stage, kclamp(tile), piece, 1)
#define KS_DMA_RAW(slot, tile)                                                         \
    do {                                                                               \
        issue_raw(slot, kclamp(tile));                                                 \
        if constexpr (AM4) { if ((((tile)) & 3) == 0) issue_am4(((tile)) >> 2); }      \
    } while (0)
#define KS_BARRIER()                                                                                   \
    do {                                                                                               \
        if constexpr (AM4) { if (((j + 3) & 3) == 0) { MBNB_VMCNT(2); } else { MBNB_VMCNT(1); } }      \
        else if constexpr (NESTED) MBNB_VMCNT(3); else MBNB_VMCNT(2); /* all but raw(j+3) landed */    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                  \
        asm volatile("" ::: "memory");                                                                 \
    } while (0)
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
#include "gemm256_kstep_valu.inc"
    };
    // what chunks 24-31 of a (virtual) step -1 would have done for tile 1
    load_raw1(1);
    {
        constexpr int Nn = 1;  // stage of tile 1 (no write falls into this span, kept for the macro)
        (void)Nn;
#include "gemm256_kstep_valu_pro.inc"
    }
#undef KS_CODE
#undef KS_FRAG
#undef KS_LOAD_RAW1
#undef KS_WRITEV
#undef KS_DMA_A
#undef KS_DMA_RAW
#undef KS_BARRIER
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    MBNB_VMCNT(0);

    // ---- epilogue (as k_gemm256)
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
                if (out_dtype == MBNB_F16) store4(static_cast<f16_t *>(out_v) + m * N + nn, v, nn, N);
                else if (out_dtype == MBNB_BF16) store4(static_cast<bf16_t *>(out_v) + m * N + nn, v, nn, N);
                else store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
            }
        }
}

// =====================================================================================
// k_gemm256pp — "ping-pong" schedule of k_gemm256p (same tile, LDS map, DMA staging, decode).
//
// Measured on k_gemm256p: its phases add up instead of overlapping (skeleton 31 us + DMA 46 +
// decode 20 + fragment reads 3 + MFMA 73 = 173 us of a 171 us launch): all 8 waves run the same
// phase at the same time, so the LDS, VALU and matrix pipes take turns.  Here the two waves that
// share a SIMD (w and w+4) are kept in OPPOSITE phases: while waves 0-3 issue 16 register-only
// MFMAs (a 512-cycle matrix segment), waves 4-7 run their memory segment (fragment reads for
// their next 16 MFMAs, decode of two weight quarters, LDS-DMA issue), then they swap.  A k-step is
// four slots per wave, each closed by s_barrier:
//     S0  MFMA k16 groups 0,1 of tile j          (fragments F loaded in the previous S3)
//     S1  read F <- groups 2,3 of tile j; decode quarters 2,3 of tile j+1 -> stage (j+1)&1;
//         DMA raw(j+3); wait A(j+1) landed
//     S2  MFMA groups 2,3 of tile j
//     S3  read F <- groups 0,1 of tile j+1; load raw(j+2); decode quarters 0,1 of tile j+2 ->
//         stage j&1; DMA A(j+2) -> stage j&1
// Waves 4-7 run the same program one slot later (one extra barrier up front, waves 0-3 one extra
// at the end), so at every slot one wave of each SIMD feeds the matrix pipe.  Stage hand-offs:
// tile j+1 is complete after both sets finished their S1 of step j, i.e. before either set's S3;
// stage j&1 is rewritten from S3 of step j on, after both sets' last reads of tile j (their S1).
// =====================================================================================
template <typename T, bool NESTED, int ablate = 0>
__global__ __launch_bounds__(512, 2) void k_gemm256pp(const T *__restrict__ X, typename Q4ProducerRT<T, NESTED>::Params wp,
                                                      const T *__restrict__ bias, void *__restrict__ out_v, int out_dtype,
                                                      int64_t M, int64_t N, int64_t K) {
    using Frag = typename Mfma<T>::frag;
    constexpr int RAW_BYTES = 8192 + 2048 + (NESTED ? 2048 : 0);
    __shared__ __attribute__((aligned(1024))) float s_lut[256];
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the delayed set: waves 4-7 share their SIMDs with waves 0-3 (debug: ablate & 32 -> odd waves, & 64 -> waves 2,3,6,7)
    const bool delayed = (ablate & 32) ? (wave & 1) != 0 : ((ablate & 64) ? ((wave >> 1) & 1) != 0 : wave >= 4);
    const int wn = wave >> 2, wm = wave & 3;

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

    fill_code_lut_rt(s_lut, tid, wp.qt);

    const T *a_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;
        a_src[i] = X + m * K + 8 * c;
    }
    auto issue_a = [&](int stage, int64_t k0, int first = 0, int count = 4) {
#pragma unroll
        for (int i = first; i < first + count; i++) {
            auto g = (const __attribute__((address_space(1))) void *)(a_src[i] + k0);
            auto l = (__attribute__((address_space(3))) void *)(smem + P_A + stage * P_IMG + (wave * 4 + i) * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
    };

    const int l32 = lane & 31;
    const int b_row = 32 * wave + 16 * (lane >> 5) + 2 * (l32 & 7) + ((l32 >> 3) & 1);
    const int b_half = l32 >> 4;
    int64_t bn = n0 + b_row;
    bn = bn < N ? bn : N - 1;
    const uint8_t *p_src = wp.packed + ((bn * wp.K_weight) >> 1) + 16 * b_half;
    const int64_t am_row = bn * wp.nblk;
    const int raw_lane = P_RAW + wave * 1024 + lane * 16;
    const int raw_am = P_RAW + 8192 + wave * 256 + lane * 4;
    const int raw_am2 = P_RAW + 8192 + 2048 + wave * 256 + lane * 4;
    auto issue_raw = [&](int rs, int64_t k0) {
        char *base = smem + P_RAW + rs * RAW_BYTES;
        {
            auto g = (const __attribute__((address_space(1))) void *)(p_src + (k0 >> 1));
            auto l = (__attribute__((address_space(3))) void *)(base + wave * 1024);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        }
        const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
        if constexpr (!NESTED) {
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.f32 + ai);
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
        } else {
            auto g = (const __attribute__((address_space(1))) void *)(wp.am.i8 + (ai & ~(int64_t)3));
            auto l = (__attribute__((address_space(3))) void *)(base + 8192 + wave * 256);
            __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
            auto g2 = (const __attribute__((address_space(1))) void *)(wp.am.am2 + (ai >> wp.bs2_shift));
            auto l2 = (__attribute__((address_space(3))) void *)(base + 8192 + 2048 + wave * 256);
            __builtin_amdgcn_global_load_lds(g2, l2, 4, 0, 0);
        }
    };
    // raw registers of the tile being decoded, by tile parity (a tile's decode spans two k-steps)
    u32x4 rw[2];
    float ram[2];
    auto load_raw = [&](auto pp, int rs, int64_t k0) {
        constexpr int P = decltype(pp)::value;
        const char *base = smem + rs * RAW_BYTES;
        rw[P] = *reinterpret_cast<const u32x4 *>(base + raw_lane);
        if constexpr (!NESTED) {
            ram[P] = *reinterpret_cast<const float *>(base + raw_am);
        } else {
            const int64_t ai = am_row + ((k0 + 32 * b_half) >> wp.bs_shift);
            const uint32_t word = *reinterpret_cast<const uint32_t *>(base + raw_am);
            const float q = (float)(int)(int8_t)(word >> (8 * (int)(ai & 3)));
            const float a2 = *reinterpret_cast<const float *>(base + raw_am2);
            ram[P] = q * (a2 / 127.0f);
        }
    };
    int bw_off[4];
#pragma unroll
    for (int d = 0; d < 4; d++) bw_off[d] = P_B + swz_off(b_row, 4 * b_half + d);
    // decode, split in two so that the table lookups of a quarter are issued one memory segment
    // before their products are formed (the lookup latency hides behind the matrix segment in between)
    auto lookup_q = [&](uint32_t w, float (&L)[8]) {
        const uint32_t wo = w & 0xF0F0F0F0u;
        const uint32_t we = (w << 2) & 0x3C3C3C3Cu;
        const char *lutb = reinterpret_cast<const char *>(s_lut);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            L[2 * j] = *reinterpret_cast<const float *>(lutb + bfe_u32(we, 8 * j, 8));
            L[2 * j + 1] = *reinterpret_cast<const float *>(lutb + bfe_u32(wo, 8 * j + 2, 6));
        }
    };
    auto finish_q = [&](const float (&L)[8], float am, int d, int stage) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = pack2<T>(L[2 * j] * am, L[2 * j + 1] * am);
        *reinterpret_cast<u32x4 *>(smem + stage * P_IMG + bw_off[d]) = o;
    };
    auto emit_q = [&](auto pp, int d, int stage) {  // unpipelined form (prologue only)
        constexpr int P = decltype(pp)::value;
        float L[8];
        lookup_q(rw[P][d], L);
        finish_q(L, ram[P], d, stage);
    };
    float L01[2][8], L23[2][8];  // looked-up code values of quarters (0,1) and (2,3) in flight

    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 64 * ROW_BYTES + f;
    }
    Frag wf[2][4], xf[2][2];  // fragments of two k16 groups (one half of a k-step)
    auto read_half = [&](int stage, int half) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                wf[g][i] = *reinterpret_cast<const Frag *>(smem + fw[2 * half + g] + stage * P_IMG + i * 32 * ROW_BYTES);
#pragma unroll
            for (int j = 0; j < 2; j++)
                xf[g][j] = *reinterpret_cast<const Frag *>(smem + fx[2 * half + g] + stage * P_IMG + j * 32 * ROW_BYTES);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    auto mfma_half = [&]() {
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[g][i], xf[g][j], acc[i][j]);
    };
    // 16 MFMAs with two LDS-DMA issue points inside (after the 8th and the 12th MFMA): the DMA
    // instructions queue behind the CU's L2->LDS path, so they are spread over the matrix segments
    // where the issuing wave has idle issue slots, instead of bursting after a barrier.
    auto mfma_half_dma = [&](auto &&dma0, auto &&dma1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[0][i], xf[0][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        dma0();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[1][i], xf[1][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        dma1();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 2; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = Mfma<T>::run(wf[1][i], xf[1][j], acc[i][j]);
    };
    int dbg_n = 0;
    auto stamp = [&]() {
#ifdef MBNB_ABLATION
        if constexpr (ablate & 512) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && dbg_n < 1024) g_dbg_stamps[wave >> 2][dbg_n] = t;
            dbg_n++;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    auto slot_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();  // work of this slot done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp();  // next slot starts
        __builtin_amdgcn_sched_barrier(0);
    };

    const int64_t nk = K >> 6;
    const int64_t k_last = (nk - 1) << 6;
    auto kclamp = [&](int64_t t) { return t < nk ? t << 6 : k_last; };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // ---- prologue (all waves together): tile 0 complete in stage 0; tile 1: quarters 0,1 in stage 1,
    //      quarters 2,3 looked up; A(1), raw(2) in flight; fragments of tile 0, groups 0,1 in registers
    issue_a(0, 0);
    issue_raw(0, 0);
    issue_raw(1, kclamp(1));
    MBNB_VMCNT(0);
    __syncthreads();
    load_raw(P0{}, 0, 0);
#pragma unroll
    for (int d = 0; d < 4; d++) emit_q(P0{}, d, 0);
    load_raw(P1{}, 1, kclamp(1));
    emit_q(P1{}, 0, 1);
    emit_q(P1{}, 1, 1);
    lookup_q(rw[1][2], L23[0]);
    lookup_q(rw[1][3], L23[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_a(1, kclamp(1));
    issue_raw(0, kclamp(2));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_half(0, 0);
    if (delayed) slot_end();  // waves 4-7 start one slot later

    // LDS-DMA issue plan per wave and k-step j (program order):
    //   S3(j-1): A(j+1) pieces 0,1 | S0(j): A(j+1) pieces 2,3 | S2(j): raw(j+3) | S3(j): A(j+2) pieces 0,1
    // Decode of tile t (parity t&1) is a 3-slot pipeline:
    //   S1(t-2): load raw(t), look up quarters 0,1 | S3(t-2): finish 0,1 -> stage t&1, look up 2,3 | S1(t-1): finish 2,3
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        using PC = std::integral_constant<int, C>;
        // S0: matrix segment, groups 0,1 (+ second half of A(j+1) -> stage Nn)
        if constexpr (!(ablate & 8)) {
            mfma_half_dma([&] { if constexpr (!(ablate & 1)) { if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 1); } },
                          [&] { if constexpr (!(ablate & 1)) { if (j > 0) issue_a(Nn, kclamp(j + 1), 3, 1); } });
        } else {
            if constexpr (!(ablate & 1)) { if (j > 0) issue_a(Nn, kclamp(j + 1), 2, 2); }
        }
        slot_end();
        // S1: memory segment.  raw(j+2) (tile parity C) was issued at S2(j-1): everything but the four
        // A(j+1) pieces issued after it has landed after vmcnt(4) (at j = 0 the prologue order differs).
        if (j == 0) { MBNB_VMCNT(0); } else { MBNB_VMCNT(4); }
        load_raw(PC{}, C, kclamp(j + 2));
        if constexpr (!(ablate & 4)) { lookup_q(rw[C][0], L01[0]); lookup_q(rw[C][1], L01[1]); }
        if constexpr (!(ablate & 4)) { finish_q(L23[0], ram[Nn], 2, Nn); finish_q(L23[1], ram[Nn], 3, Nn); }
        if constexpr (!(ablate & 16)) read_half(C, 1);
        MBNB_VMCNT(0);  // A(j+1): this wave's pieces have landed
        slot_end();
        // S2: matrix segment, groups 2,3 (+ raw(j+3) -> raw slot Nn)
        if constexpr (!(ablate & 8)) {
            mfma_half_dma([&] { if constexpr (!(ablate & 2)) issue_raw(Nn, kclamp(j + 3)); }, [] {});
        } else {
            if constexpr (!(ablate & 2)) issue_raw(Nn, kclamp(j + 3));
        }
        slot_end();
        // S3: memory segment (+ first half of A(j+2) -> stage C, free since both sets passed their S1)
        if constexpr (!(ablate & 4)) { lookup_q(rw[C][2], L23[0]); lookup_q(rw[C][3], L23[1]); }
        if constexpr (!(ablate & 4)) { finish_q(L01[0], ram[C], 0, C); finish_q(L01[1], ram[C], 1, C); }
        if constexpr (!(ablate & 16)) read_half(Nn, 0);
        if constexpr (!(ablate & 1)) issue_a(C, kclamp(j + 2), 0, 2);
        slot_end();
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    if (!delayed) slot_end();
    MBNB_VMCNT(0);

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
                if (out_dtype == MBNB_F16) store4(static_cast<f16_t *>(out_v) + m * N + nn, v, nn, N);
                else if (out_dtype == MBNB_BF16) store4(static_cast<bf16_t *>(out_v) + m * N + nn, v, nn, N);
                else store4(static_cast<float *>(out_v) + m * N + nn, v, nn, N);
            }
        }
}

#endif  // MBNB_ABLATION

template <bool NESTED> constexpr int gemm256p_lds_bytes() {
    // stages + two raw slots + the absmax-by-4 slots of the AM4 variants (plain: 2 x 4 KiB, double-quantised: 2 x 2 KiB)
    return P_RAW + 2 * (8192 + 2048 + (NESTED ? 2048 : 0)) + (NESTED ? 4096 : 8192);
}

}  // namespace mbnb
