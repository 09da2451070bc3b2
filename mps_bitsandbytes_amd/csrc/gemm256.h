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

#include "gemm_tile.h"

namespace mbnb {

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
template <typename T> struct I8ProducerRT {
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
        r.s = p.scales[n] / 127.0f;
        r.w[0] = *reinterpret_cast<const u32x4 *>(p.w + n * p.K_weight + k);
        r.w[1] = *reinterpret_cast<const u32x4 *>(p.w + n * p.K_weight + k + 16);
    }
    static __device__ __forceinline__ void emit_quarter(const Regs &r, int d, const float *, char *tile, int row, int chunk) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = r.w[d >> 1][2 * (d & 1) + (j >> 1)];
            const int q0 = (int)(int8_t)(w >> (16 * (j & 1)));
            const int q1 = (int)(int8_t)(w >> (16 * (j & 1) + 8));
            o[j] = pack2<T>((float)q0 * r.s, (float)q1 * r.s);
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

}  // namespace mbnb
