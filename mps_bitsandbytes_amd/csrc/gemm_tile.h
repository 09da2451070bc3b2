// gemm_tile.h — LDS-tiled MFMA GEMM skeleton for "activation x decoded-weight^T" products.
//
//   out[M,N] = X[M,K] · Wd[N,K]^T (+ bias),   Wd = decode(W) rounded to the 16-bit compute type
//
// The weight operand is never materialised in HBM: each workgroup decodes its own
// [BN x BK] slice (4-bit nibbles or int8) straight into the LDS image the MFMA fragments are
// read from ("B-tile producer").  Both LDS images are [rows][64] 16-bit elements = 128-byte
// rows of eight 16-byte chunks, XOR-swizzled  chunk' = chunk ^ ((row >> 1) & 7)  so that every
// ds_read_b128 fragment read (16-lane groups with 16 rows distinct mod 16) and every
// ds_write_b128 (8-lane groups) is bank-conflict free (MI355X_MICROARCH.md §LDS).
//
// MFMA orientation: the WEIGHT tile is the MFMA "A" operand (rows = n) and the ACTIVATION tile
// the "B" operand (cols = m), i.e. each 32x32 accumulator holds out^T: lane -> m, registers
// 4g..4g+3 -> four consecutive n.  A lane therefore owns 4 contiguous output elements per
// register group and stores them with one 8-byte store.
#pragma once

#include "common.h"

namespace mbnb {

constexpr int BK = 64;          // k per tile (16-bit elements) -> 128-byte LDS rows
constexpr int ROW_BYTES = 128;  // BK * 2

__device__ __forceinline__ int swz_off(int row, int chunk) {
    return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T> struct Mfma;
template <> struct Mfma<f16_t> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<bf16_t> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// LDS-DMA issued from inline assembly.  Through the builtin the instruction carries a global AND an LDS memory operand,
// which the compiler's wait-count pass treats as a "pending FLAT" access: from then on every LDS wait it inserts is
// lgkmcnt(0) instead of the exact in-order count, i.e. each fragment / lookup use drains ALL of the wave's LDS reads.
// As assembly the instruction is invisible to that pass (vmcnt for it is counted by hand here anyway).
template <int BYTES> __device__ __forceinline__ void lds_dma(const void *gptr, uint32_t lds_base) {
    static_assert(BYTES == 16 || BYTES == 4, "");
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(gptr) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_base), "v"(gptr) : "memory", "m0");
}

// 16 x 16 x 32 MFMA (skinny kernels): A lane l = row l & 15, k chunk l >> 4; D[a][b]: a = 4 * (l >> 4) + r, b = l & 15
template <typename T> struct Mfma16;
template <> struct Mfma16<f16_t> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<bf16_t> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// ------------------------------------------------------------------ B-tile producers
// A producer owns, per thread, the raw bytes of 32 consecutive k of one weight row for the
// current k-tile (`fetch`), and later turns them into 64 bytes of 16-bit values written to
// the swizzled LDS image (`emit`).  Thread t of a 256-thread group covers row t/2 (+128 per
// pass), k-half t%2.

// 4-bit (NF4 / FP4) blockwise weights, blocksize >= 32.
template <typename T, int QT, bool NESTED> struct Q4Producer {
    struct Params {
        const uint8_t *packed;  // [N, K_weight/2]
        AbsmaxView am;          // [N, K_weight/blocksize]
        int64_t N, K_weight;
        int64_t nblk;   // K_weight / blocksize
        int bs_shift;   // log2(blocksize)
    };
    struct Regs {
        u32x4 w;
        float am;
    };
    static __device__ __forceinline__ void init_lut(float *lut, int tid) { fill_code_lut<QT>(lut, tid); }
    static __device__ __forceinline__ void fetch(const Params &p, int64_t n, int64_t k, Regs &r) {
        // n: weight row, k: first of 32 consecutive k handled by this thread
        if (n < p.N && k + 32 <= p.K_weight) {
            r.w = *reinterpret_cast<const u32x4 *>(p.packed + (n * p.K_weight + k) / 2);
            r.am = load_absmax<NESTED>(p.am, n * p.nblk + (k >> p.bs_shift));
        } else {
            r.w = u32x4{0, 0, 0, 0};
            r.am = 0.0f;
        }
    }
    // lut: 16 f32 code values in LDS.  Writes 4 x 16 B at (row, chunk0..chunk0+3).
    static __device__ __forceinline__ void emit(const Regs &r, const float *lut, char *tile, int row, int chunk0) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t w = r.w[d];
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // code[idx] * absmax in f32, then RNE to the 16-bit type: the same bits as
                // dequantize_4bit(...).to(dtype) (functional.py:375-382)
                const float lo = lut[(w >> (8 * j)) & 15] * r.am;
                const float hi = lut[(w >> (8 * j + 4)) & 15] * r.am;
                o[j] = pack2<T>(lo, hi);
            }
            *reinterpret_cast<u32x4 *>(tile + swz_off(row, chunk0 + d)) = o;
        }
    }
};

// int8 rowwise weights (Linear8bit): value = round_T( (float)q * (scale[n] / 127) ), nn/linear8bit.py:76-80
template <typename T, int WF = W8_INT8> struct I8Producer {
    struct Params {
        const int8_t *w;      // [N, K]
        const float *scales;  // [N]
        int64_t N, K_weight;  // K_weight == K
    };
    struct Regs {
        u32x4 w[2];
        float s;
    };
    static __device__ __forceinline__ void init_lut(float *, int) {}
    static __device__ __forceinline__ void fetch(const Params &p, int64_t n, int64_t k, Regs &r) {
        if (n < p.N) {
            r.s = w8_row_scale<WF>(p.scales[n]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int64_t kk = k + 16 * h;
                if (kk + 16 <= p.K_weight) {
                    r.w[h] = *reinterpret_cast<const u32x4 *>(p.w + n * p.K_weight + kk);
                } else {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int j = 0; j < 16; j++)
                        if (kk + j < p.K_weight) t[j >> 2] |= (uint32_t)(uint8_t)p.w[n * p.K_weight + kk + j] << (8 * (j & 3));
                    r.w[h] = u32x4{t[0], t[1], t[2], t[3]};
                }
            }
        } else {
            r.s = 0.0f;
            r.w[0] = r.w[1] = u32x4{0, 0, 0, 0};
        }
    }
    static __device__ __forceinline__ void emit(const Regs &r, const float *, char *tile, int row, int chunk0) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // 8 int8 per output chunk: dwords 2*(d&1), 2*(d&1)+1 of half d>>1
                const uint32_t w = r.w[d >> 1][2 * (d & 1) + (j >> 1)];
                const float q0 = w8_decode_sel<WF>(w, 2 * (j & 1));
                const float q1 = w8_decode_sel<WF>(w, 2 * (j & 1) + 1);
                o[j] = pack2<T>((float)q0 * r.s, (float)q1 * r.s);
            }
            *reinterpret_cast<u32x4 *>(tile + swz_off(row, chunk0 + d)) = o;
        }
    }
};

// ------------------------------------------------------------------ the kernel
// Tile BM x BN x 64, 256 threads = 4 waves as 2 (n) x 2 (m); each wave owns a (BN/2) x (BM/2)
// block of out^T as (BN/64) x (BM/64) accumulators of 32x32.
template <typename T, typename OutT, typename Producer, int BM, int BN>
__global__ __launch_bounds__(256, 2) void k_gemm_decode(const T *__restrict__ X, typename Producer::Params wp,
                                                        const T *__restrict__ bias, OutT *__restrict__ out, int64_t M,
                                                        int64_t N, int64_t K, float *__restrict__ partial = nullptr,
                                                        int64_t k_per_slice = 0) {
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile");
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_PASSES = BM / 32;   // 256 threads cover 32 rows x 8 chunks per pass
    constexpr int B_PASSES = BN / 128;  // 256 threads cover 128 rows x 2 halves per pass
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert(B_PASSES >= 1, "BN must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lut = reinterpret_cast<float *>(smem + 2 * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;

    // XCD-aware tile order: consecutive tile ids (sharing a weight strip) stay on one XCD
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int64_t m0 = (bid % tiles_m) * BM;
    const int64_t n0 = (bid / tiles_m) * BN;

    Producer::init_lut(lut, tid);
    __syncthreads();

    // per-thread staging coordinates
    const int a_chunk = tid & 7, a_row = tid >> 3;
    const int b_row = tid >> 1, b_half = tid & 1;

    u32x4 a_regs[A_PASSES];
    typename Producer::Regs b_regs[B_PASSES];

    auto fetch_tile = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < A_PASSES; p++) {
            const int64_t m = m0 + a_row + 32 * p;
            const int64_t k = k0 + a_chunk * 8;
            if (m < M && k + 8 <= K) a_regs[p] = *reinterpret_cast<const u32x4 *>(X + m * K + k);
            else a_regs[p] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; p++) Producer::fetch(wp, n0 + b_row + 128 * p, k0 + 32 * b_half, b_regs[p]);
    };
    auto stage_tile = [&](int buf) {
        char *As = smem + buf * STAGE;
        char *Bs = As + A_BYTES;
#pragma unroll
        for (int p = 0; p < A_PASSES; p++)
            *reinterpret_cast<u32x4 *>(As + swz_off(a_row + 32 * p, a_chunk)) = a_regs[p];
#pragma unroll
        for (int p = 0; p < B_PASSES; p++) Producer::emit(b_regs[p], lut, Bs, b_row + 128 * p, 4 * b_half);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // split-K (partial != nullptr): workgroup (x, y) contracts k in [y * k_per_slice, (y + 1) * k_per_slice) and
    // writes raw f32 partial sums to partial[y][M][N]; k_splitk_reduce adds the slices in order, then bias + rounding
    const int64_t k_begin = partial ? (int64_t)blockIdx.y * k_per_slice : 0;
    const int64_t k_end = partial ? (k_begin + k_per_slice < K ? k_begin + k_per_slice : K) : K;
    const int64_t nk = (k_end - k_begin + BK - 1) / BK;
    fetch_tile(k_begin);
    stage_tile(0);
    const int fr = lane & 31, fh = lane >> 5;
    for (int64_t kt = 0; kt < nk; kt++) {
        const int buf = (int)(kt & 1);
        __syncthreads();  // tile kt visible; buffer buf^1 free (its readers finished before this barrier)
        if (kt + 1 < nk) fetch_tile(k_begin + (kt + 1) * BK);
        const char *As = smem + buf * STAGE;
        const char *Bs = As + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            typename Mfma<T>::frag wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; i++)
                wf[i] = *reinterpret_cast<const typename Mfma<T>::frag *>(Bs + swz_off(wn * (BN / 2) + i * 32 + fr, 2 * s + fh));
#pragma unroll
            for (int j = 0; j < TM; j++)
                xf[j] = *reinterpret_cast<const typename Mfma<T>::frag *>(As + swz_off(wm * (BM / 2) + j * 32 + fr, 2 * s + fh));
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) acc[i][j] = Mfma<T>::run(wf[i], xf[j], acc[i][j]);
        }
        if (kt + 1 < nk) stage_tile(buf ^ 1);
    }

    // epilogue: acc[i][j][4g+e] = out[m = m_base + j*32 + (lane&31)][n = n_base + i*32 + 8g + 4*(lane>>5) + e]
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) {
            const int64_t m = m0 + wm * (BM / 2) + j * 32 + fr;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t n = n0 + wn * (BN / 2) + i * 32 + 8 * g + 4 * fh;
                if (m >= M || n >= N) continue;
                if (partial) {
                    // accumulator order: [slice][tile][wave][i*TM+j][g][lane][4] -- a wave's store is 1 KiB contiguous
                    float *pp = partial + (((((int64_t)blockIdx.y * nwg + bid) * 4 + wave) * (TN * TM) + (i * TM + j)) * 4 + g) * 256 + lane * 4;
                    store_f32x4_wt(pp, f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]});   // write-through: gemm256.h store4_partial
                    continue;
                }
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float s = acc[i][j][4 * g + e];
                    if (bias != nullptr && n + e < N) s += to_f32(bias[n + e]);
                    v[e] = to_f32(from_f32<T>(s));  // one rounding to the compute dtype (F.linear output)
                }
                OutT *o = out + m * N + n;
                if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(o) & (4 * sizeof(OutT) - 1)) == 0)) {
                    if constexpr (sizeof(OutT) == 2) {
                        *reinterpret_cast<u32x2 *>(o) = u32x2{pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3])};
                    } else {
                        *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (n + e < N) o[e] = from_f32<OutT>(v[e]);
                }
            }
        }
}

// out[m, n] = cast_out(round_T(sum_s partial[s][...] + bias[n])): slices added in index order (deterministic).
// partial is in the 128 x 128 kernel's accumulator order [slice][tile][wave][i*2+j][g][lane][4] (TN = TM = 2), so
// both the GEMM's stores and these loads are 1 KiB-contiguous per wave; a thread owns 4 consecutive n of one m.
template <typename T, typename OutT>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float *__restrict__ partial, int slices, const T *__restrict__ bias,
                                                      OutT *__restrict__ out, int64_t M, int64_t N, int64_t tiles_m,
                                                      int64_t nwg) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;   // = ((((tile*4 + wave)*4 + ij)*4 + g)*64 + lane)
    if (t >= nwg * 4096) return;
    const int lane = (int)(t & 63), g = (int)((t >> 6) & 3), ij = (int)((t >> 8) & 3), wave = (int)((t >> 10) & 3);
    const int64_t tile = t >> 12;
    const int i = ij >> 1, j = ij & 1, wn = wave >> 1, wm = wave & 1;
    const int64_t m = (tile % tiles_m) * 128 + wm * 64 + j * 32 + (lane & 31);
    const int64_t n = (tile / tiles_m) * 128 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
    if (m >= M || n >= N) return;
    f32x4 a = *reinterpret_cast<const f32x4 *>(partial + t * 4);
    const int64_t stride = nwg * 16384;
    for (int sl = 1; sl < slices; sl++) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(partial + (int64_t)sl * stride + t * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] += b[e];
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float s = a[e];
        if (bias != nullptr && n + e < N) s += to_f32(bias[n + e]);
        v[e] = to_f32(from_f32<T>(s));
    }
    OutT *o = out + m * N + n;
    if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(o) & (4 * sizeof(OutT) - 1)) == 0)) {
        if constexpr (sizeof(OutT) == 2) *reinterpret_cast<u32x2 *>(o) = u32x2{pack2<OutT>(v[0], v[1]), pack2<OutT>(v[2], v[3])};
        else *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (n + e < N) o[e] = from_f32<OutT>(v[e]);
    }
}

template <int BM, int BN> constexpr int gemm_decode_lds_bytes() { return 2 * (BM + BN) * ROW_BYTES + 64; }

}  // namespace mbnb
