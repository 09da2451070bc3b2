// int8_kernels.hip — rowwise-INT8 matmuls for gfx950.
//
//   matmul_int8  int8 x int8 -> int32 on v_mfma_i32_32x32x32_i8, epilogue acc*(sA[m]/127)*(sB[n]/127)
//                (reference: functional.py:788-793; Metal kernel int8_matmul_dequant mm:155-196)
//   linear_int8  16-bit activations x int8 weights decoded in the B-tile producer (gemm_tile.h)
//                (reference: Linear8bit.forward nn/linear8bit.py:70-102; Metal int8_matmul_simd mm:203-305)
#include "gemm256.h"
#include "gemm256w.h"

namespace mbnb {

// ------------------------------------------------------------------ B[K,N] -> Bt[N,K] (bytes)
__global__ __launch_bounds__(256) void k_transpose_i8(const int8_t *__restrict__ B, int8_t *__restrict__ Bt, int64_t K,
                                                     int64_t N) {
    __shared__ int8_t tile[64][65];
    const int64_t k0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t k = k0 + i, n = n0 + tx;
        tile[i][tx] = (k < K && n < N) ? B[k * N + n] : (int8_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t n = n0 + i, k = k0 + tx;
        if (n < N && k < K) Bt[n * K + k] = tile[tx][i];
    }
}

// 64 x 64 tiles with 4 x 4 byte blocks transposed in registers (v_perm_b32): a thread loads 4 bytes (4 n) of 4
// consecutive k rows -- a wave's load covers 64-byte row segments --, writes the 4 transposed dwords into an LDS
// tile [n][k] and stores 16 bytes (16 k of one n): 4 lanes complete a 64-byte segment of a Bt row.
// Needs K % 64 == 0, N % 64 == 0 and 16-byte aligned pointers.
__global__ __launch_bounds__(256) void k_transpose_i8_64(const int8_t *__restrict__ B, int8_t *__restrict__ Bt, int64_t K,
                                                        int64_t N) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64][17];   // [n][k / 4], one dword of padding per row
    const int64_t k0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    const int n4 = threadIdx.x & 15, kg = threadIdx.x >> 4;
    uint32_t r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = *reinterpret_cast<const uint32_t *>(B + (k0 + 4 * kg + j) * N + n0 + 4 * n4);
    const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    tile[4 * n4 + 0][kg] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    tile[4 * n4 + 1][kg] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    tile[4 * n4 + 2][kg] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    tile[4 * n4 + 3][kg] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
    __syncthreads();
    const int n = threadIdx.x >> 2, kc = threadIdx.x & 3;
    const u32x4 v = u32x4{tile[n][4 * kc], tile[n][4 * kc + 1], tile[n][4 * kc + 2], tile[n][4 * kc + 3]};
    *reinterpret_cast<u32x4 *>(Bt + (n0 + n) * K + k0 + 16 * kc) = v;
}

// 128 x 128 tiles, 16-byte accesses on both sides: a thread loads 16 bytes (16 n) of 4 consecutive k rows (8 lanes cover a
// 128-byte row segment), transposes the four 4 x 4 byte blocks in registers, writes 16 dwords into the LDS tile [n][k / 4]
// (33-dword pitch) and stores 16 bytes (16 k of one n): 8 lanes complete a 128-byte segment of a Bt row.
// 4096^2: 8.8 us against 9.9 us for the 64 x 64 form (64-byte segments, 4-byte loads; rocprof).  Needs K % 128 == 0, N % 128 == 0.
__global__ __launch_bounds__(256) void k_transpose_i8_128(const int8_t *__restrict__ B, int8_t *__restrict__ Bt, int64_t K,
                                                         int64_t N) {
    __shared__ uint32_t tile[128][33];   // [n][k / 4]
    const int64_t k0 = (int64_t)blockIdx.y * 128, n0 = (int64_t)blockIdx.x * 128;
    const int n16 = threadIdx.x & 7, kg = threadIdx.x >> 3;
    u32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(B + (k0 + 4 * kg + j) * N + n0 + 16 * n16));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t0 = __builtin_amdgcn_perm(r[1][q], r[0][q], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1][q], r[0][q], 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(r[3][q], r[2][q], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3][q], r[2][q], 0x07030602u);
        tile[16 * n16 + 4 * q + 0][kg] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
        tile[16 * n16 + 4 * q + 1][kg] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
        tile[16 * n16 + 4 * q + 2][kg] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
        tile[16 * n16 + 4 * q + 3][kg] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
    }
    __syncthreads();
    const int kc = threadIdx.x & 7, nb = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n = nb + 32 * i;
        const u32x4 v = u32x4{tile[n][4 * kc], tile[n][4 * kc + 1], tile[n][4 * kc + 2], tile[n][4 * kc + 3]};
        *reinterpret_cast<u32x4 *>(Bt + (n0 + n) * K + k0 + 16 * kc) = v;
    }
}

// ------------------------------------------------------------------ int8 x int8 MFMA GEMM
// Tile 128 x 128 x 128(k, int8) -> the same 128-byte-row swizzled LDS images as gemm_tile.h.
// Orientation as there: Bt rows (n) are the MFMA "A" operand, A rows (m) the "B" operand.
template <typename OutT>
__global__ __launch_bounds__(256, 2) void k_gemm_i8(const int8_t *__restrict__ A, const int8_t *__restrict__ Bt,
                                                    const float *__restrict__ sA, const float *__restrict__ sB,
                                                    OutT *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    constexpr int BM = 128, BN = 128, BK8 = 128;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = tiles_m * tiles_n;
    int64_t bid = blockIdx.x;
    {
        const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int64_t m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;
    const int chunk = tid & 7, row = tid >> 3;

    u32x4 a_regs[4], b_regs[4];
    auto fetch_tile = [&](int64_t k0) {
        const int64_t k = k0 + chunk * 16;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int64_t m = m0 + row + 32 * p, n = n0 + row + 32 * p;
            a_regs[p] = (m < M && k + 16 <= K) ? *reinterpret_cast<const u32x4 *>(A + m * K + k) : u32x4{0, 0, 0, 0};
            b_regs[p] = (n < N && k + 16 <= K) ? *reinterpret_cast<const u32x4 *>(Bt + n * K + k) : u32x4{0, 0, 0, 0};
        }
    };
    auto stage_tile = [&](int buf) {
        char *As = smem + buf * STAGE, *Bs = As + A_BYTES;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            *reinterpret_cast<u32x4 *>(As + swz_off(row + 32 * p, chunk)) = a_regs[p];
            *reinterpret_cast<u32x4 *>(Bs + swz_off(row + 32 * p, chunk)) = b_regs[p];
        }
    };
    i32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;

    const int64_t nk = (K + BK8 - 1) / BK8;
    fetch_tile(0);
    stage_tile(0);
    const int fr = lane & 31, fh = lane >> 5;
    for (int64_t kt = 0; kt < nk; kt++) {
        const int buf = (int)(kt & 1);
        __syncthreads();
        if (kt + 1 < nk) fetch_tile((kt + 1) * BK8);
        const char *As = smem + buf * STAGE, *Bs = As + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            i32x4 wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                wf[i] = *reinterpret_cast<const i32x4 *>(Bs + swz_off(wn * 64 + i * 32 + fr, 2 * s + fh));
                xf[i] = *reinterpret_cast<const i32x4 *>(As + swz_off(wm * 64 + i * 32 + fr, 2 * s + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stage_tile(buf ^ 1);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t m = m0 + wm * 64 + j * 32 + fr;
            if (m >= M) continue;
            const float sa = sA[m] / 127.0f;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t n = n0 + wn * 64 + i * 32 + 8 * g + 4 * fh;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float sb = (n + e < N) ? sB[n + e] / 127.0f : 0.0f;
                    v[e] = (float)acc[i][j][4 * g + e] * sa * sb;
                }
                OutT *o = out + m * N + n;
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
        }
}

// ------------------------------------------------------------------ 256 x 256 x 128 pipelined int8 GEMM
// The large-problem kernel: the tile, wave grid, LDS images and fragment reads of k_gemm256p (gemm256.h)
// with BOTH operands arriving by LDS-DMA (no decode): 8 x 1 KiB pieces per wave and k-step, issued one
// per MFMA right after the barrier that frees the stage; one barrier per k-step between MFMA groups 2
// and 3, next group's fragments read before the current group's MFMAs.  Requires K % 128 == 0.
// Epilogue of the 256 x 256 int8 kernels (wave tile 128 n x 64 m, acc[i][j][4g+e] = out[m0 + 64 wm + 32 j + fr][n0 + 128 wn +
// 32 i + 8 g + 4 fh + e]): scales, optional outlier term and bias (OutlierEpilogue), rounding chain of the reference.
template <typename OutT>
__device__ __forceinline__ void i8_256_epilogue(const i32x16 (&acc)[4][2], const float *__restrict__ sA, const float *__restrict__ sB,
                                                OutT *__restrict__ out, int64_t M, int64_t N, int64_t m0, int64_t n0, int wn,
                                                int wm, int fr, int fh, const OutlierEpilogue &ep) {
    const bool with_outliers = ep.x != nullptr && ep.n_out > 0;
    const OutT *bias = static_cast<const OutT *>(ep.bias);
    // Outlier term: sum over the outlier columns of x[m, idx_j] * ow[n, j], one 16-bit MFMA per 32 x 32 tile and CHUNK of 16
    // outliers (any number of chunks: the f32 accumulator runs through them, one rounding at the end as the reference's
    // single GEMM, nn/outlier_aware.py:141-143).  ep.x: compact [M, ldx] outlier activations, zero padded to ldx =
    // 16 * chunks, written by the quantize kernel; the weights [N, n_out] are read 8 at a time (one 16-byte load when the
    // row is 16-byte aligned, a guarded gather at the ragged end).  The activation fragments of chunk 0 are fetched once.
    const int64_t nchunks = with_outliers ? (ep.n_out + 15) / 16 : 0;
    u32x4 xfr0[2];
    if constexpr (sizeof(OutT) == 2) {
        if (with_outliers) {
            const OutT *xx = static_cast<const OutT *>(ep.x);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                int64_t mrow = m0 + wm * 64 + j * 32 + fr;
                mrow = mrow < M ? mrow : M - 1;
                xfr0[j] = *reinterpret_cast<const u32x4 *>(xx + mrow * ep.ldx + 8 * fh);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f32x16 o[2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) o[j][e] = 0.0f;
        if constexpr (sizeof(OutT) == 2) {
            using Fr = typename Mfma<OutT>::frag;
            if (with_outliers) {
                const OutT *xx = static_cast<const OutT *>(ep.x);
                const OutT *ow = static_cast<const OutT *>(ep.ow);
                const bool vec = (ep.n_out % 8 == 0) && ((reinterpret_cast<uintptr_t>(ow) & 15) == 0);
                int64_t nrow = n0 + wn * 128 + i * 32 + fr;
                nrow = nrow < N ? nrow : N - 1;
                for (int64_t c = 0; c < nchunks; c++) {
                    const int64_t j0 = 16 * c + 8 * fh;      // first of this lane's 8 outliers
                    u32x4 wfr;
                    if (vec && j0 + 8 <= ep.n_out) {
                        wfr = *reinterpret_cast<const u32x4 *>(ow + nrow * ep.n_out + j0);
                    } else {
                        __attribute__((aligned(16))) OutT t[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) t[e] = (j0 + e < ep.n_out) ? ow[nrow * ep.n_out + j0 + e] : from_f32<OutT>(0.0f);
                        wfr = *reinterpret_cast<const u32x4 *>(t);
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        u32x4 xfr = xfr0[j];
                        if (c > 0) {
                            int64_t mrow = m0 + wm * 64 + j * 32 + fr;
                            mrow = mrow < M ? mrow : M - 1;
                            xfr = *reinterpret_cast<const u32x4 *>(xx + mrow * ep.ldx + 16 * c + 8 * fh);
                        }
                        o[j] = Mfma<OutT>::run(__builtin_bit_cast(Fr, wfr), __builtin_bit_cast(Fr, xfr), o[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int64_t n = n0 + wn * 128 + i * 32 + 8 * g + 4 * fh;
            if (n >= N) continue;
            float sb[4], bv[4];   // column scales and bias of the 4 outputs: once per (i, g), shared by both row tiles
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sb[e] = (n + e < N) ? sB[n + e] / 127.0f : 0.0f;
                bv[e] = (bias != nullptr && n + e < N) ? to_f32(bias[n + e]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t m = m0 + wm * 64 + j * 32 + fr;
                if (m >= M) continue;
                const float sa = sA[m] / 127.0f;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = (float)acc[i][j][4 * g + e] * sa * sb[e];
                    if constexpr (sizeof(OutT) == 2) {
                        if (with_outliers) v[e] = to_f32(from_f32<OutT>(to_f32(from_f32<OutT>(v[e])) + to_f32(from_f32<OutT>(o[j][4 * g + e]))));
                        if (bias != nullptr) v[e] = to_f32(from_f32<OutT>(to_f32(from_f32<OutT>(v[e])) + bv[e]));
                    }
                }
                store4(out + m * N + n, v, n, N);
            }
        }
    }
}

template <typename OutT, bool BNN = false>
__global__ __launch_bounds__(512, 2) void k_gemm_i8_256(const int8_t *__restrict__ A, const int8_t *__restrict__ Bt,
                                                        const float *__restrict__ sA, const float *__restrict__ sB,
                                                        OutT *__restrict__ out, int64_t M, int64_t N, int64_t K,
                                                        OutlierEpilogue ep) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
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

    // DMA pieces: wave w moves rows 32w .. 32w+31 of the activation image (4 pieces of 8 rows x 128 B), swizzle on the
    // source.  Weight image, BNN = false: B^T [N, K], the same shape.  BNN = true: B as it lies, [K, N] row-major
    // (functional.py:788-793): the image is [128 k][256 n], wave w moves k-rows 16w .. 16w+15 (4 pieces of 4 rows x 256 B,
    // whole lines), 16-byte chunk c of row k stored at position c ^ ((k & 7) << 1); the MFMA operand (16 consecutive k
    // of one n per lane) comes out of it with two ds_read_b64_tr_b8 -- no transpose kernel, no workspace.
    const int8_t *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row, n = n0 + row;
        m = m < M ? m : M - 1;
        n = n < N ? n : N - 1;
        a_src[i] = A + m * K + 16 * c;
        if constexpr (!BNN) {
            b_src[i] = Bt + n * K + 16 * c;
        } else {
            const int kr = 4 * (wave * 4 + i) + (lane >> 4);
            int c16 = (lane & 15) ^ ((kr & 7) << 1);
            if (n0 + 16 * c16 + 16 > N) c16 = 0;      // columns past N: re-read chunk 0 (never stored)
            b_src[i] = Bt + (int64_t)kr * N + n0 + 16 * c16;
        }
    }
    auto issue_piece = [&](int stage, int64_t k0, int p) {  // p 0..3: A pieces, 4..7: B pieces
        const int i = p & 3;
        const int8_t *src = p < 4 ? a_src[i] + k0 : (BNN ? b_src[i] + k0 * N : b_src[i] + k0);
        auto g = (const __attribute__((address_space(1))) void *)src;
        auto l = (__attribute__((address_space(3))) void *)(smem + (p < 4 ? P_A : P_B) + stage * P_IMG + (wave * 4 + i) * 1024);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    };
    const int fr = lane & 31, fh = lane >> 5;
    int fw[4], fx[4], fwt[4][2];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int f = fr * ROW_BYTES + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
        fw[s] = P_B + wn * 128 * ROW_BYTES + f;
        fx[s] = P_A + wm * 64 * ROW_BYTES + f;
        // transposing reads: lane l = 16 g + 2 q + p; read r of MFMA group s covers k-rows 32 s + 16 fh + 8 r + q
        const int q = (lane & 15) >> 1, p = lane & 1, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int r = 0; r < 2; r++)
            fwt[s][r] = P_B + (32 * s + 16 * fh + 8 * r + q) * 256 + (((wn * 8 + gsel) ^ (q << 1)) << 4) + 8 * p;
    }
    auto read_frags = [&](int stage, int s, i32x4 (&wf)[4], i32x4 (&xf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (!BNN) {
                wf[i] = *reinterpret_cast<const i32x4 *>(smem + fw[s] + stage * P_IMG + i * 32 * ROW_BYTES);
            } else {
                // tile i: chunk 8 wn + 2 i + gsel (disjoint bit fields) -> the i = 0 address ^ (i << 5)
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(smem + stage * P_IMG + (fwt[s][0] ^ (i << 5))));
                const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(smem + stage * P_IMG + (fwt[s][1] ^ (i << 5))));
                wf[i] = i32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) xf[j] = *reinterpret_cast<const i32x4 *>(smem + fx[s] + stage * P_IMG + j * 32 * ROW_BYTES);
    };
    i32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;

    const int64_t nk = K >> 7;
    const int64_t k_last = (nk - 1) << 7;
    auto kclamp = [&](int64_t t) { return t < nk ? t << 7 : k_last; };

#pragma unroll
    for (int p = 0; p < 8; p++) issue_piece(0, 0, p);
#pragma unroll
    for (int p = 0; p < 8; p++) issue_piece(1, kclamp(1), p);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 landed (tile 1 still in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    i32x4 wfA[4], xfA[2], wfB[4], xfB[2];
    read_frags(0, 0, wfA, xfA);

    auto group = [&](const i32x4 (&wf)[4], const i32x4 (&xf)[2], auto &&filler) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[i][j], 0, 0, 0);
                filler(i * 2 + j);
            }
    };
    auto kstep = [&](auto cc, int64_t j) {
        constexpr int C = decltype(cc)::value, Nn = C ^ 1;
        read_frags(C, 1, wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        group(wfA, xfA, [](int) {});
        __builtin_amdgcn_sched_barrier(0);
        read_frags(C, 2, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        group(wfB, xfB, [](int) {});
        __builtin_amdgcn_sched_barrier(0);
        read_frags(C, 3, wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        group(wfA, xfA, [](int) {});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile j+1 landed (issued one k-step ago)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // stage Nn complete, stage C free
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_frags(Nn, 0, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        group(wfB, xfB, [&](int r) { issue_piece(C, kclamp(j + 2), r); });  // one DMA piece behind each MFMA
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int64_t j = 0; j < nk; j += 2) {
        kstep(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nk) kstep(std::integral_constant<int, 1>{}, j + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    i8_256_epilogue<OutT>(acc, sA, sB, out, M, N, m0, n0, wn, wm, fr, fh, ep);
}

// odd K (not a multiple of 16) or unaligned pointers: one wave per output element row
template <typename OutT>
__global__ __launch_bounds__(256) void k_matmul_i8_generic(const int8_t *__restrict__ A, const int8_t *__restrict__ B,
                                                          const float *__restrict__ sA, const float *__restrict__ sB,
                                                          OutT *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N, n = i % N;
    int acc = 0;
    for (int64_t k = 0; k < K; k++) acc += (int)A[m * K + k] * (int)B[k * N + n];
    out[i] = from_f32<OutT>((float)acc * (sA[m] / 127.0f) * (sB[n] / 127.0f));
}

// B already K-contiguous ([N, K]: Linear8bit / OutlierAwareLinear weights, or the transposed workspace of matmul_int8)
template <typename OutT>
__global__ __launch_bounds__(256) void k_matmul_i8_generic_nt(const int8_t *__restrict__ A, const int8_t *__restrict__ Bt,
                                                             const float *__restrict__ sA, const float *__restrict__ sB,
                                                             OutT *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N, n = i % N;
    int acc = 0;
    for (int64_t k = 0; k < K; k++) acc += (int)A[m * K + k] * (int)Bt[n * K + k];
    out[i] = from_f32<OutT>((float)acc * (sA[m] / 127.0f) * (sB[n] / 127.0f));
}

// gemm_dense.hip: the four-wave pipeline on int8 operands (ep: OutlierAwareLinear's second term and bias in its epilogue)
bool gemm_i8_dense_shape(int64_t M, int64_t N, int64_t K);
bool gemm_i8_dense_outlier_ok(const OutlierEpilogue &ep, int out_dtype);
int launch_gemm_i8_dense(const int8_t *, const int8_t *, const float *, const float *, int64_t, int64_t, int64_t, int, void *, hipStream_t,
                         const OutlierEpilogue *ep = nullptr);

// `ep` (may be nullptr): outlier / bias epilogue.  It is applied only by the 256 x 256 kernels with a 16-bit output;
// *ep_done tells the caller whether it was (otherwise the caller runs k_outlier_add afterwards).
int matmul_int8_nt_dispatch(const int8_t *A, const int8_t *Bt, const float *sA, const float *sB, int64_t M, int64_t N,
                            int64_t K, int out_dtype, void *out, hipStream_t st, const OutlierEpilogue *ep = nullptr,
                            bool *ep_done = nullptr) {
    if (ep_done) *ep_done = false;
    const bool fast = (K % 16 == 0) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bt)) & 15) == 0;
    if (!fast) {
        const unsigned grid = (unsigned)((M * N + 255) / 256);
        switch (out_dtype) {
            case MBNB_F16: hipLaunchKernelGGL(k_matmul_i8_generic_nt<f16_t>, dim3(grid), dim3(256), 0, st, A, Bt, sA, sB, static_cast<f16_t *>(out), M, N, K); break;
            case MBNB_BF16: hipLaunchKernelGGL(k_matmul_i8_generic_nt<bf16_t>, dim3(grid), dim3(256), 0, st, A, Bt, sA, sB, static_cast<bf16_t *>(out), M, N, K); break;
            default: hipLaunchKernelGGL(k_matmul_i8_generic_nt<float>, dim3(grid), dim3(256), 0, st, A, Bt, sA, sB, static_cast<float *>(out), M, N, K); break;
        }
        set_kernel_name("i8_generic");
        return check_launch("matmul_int8(generic nt)");
    }
    // OutlierAwareLinear with at most 64 outlier columns: the four-wave pipeline of gemm_dense.h, outlier term and bias in
    // its epilogue (one 16 x 16 x 32 MFMA per output fragment and 32 columns)
    if (ep != nullptr && gemm_i8_dense_shape(M, N, K) && gemm_i8_dense_outlier_ok(*ep, out_dtype)) {
        const int rc = launch_gemm_i8_dense(A, Bt, sA, sB, M, N, K, out_dtype, out, st, ep);
        if (ep_done) *ep_done = true;
        set_kernel_name("i8_dense+outliers");
        return rc;
    }
    if ((K % 128 == 0) && ((M + 255) / 256) * ((N + 255) / 256) >= 96) {
        const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
        constexpr int lds256 = 4 * P_IMG;
        OutlierEpilogue epv{nullptr, 0, nullptr, 0, nullptr, nullptr};
        if (ep != nullptr && out_dtype != MBNB_F32) {
            epv = *ep;
            if (ep_done) *ep_done = true;
        }
#define MBNB_I8_256(OT)                                                                                              \
    do {                                                                                                             \
        auto kern = k_gemm_i8_256<OT>;                                                                               \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds256, "matmul_int8(mfma256)")) return rc;  \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles256), dim3(512), lds256, st, A, Bt, sA, sB, static_cast<OT *>(out), M, N, K, epv); \
    } while (0)
        switch (out_dtype) {
            case MBNB_F16: MBNB_I8_256(f16_t); break;
            case MBNB_BF16: MBNB_I8_256(bf16_t); break;
            default: MBNB_I8_256(float); break;
        }
#undef MBNB_I8_256
        set_kernel_name("i8_mfma256");
        return check_launch("matmul_int8(mfma256)");
    }
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    constexpr int lds = 2 * (128 + 128) * ROW_BYTES;
#define MBNB_I8(OT)                                                                                                  \
    do {                                                                                                             \
        auto kern = k_gemm_i8<OT>;                                                                                   \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "matmul_int8(mfma128)")) return rc;     \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, A, Bt, sA, sB, static_cast<OT *>(out), M, N, K); \
    } while (0)
    switch (out_dtype) {
        case MBNB_F16: MBNB_I8(f16_t); break;
        case MBNB_BF16: MBNB_I8(bf16_t); break;
        default: MBNB_I8(float); break;
    }
#undef MBNB_I8
    set_kernel_name("i8_mfma128");
    return check_launch("matmul_int8(mfma)");
}

bool gemm_i8_inplace_shape(const int8_t *, const int8_t *, int64_t, int64_t, int64_t);
int launch_gemm_i8_inplace(const int8_t *, const int8_t *, const float *, const float *, int64_t, int64_t, int64_t, int, void *, hipStream_t);
// matmul_int8 reads B as the reference passes it, [K, N] row-major.  Large aligned problems go straight to the 256 x 256
// kernel's transposing-read form (no workspace); everything else is first re-laid out K-contiguous into the caller's
// workspace (N * K bytes) or, without one, served by the generic kernel.

// large problems with a caller workspace: B transposed once into it, then the four-wave pipeline of gemm_dense.h on int8
static bool matmul_int8_dense(int64_t M, int64_t N, int64_t K) {
    return gemm_i8_dense_shape(M, N, K) && (N % 64 == 0) && (K % 64 == 0);
}

bool matmul_int8_direct(const int8_t *A, const int8_t *B, int64_t M, int64_t N, int64_t K) {
    return (K % 128 == 0) && (N % 16 == 0) && ((M + 255) / 256) * ((N + 255) / 256) >= 96 &&
           ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
}
int64_t matmul_int8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (gemm_i8_inplace_shape(nullptr, nullptr, M, N, K)) return 0;   // four waves, B read in place (gemm_i8_inplace.h)
    if (matmul_int8_dense(M, N, K)) return N * K;     // B^T for the dense pipeline (without it: the in-place kernel below)
    return matmul_int8_direct(nullptr, nullptr, M, N, K) ? 0 : N * K;
}

int matmul_int8_dispatch(const int8_t *A, const int8_t *B, const float *sA, const float *sB, int64_t M, int64_t N,
                         int64_t K, int out_dtype, void *out, void *workspace, hipStream_t st) {
    // large aligned problems: the four-wave pipeline with B read in place (round 3: 52 us at 4096^3 against 61 us for the
    // transpose pass + k_gemm_dense<I8>; no workspace)
    if (gemm_i8_inplace_shape(A, B, M, N, K)) return launch_gemm_i8_inplace(A, B, sA, sB, M, N, K, out_dtype, out, st);
    if (workspace != nullptr && matmul_int8_dense(M, N, K) &&
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0) {
        int8_t *Bt = static_cast<int8_t *>(workspace);
        if (K % 128 == 0 && N % 128 == 0)
            hipLaunchKernelGGL(k_transpose_i8_128, dim3((unsigned)(N / 128), (unsigned)(K / 128)), dim3(256), 0, st, B, Bt, K, N);
        else
            hipLaunchKernelGGL(k_transpose_i8_64, dim3((unsigned)(N / 64), (unsigned)(K / 64)), dim3(256), 0, st, B, Bt, K, N);
        if (int rc = check_launch("matmul_int8(transpose)")) return rc;
        const int rc = launch_gemm_i8_dense(A, Bt, sA, sB, M, N, K, out_dtype, out, st);
        set_kernel_name("i8_transpose+dense");
        return rc;
    }
    if (matmul_int8_direct(A, B, M, N, K)) {
        const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
        constexpr int lds256 = 4 * P_IMG;
        OutlierEpilogue epv{nullptr, 0, nullptr, 0, nullptr, nullptr};
#define MBNB_I8_NN(OT)                                                                                               \
    do {                                                                                                             \
        auto kern = k_gemm_i8_256<OT, true>;                                                                         \
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds256, "matmul_int8(mfma256)")) return rc;  \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles256), dim3(512), lds256, st, A, B, sA, sB, static_cast<OT *>(out), M, N, K, epv); \
    } while (0)
        switch (out_dtype) {
            case MBNB_F16: MBNB_I8_NN(f16_t); break;
            case MBNB_BF16: MBNB_I8_NN(bf16_t); break;
            default: MBNB_I8_NN(float); break;
        }
#undef MBNB_I8_NN
        set_kernel_name("i8_mfma256");
        return check_launch("matmul_int8(mfma256)");
    }
    const bool fast = (K % 16 == 0) && workspace != nullptr &&
                      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (!fast) {
        const unsigned grid = (unsigned)((M * N + 255) / 256);
        switch (out_dtype) {
            case MBNB_F16: hipLaunchKernelGGL(k_matmul_i8_generic<f16_t>, dim3(grid), dim3(256), 0, st, A, B, sA, sB, static_cast<f16_t *>(out), M, N, K); break;
            case MBNB_BF16: hipLaunchKernelGGL(k_matmul_i8_generic<bf16_t>, dim3(grid), dim3(256), 0, st, A, B, sA, sB, static_cast<bf16_t *>(out), M, N, K); break;
            default: hipLaunchKernelGGL(k_matmul_i8_generic<float>, dim3(grid), dim3(256), 0, st, A, B, sA, sB, static_cast<float *>(out), M, N, K); break;
        }
        set_kernel_name("i8_generic");
        return check_launch("matmul_int8(generic)");
    }
    int8_t *Bt = static_cast<int8_t *>(workspace);
    if ((K % 64 == 0) && (N % 64 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0))
        hipLaunchKernelGGL(k_transpose_i8_64, dim3((unsigned)(N / 64), (unsigned)(K / 64)), dim3(256), 0, st, B, Bt, K, N);
    else
        hipLaunchKernelGGL(k_transpose_i8, dim3((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, B, Bt, K, N);
    return matmul_int8_nt_dispatch(A, Bt, sA, sB, M, N, K, out_dtype, out, st);
}

// ------------------------------------------------------------------ linear_int8 (W8A16)
template <typename T, int WF = W8_INT8>
__global__ __launch_bounds__(256) void k_linear_i8_generic(const T *__restrict__ X, const int8_t *__restrict__ W,
                                                          const float *__restrict__ scales, const T *__restrict__ bias,
                                                          T *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t m = blockIdx.y;
    if (n >= N) return;
    const float s = w8_row_scale<WF>(scales[n]);
    float acc = 0.0f;
    for (int64_t k = lane; k < K; k += 64) {
        const float w = to_f32(from_f32<T>(w8_decode<WF>((uint32_t)(uint8_t)W[n * K + k]) * s));
        acc = fmaf(to_f32(X[m * K + k]), w, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[m * N + n] = from_f32<T>(acc + (bias ? to_f32(bias[n]) : 0.0f));
}

// ------------------------------------------------------------------ skinny W8A16 (1 <= M <= 64)
// The weight-streaming kernel of matmul4_kernels.hip (k_skinny4) for int8 weights: a workgroup = 16 waves x one group of
// 16 weight rows; wave w contracts the 128-k blocks w, w+16, ... with all M activation rows on v_mfma_f32_16x16x32.
// Lane quarter q owns k in [32q, 32q+32) of a block: 32 weight bytes = two 16-byte loads; bytes 8g .. 8g+7 are the A
// fragment of MFMA g after sign-extend -> * (scale/127) -> RNE 16 bit (dequantize_rowwise bits); the activation
// fragment of MFMA g is the 16 bytes at k = 32q + 8g.  Partial tiles are added in wave order through LDS.
template <typename T, int MT, int WF = W8_INT8>
__global__ __launch_bounds__(1024) void k_skinny8(const T *__restrict__ X, const int8_t *__restrict__ W, const float *__restrict__ scales,
                                                 const T *__restrict__ bias, T *__restrict__ out, int64_t M, int64_t N, int64_t K) {
    constexpr int WV = 16;
    extern __shared__ __attribute__((aligned(16))) char red_raw[];   // [WV][MT][256] f32
    float *red = reinterpret_cast<float *>(red_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    int64_t nrow = n0 + r16;
    nrow = nrow < N ? nrow : N - 1;
    const int8_t *wrow = W + nrow * K;
    const float sc = w8_row_scale<WF>(scales[nrow]);
    const T *xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int64_t m = 16 * mt + r16;
        m = m < M ? m : M - 1;
        xrow[mt] = X + m * K;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t nb128 = K >> 7;
    u32x4 w[2], w_n[2];
    auto request_w = [&](int64_t b) {
        const int64_t k_lane = (b << 7) + 32 * q;
        w_n[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + k_lane));
        w_n[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + k_lane + 16));
    };
    if (wave < nb128) request_w(wave);
    for (int64_t b = wave; b < nb128; b += WV) {
        const int64_t k_lane = (b << 7) + 32 * q;
        typename Mfma16<T>::frag xf[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                xf[mt][g] = *reinterpret_cast<const typename Mfma16<T>::frag *>(xrow[mt] + k_lane + 8 * g);
        __builtin_amdgcn_sched_barrier(0);
        w[0] = w_n[0];
        w[1] = w_n[1];
        if (b + WV < nb128) request_w(b + WV);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            u32x4 fr;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t wd = w[g >> 1][2 * (g & 1) + (j >> 1)];
                const float q0 = w8_decode_sel<WF>(wd, 2 * (j & 1));
                const float q1 = w8_decode_sel<WF>(wd, 2 * (j & 1) + 1);
                fr[j] = pack2<T>(q0 * sc, q1 * sc);
            }
            const auto af = __builtin_bit_cast(typename Mfma16<T>::frag, fr);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = Mfma16<T>::run(af, xf[mt][g], acc[mt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) *reinterpret_cast<f32x4 *>(red + ((wave * MT + mt) * 256 + lane * 4)) = acc[mt];
    __syncthreads();
    for (int t = threadIdx.x; t < MT * 256; t += 1024) {
        const int mt = t >> 8, e = t & 255;
        float s = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WV; wv++) s += red[(wv * MT + mt) * 256 + e];
        const int ln = e >> 2, r = e & 3;
        const int64_t n = n0 + 4 * (ln >> 4) + r;
        const int64_t m = 16 * mt + (ln & 15);
        if (n < N && m < M) {
            const float v = s + (bias ? to_f32(bias[n]) : 0.0f);
            out[m * N + n] = from_f32<T>(v);
        }
    }
}

int64_t matmul4_splitk_slices(int64_t M, int64_t N, int64_t K);

template <typename T, int WF>
int launch_gemm_small8(const T *, const uint8_t *, const float *, const T *, T *, int64_t, int64_t, int64_t, float *, int64_t, hipStream_t);

template <typename T, int WF = W8_INT8>
static int launch_linear_int8(const void *X, int64_t M, int64_t K, const int8_t *W, const float *scales, int64_t N,
                              const void *bias, void *out, float *ws, int64_t ws_bytes, hipStream_t st) {
    const T *x = static_cast<const T *>(X);
    const T *b = static_cast<const T *>(bias);
    T *o = static_cast<T *>(out);
    if constexpr (sizeof(T) == 2) {
        if (M > 32 && M <= 256) {
            // 32 < M <= 256: weight operand decoded registers -> registers (gemm_small8.h); 1 = does not apply
            const int rc = launch_gemm_small8<T, WF>(x, reinterpret_cast<const uint8_t *>(W), scales, b, o, M, N, K, ws, ws_bytes, st);
            if (rc != MBNB_NOT_APPLICABLE) return rc;
        }
        if (M >= 1 && M <= 64 && (K % 128 == 0) && (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) == 0)) {
            const unsigned grid = (unsigned)((N + 15) / 16);
#define MBNB_SKINNY8(MT)                                                                                            \
    do {                                                                                                            \
        constexpr int lds = 16 * MT * 1024;                                                                         \
        hipLaunchKernelGGL((k_skinny8<T, MT, WF>), dim3(grid), dim3(1024), lds, st, x, W, scales, b, o, M, N, K);       \
    } while (0)
            if (M <= 16) MBNB_SKINNY8(1);
            else if (M <= 32) MBNB_SKINNY8(2);
            else MBNB_SKINNY8(4);
#undef MBNB_SKINNY8
            set_kernel_name(WF == W8_INT8 ? "w8a16_skinny" : "fp8a16_skinny");
            return check_launch("linear_int8(skinny)");
        }
        const bool fast = (K % 16 == 0) && (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) == 0) && M > 4;
        if (fast && (K % 64 == 0) && ((M + 255) / 256) * ((N + 255) / 256) >= 96) {
            // large problems: the 256 x 256 one-workgroup-per-CU kernel with the int8 -> 16-bit decode in the
            // weight-tile producer (gemm256.h, k_gemm256)
            using P = I8ProducerRT<T, WF>;
            typename P::Params wp{W, scales, N, K};
            const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
            const int od = std::is_same<T, f16_t>::value ? MBNB_F16 : MBNB_BF16;
#ifdef MBNB_ABLATION
            static const bool old_w8 = getenv("MBNB_W8_REGSTAGED") != nullptr;   // diagnostic builds only: register-staged k_gemm256
#else
            constexpr bool old_w8 = false;
#endif
            if (!old_w8 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0)) {
                // LDS-DMA pipeline (gemm256w.h): activations and raw int8 weights by global_load_lds
                auto kw = k_gemm256w<T, WF>;
                constexpr int ldsw = gemm256w_lds_bytes();
                if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kw), ldsw, "linear_int8(mfma256w)")) return rc;
                hipLaunchKernelGGL(kw, dim3((unsigned)tiles), dim3(512), ldsw, st, x, wp, b, static_cast<void *>(o), od, M, N, K);
                set_kernel_name(WF == W8_INT8 ? "w8a16_mfma256" : "fp8a16_mfma256");
                return check_launch("linear_int8(mfma256w)");
            }
            auto kern = k_gemm256<T, P>;
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), G256_LDS, "linear_int8(mfma256)")) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), G256_LDS, st, x, wp, b, static_cast<void *>(o), od, M, N, K);
            set_kernel_name(WF == W8_INT8 ? "w8a16_mfma256" : "fp8a16_mfma256");
            return check_launch("linear_int8(mfma256)");
        }
        if (fast) {
            using P = I8Producer<T, WF>;
            typename P::Params wp{W, scales, N, K};
            constexpr int BM = 128, BN = 128;
            constexpr int lds = gemm_decode_lds_bytes<BM, BN>();
            auto kern = k_gemm_decode<T, T, P, BM, BN>;
            if (int rc = ensure_dyn_lds(reinterpret_cast<const void *>(kern), lds, "linear_int8(mfma128)")) return rc;
            const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
            const int64_t slices = matmul4_splitk_slices(M, N, K);   // same policy as the 4-bit path
            if (slices > 1 && ws != nullptr && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0) &&
                ws_bytes >= slices * tiles * 65536) {
                const int64_t kps = (((K / 64) + slices - 1) / slices) * 64;
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)slices), dim3(256), lds, st, x, wp, b, o, M, N, K, ws, kps);
                int rc = check_launch("linear_int8(mfma128 split-K)");
                if (rc) return rc;
                hipLaunchKernelGGL((k_splitk_reduce<T, T>), dim3((unsigned)(tiles * 16)), dim3(256), 0, st, ws, (int)slices, b, o,
                                   M, N, (M + BM - 1) / BM, tiles);
                set_kernel_name(WF == W8_INT8 ? "w8a16_mfma128_splitk" : "fp8a16_mfma128_splitk");
                return check_launch("linear_int8(split-K reduce)");
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, x, wp, b, o, M, N, K, static_cast<float *>(nullptr), (int64_t)0);
            set_kernel_name(WF == W8_INT8 ? "w8a16_mfma128" : "fp8a16_mfma128");
            return check_launch("linear_int8(mfma)");
        }
    }
    hipLaunchKernelGGL((k_linear_i8_generic<T, WF>), dim3((unsigned)((N + 3) / 4), (unsigned)M), dim3(256), 0, st, x, W, scales,
                       b, o, M, N, K);
    set_kernel_name(WF == W8_INT8 ? "w8a16_generic" : "fp8a16_generic");
    return check_launch("linear_int8(generic)");
}

int linear8_dense_path(const void *, int, int64_t, int64_t, const void *, const float *, int64_t, bool, const void *, void *, void *, int64_t,
                       hipStream_t);

int linear_int8_dispatch(const void *X, int dtype, int64_t M, int64_t K, const int8_t *W, const float *scales,
                         int64_t N, const void *bias, void *out, void *workspace, int64_t ws_bytes, bool fused_only, hipStream_t st) {
    if (!fused_only) {   // large M: dequantise once into the workspace + dense GEMM (gemm_dense.hip)
        const int rc = linear8_dense_path(X, dtype, M, K, W, scales, N, false, bias, out, workspace, ws_bytes, st);
        if (rc != MBNB_NOT_APPLICABLE) return rc;
    }
    float *ws = static_cast<float *>(workspace);
    switch (dtype) {
        case MBNB_F16: return launch_linear_int8<f16_t>(X, M, K, W, scales, N, bias, out, ws, ws_bytes, st);
        case MBNB_BF16: return launch_linear_int8<bf16_t>(X, M, K, W, scales, N, bias, out, ws, ws_bytes, st);
        default: return launch_linear_int8<float>(X, M, K, W, scales, N, bias, out, ws, ws_bytes, st);
    }
}

// LinearFP8.forward / matmul_fp8_e4m3 (functional.py:796-807): the same W8A16 kernels with the FP8 byte decoder
int linear_fp8_dispatch(const void *X, int dtype, int64_t M, int64_t K, const uint8_t *W, const float *scales, int64_t N,
                        const void *bias, void *out, void *workspace, int64_t ws_bytes, bool fused_only, hipStream_t st) {
    if (!fused_only) {
        const int rc = linear8_dense_path(X, dtype, M, K, W, scales, N, true, bias, out, workspace, ws_bytes, st);
        if (rc != MBNB_NOT_APPLICABLE) return rc;
    }
    float *ws = static_cast<float *>(workspace);
    const int8_t *w = reinterpret_cast<const int8_t *>(W);
    switch (dtype) {
        case MBNB_F16: return launch_linear_int8<f16_t, W8_FP8>(X, M, K, w, scales, N, bias, out, ws, ws_bytes, st);
        case MBNB_BF16: return launch_linear_int8<bf16_t, W8_FP8>(X, M, K, w, scales, N, bias, out, ws, ws_bytes, st);
        default: return launch_linear_int8<float, W8_FP8>(X, M, K, w, scales, N, bias, out, ws, ws_bytes, st);
    }
}

}  // namespace mbnb
