// quant_kernels.hip — quantize / dequantize kernels (HBM-bound byte work) for gfx950.
//
// Bit-exact counterparts of the reference's pure-torch functions:
//   quantize_4bit      functional.py:163-303      dequantize_4bit      functional.py:306-416
//   quantize_blockwise functional.py:469-539      dequantize_blockwise functional.py:542-600
//   quantize_rowwise   functional.py:607-625      dequantize_rowwise   functional.py:628-636
//   double_quant       functional.py:814-863
// Compiled with -ffp-contract=off: every f32 operation below is a single IEEE operation in
// the order the reference performs it (division, not reciprocal-multiply, for x / absmax).
#include "common.h"

namespace mbnb {

// =====================================================================================
// quantize_4bit
// Lane layout (blocksize >= 8): each lane owns 8 consecutive elements of a padded row
// (16 B of fp16/bf16 in, 4 B of packed nibbles out); a quantisation block is owned by a
// team of blocksize/8 consecutive lanes (team <= 64 lanes = one wave); for blocksize > 512
// a wave walks the block in 512-element steps.  Coalesced: a wave reads 1 KiB and writes
// 256 B per step.
// =====================================================================================
template <typename T>
__device__ __forceinline__ void load8(const T *A, int64_t rows, int64_t cols, int64_t r, int64_t k0,
                                      bool vec_ok, float (&x)[8]) {
    if (vec_ok && k0 + 8 <= cols) {
        if constexpr (sizeof(T) == 2) {
            u32x4 v = *reinterpret_cast<const u32x4 *>(A + r * cols + k0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[2 * j] = unpack_lo<T>(v[j]);
                x[2 * j + 1] = unpack_hi<T>(v[j]);
            }
        } else {
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(A + r * cols + k0);
            f32x4 v1 = *reinterpret_cast<const f32x4 *>(A + r * cols + k0 + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[j] = v0[j];
                x[4 + j] = v1[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = (k0 + j < cols) ? to_f32(A[r * cols + k0 + j]) : 0.0f;
    }
}

// argmin_i fl(|xn - code[i]|) with the first minimum (functional.py:242-243), without the 16-way search.
// For a sorted table the reference's own rounded comparison between neighbours flips exactly once along the f32
// number line, at a threshold t_i (the largest f32 that still prefers code[i]); the argmin is #{i : xn > t_i}.
// The thresholds are found and checked against the brute-force argmin (2.8 M values: every f32 within 4096 ulps of
// each threshold and code, plus random ones) by tools/gen_code_thresholds.py.  FP4's table is symmetric: search
// |xn| among the non-negative codes, then index 8 + j for negative inputs -- except j = 0: -0.0 (index 8) ties with
// +0.0 (index 0) and the first minimum wins, so index 8 is never produced.  NaN compares false everywhere -> 0, as
// torch.argmin over all-NaN distances.
constexpr uint32_t NF4_THR_BITS[15] = {0xbf591cd9u, 0xbf1c5271u, 0xbeeb8480u, 0xbeadea77u, 0xbe703cedu, 0xbe0d38bcu, 0xbd3a7871u,
                                       0x3d22faffu, 0x3df64862u, 0x3e5067e0u, 0x3e9582d4u, 0x3ec753f9u, 0x3f006d03u, 0x3f248dafu,
                                       0x3f5c89d9u};
constexpr float FP4_THR[7] = {0.03125f, 0.09375f, 0.1875f, 0.3125f, 0.4375f, 0.625f, 0.875f};

template <int QT> __device__ __forceinline__ uint32_t nearest_code(float xn) {
    if constexpr (QT == MBNB_NF4) {
        uint32_t idx = 0;
#pragma unroll
        for (int i = 0; i < 15; i++) idx += (xn > __builtin_bit_cast(float, NF4_THR_BITS[i])) ? 1u : 0u;
        return idx;
    } else {
        const float a = fabsf(xn);
        uint32_t j = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) j += (a > FP4_THR[i]) ? 1u : 0u;
        return (xn < 0.0f && j > 0) ? 8u + j : j;
    }
}

// The same count with a first guess from LDS: 256 bins over xn in [-1, 1) (NF4) / |xn| in [0, 1) (FP4) hold the number
// of thresholds safely below the bin; at most one more can lie at or inside it (tools/gen_code_thresholds.py builds and
// checks the tables, code_bins.inc), so one exact compare against that threshold finishes the count: 2 LDS reads and
// ~6 VALU instead of 30 VALU per element -- the quantize kernel is VALU-bound, not HBM-bound.
#include "code_bins.inc"
template <int QT> __device__ __forceinline__ float code_threshold(int i) {   // thresholds padded with +inf
    float t = __builtin_inff();
    if constexpr (QT == MBNB_NF4) {
#pragma unroll
        for (int k = 0; k < 15; k++)
            if (i == k) t = __builtin_bit_cast(float, NF4_THR_BITS[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (i == k) t = FP4_THR[k];
    }
    return t;
}
template <int QT> __device__ __forceinline__ void fill_code_bins(uint8_t *bins, float *thr, int tid) {   // 256 threads
    bins[tid] = (QT == MBNB_NF4) ? g_nf4_bins[tid] : g_fp4_bins[tid];
    if (tid < 16) thr[tid] = code_threshold<QT>(tid);
}
template <int QT> __device__ __forceinline__ uint32_t nearest_code_lut(float xn, const uint8_t *bins, const float *thr) {
    if constexpr (QT == MBNB_NF4) {
        int b = (int)__builtin_fmaf(xn, 128.0f, 128.0f);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        const uint32_t low = bins[b];
        return low + ((xn > thr[low]) ? 1u : 0u);
    } else {
        const float a = fabsf(xn);
        int b = (int)(a * 256.0f);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        const uint32_t low = bins[b];
        const uint32_t j = low + ((a > thr[low]) ? 1u : 0u);
        return (xn < 0.0f && j > 0) ? 8u + j : j;
    }
}

// absmax of one quantisation block held by a team of `team` consecutive lanes (8 values each): abs().max() with the
// reference's clamp(min=1e-8) (functional.py:232).  torch's max PROPAGATES NaN (fmaxf drops it): a block that holds a NaN
// gets a NaN absmax, every x / absmax is then NaN and the threshold count below returns index 0 for all of them -- what
// argmin over all-NaN distances gives in the reference.
__device__ __forceinline__ float block_absmax8(const float (&x)[8], int team) {
    float am = 0.0f, nan = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        am = fmaxf(am, fabsf(x[j]));
        nan = (x[j] != x[j]) ? 1.0f : nan;
    }
    for (int off = 1; off < team; off <<= 1) {
        am = fmaxf(am, __shfl_xor(am, off, 64));
        nan = fmaxf(nan, __shfl_xor(nan, off, 64));
    }
    am = fmaxf(am, 1e-8f);
    return nan != 0.0f ? __builtin_bit_cast(float, 0x7FC00000u) : am;
}


// Outside the interval where f32 distances to the 16 codes are strictly ordered -- |xn| >= 2^22, reachable only with a
// caller-supplied absmax far below |x|, or inf -- the reference's argmin over |xn - code| (functional.py:239-240) no longer
// returns the nearest code: the distances round to the same f32 value and the FIRST index of the tie wins (2^23 -> 13, 2^25 ->
// 0, inf -> 0).  There the distances are evaluated literally.
template <int QT> __device__ __forceinline__ uint32_t nearest_code_literal(float xn) {
    float best = fabsf(xn - code_value<QT>(0));
    uint32_t idx = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const float d = fabsf(xn - code_value<QT>(i));
        if (d < best) {
            best = d;
            idx = (uint32_t)i;
        }
    }
    return idx;
}

// x / absmax for the 8 values a lane holds of one block (functional.py:236, an f32 true division -- its bits decide which side of
// a threshold a value falls).  The compiler expands every `x / am` into the IEEE sequence
//   r0 = rcp(am); r1 = fma(fma(-am, r0, 1), r0, r0); q0 = x r1; q1 = fma(fma(-am, q0, x), r1, q0); q = fma(fma(-am, q1, x), r1, q1)
// wrapped in v_div_scale / v_div_fmas / v_div_fixup, ~11 VALU per element of a VALU-bound kernel.  The reciprocal part depends
// on am alone, so it is computed once per lane; the per-element part (1 mul + 4 fma) is the same instruction chain on the
// same operands, hence the same bits, whenever the scale instructions would have been the identity: am in [2^-60, 2^60]
// (no scaling of the denominator) and |x| <= am.  What they additionally rescue -- quotients or numerators below 2^-100 --
// lies deep inside the zero code's interval, where any tiny value selects the same index.  Outside that range of am
// (wave-uniform test) the plain division runs.
struct SharedDiv {
    float am, r1;
    __device__ __forceinline__ float operator()(float x) const {
        const float q0 = x * r1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-am, q0, x), r1, q0);
        return __builtin_fmaf(__builtin_fmaf(-am, q1, x), r1, q1);
    }
};
__device__ __forceinline__ SharedDiv shared_div(float am) {
    const float r0 = __builtin_amdgcn_rcpf(am);
    return SharedDiv{am, __builtin_fmaf(__builtin_fmaf(-am, r0, 1.0f), r0, r0)};
}
__device__ __forceinline__ bool shared_div_ok(float am) {   // wave-uniform: every lane's absmax in the unscaled range
    return __builtin_amdgcn_ballot_w64(!(am >= 0x1p-60f && am <= 0x1p60f)) == 0;
}
// own_absmax: am is the block's own max |x| (so |x| <= am); a caller-supplied absmax may be smaller than |x| by any factor and
// keeps the plain division.
template <int QT> __device__ __forceinline__ uint32_t quantize8(const float (&x)[8], float am, const uint8_t *bins, const float *thr,
                                                              bool own_absmax) {
    uint32_t w = 0;
    if (own_absmax && shared_div_ok(am)) {
        const SharedDiv d = shared_div(am);
#pragma unroll
        for (int j = 0; j < 8; j++) w |= nearest_code_lut<QT>(d(x[j]), bins, thr) << (4 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float q = x[j] / am;
            w |= ((fabsf(q) < 0x1p22f) ? nearest_code_lut<QT>(q, bins, thr) : nearest_code_literal<QT>(q)) << (4 * j);
        }
    }
    return w;
}

template <typename T, int QT, int SPW>
__global__ __launch_bounds__(256) void k_quantize_4bit(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                      int64_t cols_padded, int blocksize,
                                                      const float *__restrict__ absmax_in,
                                                      uint8_t *__restrict__ packed,
                                                      float *__restrict__ absmax_out, bool vec_ok, bool row_grid) {
    __shared__ uint8_t s_bins[256];
    __shared__ float s_thr[16];
    // one wave handles SPW spans of max(blocksize, 512) consecutive padded elements of one row
    const int lane = threadIdx.x & 63;
    const int span = blocksize > 512 ? blocksize : 512;
    const int bs_shift = __builtin_ctz(blocksize);
    const int64_t spans_per_row = (cols_padded + span - 1) / span;
    const int64_t nblk = cols_padded >> bs_shift;

    if (blocksize <= 512) {
        // the loads of every span go out before the code table is filled: the table's global read and the barrier sit
        // under the data's HBM round trip instead of in front of it
        const int team = blocksize >> 3;  // lanes per block (1..64)
        int64_t r[SPW], k0[SPW];
        bool active[SPW];
        float x[SPW][8];
#pragma unroll
        for (int i = 0; i < SPW; i++) {
            if (row_grid) {   // blockIdx.x = row, blockIdx.y = group of 4 * SPW spans: no 64-bit divisions per thread
                const int64_t sp = ((int64_t)blockIdx.y * SPW + i) * 4 + (threadIdx.x >> 6);
                r[i] = blockIdx.x;
                k0[i] = sp * span + (int64_t)lane * 8;
                active[i] = sp < spans_per_row && k0[i] < cols_padded;
            } else {
                const int64_t wave = ((int64_t)blockIdx.x * SPW + i) * 4 + (threadIdx.x >> 6);
                const bool in = wave < rows * spans_per_row;
                r[i] = in ? wave / spans_per_row : 0;
                k0[i] = (in ? (wave % spans_per_row) * span : 0) + (int64_t)lane * 8;
                active[i] = in && k0[i] < cols_padded;
            }
            if (active[i]) load8<T>(A, rows, cols, r[i], k0[i], vec_ok, x[i]);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) x[i][j] = 0.0f;
            }
        }
        fill_code_bins<QT>(s_bins, s_thr, threadIdx.x);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SPW; i++) {
            const int64_t blk = active[i] ? (k0[i] >> bs_shift) : 0;
            float am;
            if (absmax_in) {
                am = active[i] ? absmax_in[r[i] * nblk + blk] : 1.0f;
            } else {
                am = block_absmax8(x[i], team);
            }
            if (!active[i]) continue;
            if ((lane & (team - 1)) == 0) absmax_out[r[i] * nblk + blk] = am;
            const uint32_t w = quantize8<QT>(x[i], am, s_bins, s_thr, absmax_in == nullptr);
            *reinterpret_cast<uint32_t *>(packed + (r[i] * cols_padded + k0[i]) / 2) = w;
        }
    } else {
        fill_code_bins<QT>(s_bins, s_thr, threadIdx.x);
        __syncthreads();
        int64_t r, kspan;
        if (row_grid) {
            const int64_t sp = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
            if (sp >= spans_per_row) return;
            r = blockIdx.x;
            kspan = sp * span;
        } else {
            const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
            if (wave >= rows * spans_per_row) return;
            r = wave / spans_per_row;
            kspan = (wave % spans_per_row) * span;
        }
        // block larger than one wave step: pass 1 absmax over the block, pass 2 quantise
        const int64_t blk = kspan / blocksize;
        float am;
        if (absmax_in) {
            am = absmax_in[r * nblk + blk];
        } else {
            am = 0.0f;
            float nan = 0.0f;
            for (int s = 0; s < span; s += 512) {
                float x[8];
                load8<T>(A, rows, cols, r, kspan + s + lane * 8, vec_ok, x);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    am = fmaxf(am, fabsf(x[j]));
                    nan = (x[j] != x[j]) ? 1.0f : nan;
                }
            }
            am = fmaxf(wave_max(am), 1e-8f);
            if (wave_max(nan) != 0.0f) am = __builtin_bit_cast(float, 0x7FC00000u);   // abs().max() propagates NaN
        }
        if (lane == 0) absmax_out[r * nblk + blk] = am;
        for (int s = 0; s < span; s += 512) {
            const int64_t k0 = kspan + s + lane * 8;
            float x[8];
            load8<T>(A, rows, cols, r, k0, vec_ok, x);
            const uint32_t w = quantize8<QT>(x, am, s_bins, s_thr, absmax_in == nullptr);
            *reinterpret_cast<uint32_t *>(packed + (r * cols_padded + k0) / 2) = w;
        }
    }
}

// quantize_4bit with compress_statistics=True in ONE launch (functional.py:288-292: quantize_blockwise(absmax, 256) right
// after the 4-bit quantisation): a workgroup owns one GROUP of 256 consecutive quantisation blocks (flat order, the order
// quantize_blockwise sees), i.e. 256 * blocksize elements walked in steps of 2048; the 256 absmax values stay in LDS and
// leave as int8 codes + one f32 absmax2 (quantize_blockwise's arithmetic: max|.| clamp 1e-8, rscale127, RNE, clamp) --
// the f32 absmax never goes to memory.  8 <= blocksize <= 512.
template <typename T, int QT>
__global__ __launch_bounds__(256) void k_quantize_4bit_dq(const T *__restrict__ A, int64_t rows, int64_t cols, int64_t cols_padded,
                                                         int blocksize, uint8_t *__restrict__ packed,
                                                         int8_t *__restrict__ am_codes, float *__restrict__ absmax2, bool vec_ok) {
    __shared__ uint8_t s_bins[256];
    __shared__ float s_thr[16];
    __shared__ float s_am[256];
    __shared__ float red[4];
    fill_code_bins<QT>(s_bins, s_thr, threadIdx.x);
    __syncthreads();
    const int tid = threadIdx.x;
    const int team = blocksize >> 3;                 // lanes per block: 1 .. 64
    const int bpi = 256 / team;                      // blocks per 2048-element step
    const int64_t nblk = cols_padded / blocksize, total = rows * nblk;
    const int64_t b0 = (int64_t)blockIdx.x * 256;
    const int tb = tid / team, tl = tid - tb * team;
    for (int it = 0; it < team; it++) {              // 256 / bpi = team steps
        const int64_t b = b0 + (int64_t)it * bpi + tb;
        const bool active = b < total;
        const int64_t bb = active ? b : 0;
        const int64_t r = bb / nblk, k0 = (bb - r * nblk) * blocksize + (int64_t)tl * 8;
        float x[8];
        if (active) load8<T>(A, rows, cols, r, k0, vec_ok, x);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = 0.0f;
        }
        const float am = block_absmax8(x, team);
        if (tl == 0) s_am[it * bpi + tb] = active ? am : 0.0f;
        if (active) {
            const uint32_t w = quantize8<QT>(x, am, s_bins, s_thr, true);
            *reinterpret_cast<uint32_t *>(packed + (r * cols_padded + k0) / 2) = w;
        }
    }
    __syncthreads();
    const float v = s_am[tid];
    float am2 = wave_max(fabsf(v));
    if ((tid & 63) == 0) red[tid >> 6] = am2;
    __syncthreads();
    am2 = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);   // functional.py:513-514
    if (tid == 0) absmax2[blockIdx.x] = am2;
    if (b0 + tid < total) am_codes[b0 + tid] = quant_i8(v, rscale127(am2));    // functional.py:518-521
}

// blocksize in {1, 2, 4}: one thread per output byte
template <typename T, int QT>
__global__ __launch_bounds__(256) void k_quantize_4bit_tiny(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                           int64_t cols_padded, int blocksize,
                                                           const float *__restrict__ absmax_in,
                                                           uint8_t *__restrict__ packed,
                                                           float *__restrict__ absmax_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // byte index
    const int64_t nbytes = rows * cols_padded / 2;
    if (j >= nbytes) return;
    const int64_t nblk = cols_padded / blocksize;
    uint32_t byte = 0;
    for (int e = 0; e < 2; e++) {
        const int64_t flat = 2 * j + e;
        const int64_t r = flat / cols_padded, k = flat % cols_padded;
        const int64_t blk = k / blocksize;
        float am;
        if (absmax_in) am = absmax_in[r * nblk + blk];
        else {
            am = 0.0f;
            for (int i = 0; i < blocksize; i++) {
                int64_t kk = blk * blocksize + i;
                am = fmaxf(am, kk < cols ? fabsf(to_f32(A[r * cols + kk])) : 0.0f);
            }
            am = fmaxf(am, 1e-8f);
        }
        if (k % blocksize == 0) absmax_out[r * nblk + blk] = am;
        float x = k < cols ? to_f32(A[r * cols + k]) : 0.0f;
        const float q = x / am;
        byte |= ((fabsf(q) < 0x1p22f) ? nearest_code<QT>(q) : nearest_code_literal<QT>(q)) << (4 * e);
    }
    packed[j] = (uint8_t)byte;
}

// =====================================================================================
// dequantize_4bit: each lane decodes one packed dword (8 elements): value = code[idx] *
// absmax (f32), cast to the output dtype (RNE).  functional.py:360-382.
// =====================================================================================
template <typename T, int QT, bool NESTED>
__global__ __launch_bounds__(256) void k_dequantize_4bit(const uint8_t *__restrict__ packed, AbsmaxView am,
                                                        int64_t rows, int64_t cols, int64_t cols_padded,
                                                        int blocksize, T *__restrict__ out, bool vec_ok, bool row_grid) {
    __shared__ float lut[16];
    const int64_t groups_per_row = (cols + 7) / 8;
    const int bs_shift = __builtin_ctz(blocksize);   // a power of two (checked at the boundary): shifts, not 64-bit divisions
    int64_t r, k0;
    bool active;
    if (row_grid) {   // blockIdx.x = row, blockIdx.y = 256 groups of 8 values: no 64-bit divisions per thread
        const int64_t gc = (int64_t)blockIdx.y * 256 + threadIdx.x;
        active = gc < groups_per_row;
        r = blockIdx.x;
        k0 = active ? gc * 8 : 0;
    } else {
        const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        active = g < rows * groups_per_row;
        r = active ? g / groups_per_row : 0;
        k0 = active ? (g % groups_per_row) * 8 : 0;
    }
    const int64_t nblk = cols_padded >> bs_shift;
    const int64_t flat0 = r * cols_padded + k0;  // even (cols_padded even, k0 multiple of 8)
    const bool dword = (cols_padded & 7) == 0;
    // the packed dword and its absmax are requested before the code table is filled: the table's barrier sits under the
    // HBM round trip instead of in front of it
    uint32_t w = 0;
    float a = 0.0f;
    if (active && dword) {
        w = *reinterpret_cast<const uint32_t *>(packed + flat0 / 2);
        if (blocksize >= 8) a = load_absmax<NESTED>(am, r * nblk + (k0 >> bs_shift));
    }
    fill_code_lut<QT>(lut, threadIdx.x);
    __syncthreads();
    if (!active) return;
    float v[8];
    if (dword) {
        if (blocksize >= 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lut[(w >> (4 * j)) & 15] * a;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                v[j] = lut[(w >> (4 * j)) & 15] * load_absmax<NESTED>(am, r * nblk + ((k0 + j) >> bs_shift));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int64_t k = k0 + j;
            if (k < cols) {
                const uint8_t b = packed[(flat0 + j) / 2];
                const int idx = ((flat0 + j) & 1) ? (b >> 4) : (b & 15);
                v[j] = lut[idx] * load_absmax<NESTED>(am, r * nblk + (k >> bs_shift));
            } else v[j] = 0.0f;
        }
    }
    T *o = out + r * cols + k0;
    if (vec_ok && k0 + 8 <= cols) {
        if constexpr (sizeof(T) == 2) {
            u32x4 p;
#pragma unroll
            for (int j = 0; j < 4; j++) p[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
            *reinterpret_cast<u32x4 *>(o) = p;
        } else {
            *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < cols) o[j] = from_f32<T>(v[j]);
    }
}

// =====================================================================================
// quantize_blockwise / dequantize_blockwise (int8, flat blocks)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_blockwise(const T *__restrict__ A, int64_t numel, int blocksize,
                                                           const float *__restrict__ absmax_in,
                                                           int8_t *__restrict__ out,
                                                           float *__restrict__ absmax_out) {
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const int64_t i0 = b * blocksize;
    const int64_t i1 = (i0 + blocksize < numel) ? i0 + blocksize : numel;
    float am;
    if (absmax_in) am = absmax_in[b];
    else {
        am = 0.0f;
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) am = fmaxf(am, fabsf(to_f32(A[i])));
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
        __syncthreads();
        am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    }
    if (threadIdx.x == 0) absmax_out[b] = am;
    const float scale = rscale127(am);  // functional.py:518
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) out[i] = quant_i8(to_f32(A[i]), scale);
}

template <typename T>
__global__ __launch_bounds__(256) void k_dequantize_blockwise(const int8_t *__restrict__ q, int64_t numel,
                                                             const float *__restrict__ absmax, int blocksize,
                                                             T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    const bool pow2 = (blocksize & (blocksize - 1)) == 0;   // a shift instead of a 64-bit division where the blocksize allows
    const float scale = absmax[pow2 ? (i >> __builtin_ctz(blocksize)) : i / blocksize] / 127.0f;  // functional.py:592
    out[i] = from_f32<T>((float)q[i] * scale);
}

// =====================================================================================
// dequant_absmax, legacy form (functional.py:878-889): absmax[r, j] = (float)code[r, j] * scales[r, j / blocksize] for
// j < dq_blocks * blocksize, 0 beyond (the reference's zeros_like).  QK: 0 int8, 1 uint8, 2 f32 codes.
// =====================================================================================
template <int QK>
__global__ __launch_bounds__(256) void k_dequant_absmax(const void *__restrict__ q, int64_t rows, int64_t num_blocks,
                                                       const float *__restrict__ scales, int64_t dq_blocks, int blocksize,
                                                       float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * num_blocks) return;
    const int64_t r = i / num_blocks, j = i - r * num_blocks, dqb = j / blocksize;
    float v = 0.0f;
    if (dqb < dq_blocks) {
        float c;
        if constexpr (QK == 0) c = (float)static_cast<const int8_t *>(q)[i];
        else if constexpr (QK == 1) c = (float)static_cast<const uint8_t *>(q)[i];
        else c = static_cast<const float *>(q)[i];
        v = c * scales[r * dq_blocks + dqb];
    }
    out[i] = v;
}

int dequant_absmax_dispatch(const void *q, int q_kind, int64_t rows, int64_t num_blocks, const float *scales,
                            int64_t dq_blocks, int blocksize, float *out, hipStream_t st) {
    const unsigned grid = (unsigned)((rows * num_blocks + 255) / 256);
    switch (q_kind) {
        case 0: hipLaunchKernelGGL(k_dequant_absmax<0>, dim3(grid), dim3(256), 0, st, q, rows, num_blocks, scales, dq_blocks, blocksize, out); break;
        case 1: hipLaunchKernelGGL(k_dequant_absmax<1>, dim3(grid), dim3(256), 0, st, q, rows, num_blocks, scales, dq_blocks, blocksize, out); break;
        default: hipLaunchKernelGGL(k_dequant_absmax<2>, dim3(grid), dim3(256), 0, st, q, rows, num_blocks, scales, dq_blocks, blocksize, out); break;
    }
    return check_launch("dequant_absmax");
}

// =====================================================================================
// quantize_rowwise: one workgroup per row (pass 1 absmax, pass 2 quantise; the row stays in L2)
// =====================================================================================
// Rows of up to 8192 16-bit values (vector path): the row is read ONCE -- a thread keeps its (up to four) 16-byte pieces in
// registers between the absmax pass and the quantise pass.  Same arithmetic as k_quantize_rowwise below.
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_rowwise_regs(const T *__restrict__ A, int64_t cols, int8_t *__restrict__ out,
                                                              float *__restrict__ scales) {
    static_assert(sizeof(T) == 2, "16-bit rows");
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    const int nvec = (int)(cols / 8);   // cols % 8 == 0, nvec <= 1024
    u32x4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int g = threadIdx.x + 256 * i;
        raw[i] = g < nvec ? *reinterpret_cast<const u32x4 *>(row + (int64_t)g * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) am = fmaxf(am, fmaxf(fabsf(unpack_lo<T>(raw[i][j])), fabsf(unpack_hi<T>(raw[i][j]))));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    if (threadIdx.x == 0) scales[r] = am;  // the absmax itself, functional.py:617-618
    const float s = rscale127(am);         // functional.py:621
    int8_t *orow = out + r * cols;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int g = threadIdx.x + 256 * i;
        if (g >= nvec) continue;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            lo |= (uint32_t)(uint8_t)quant_i8(unpack_lo<T>(raw[i][j]), s) << (16 * j);
            lo |= (uint32_t)(uint8_t)quant_i8(unpack_hi<T>(raw[i][j]), s) << (16 * j + 8);
            hi |= (uint32_t)(uint8_t)quant_i8(unpack_lo<T>(raw[i][2 + j]), s) << (16 * j);
            hi |= (uint32_t)(uint8_t)quant_i8(unpack_hi<T>(raw[i][2 + j]), s) << (16 * j + 8);
        }
        *reinterpret_cast<u32x2 *>(orow + (int64_t)g * 8) = u32x2{lo, hi};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_quantize_rowwise(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                         int8_t *__restrict__ out, float *__restrict__ scales,
                                                         bool vec_ok) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    float am = 0.0f;
    const int64_t nvec = vec_ok ? cols / 8 : 0;
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8<T>(A, rows, cols, r, g * 8, true, x);
#pragma unroll
        for (int j = 0; j < 8; j++) am = fmaxf(am, fabsf(x[j]));
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) am = fmaxf(am, fabsf(to_f32(row[c])));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
    if (threadIdx.x == 0) scales[r] = am;  // the absmax itself, functional.py:617-618
    const float s = rscale127(am);         // functional.py:621
    int8_t *orow = out + r * cols;
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8<T>(A, rows, cols, r, g * 8, true, x);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lo |= (uint32_t)(uint8_t)quant_i8(x[j], s) << (8 * j);
            hi |= (uint32_t)(uint8_t)quant_i8(x[4 + j], s) << (8 * j);
        }
        *reinterpret_cast<u32x2 *>(orow + g * 8) = u32x2{lo, hi};
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) orow[c] = quant_i8(to_f32(row[c]), s);
}

// ---------------------------------------------------------------------------------------------
// FP8 E4M3, the reference's own format (functional.py:643-673, :1086-1215; common.h float_to_fp8_e4m3 /
// fp8_e4m3_to_float): scales[r] = clamp(max|row| / 448, 1e-12); byte = encode(clamp(x / scale, +-448)).
// One workgroup per row, 8 values per thread and trip; a NaN in the row propagates into the scale as torch.max does.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_fp8(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                     uint8_t *__restrict__ out, float *__restrict__ scales, bool vec_ok) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const T *row = A + r * cols;
    float am = 0.0f, nan = 0.0f;
    const int64_t nvec = vec_ok ? cols / 8 : 0;
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8<T>(A, rows, cols, r, g * 8, true, x);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            am = fmaxf(am, fabsf(x[j]));
            nan = (x[j] != x[j]) ? 1.0f : nan;
        }
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) {
        const float v = to_f32(row[c]);
        am = fmaxf(am, fabsf(v));
        nan = (v != v) ? 1.0f : nan;
    }
    am = wave_max(am);
    nan = wave_max(nan);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = am;
        red[4 + (threadIdx.x >> 6)] = nan;
    }
    __syncthreads();
    am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    nan = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    float s = fmaxf(am / 448.0f, 1e-12f);
    if (nan > 0.0f) s = __builtin_bit_cast(float, 0x7FC00000u);
    if (threadIdx.x == 0) scales[r] = s;
    uint8_t *orow = out + r * cols;
    // v / s, a true division in the reference: the row's scale is shared, so the reciprocal part of the IEEE expansion is
    // computed once (SharedDiv above) while v_div_scale would be the identity -- s in [2^-20, 2^60] (|v| <= 448 s: no
    // overflow, no denormal quotient from a normal numerator) and |v| >= 2^-90 or v == 0 (a tinier numerator would be
    // rescaled by the expansion and decides the sign byte of a flushed result: those lanes keep the plain division)
    const bool shared = s >= 0x1p-20f && s <= 0x1p60f;
    const SharedDiv sd = shared_div(s);
    auto enc = [&](float v) {
        float n;
        if (shared && (fabsf(v) >= 0x1p-90f || v == 0.0f)) n = sd(v);
        else n = v / s;                                    // true division, as the reference
        n = (n < -448.0f) ? -448.0f : ((n > 448.0f) ? 448.0f : n);   // clamp keeps NaN
        return float_to_fp8_e4m3(n);
    };
    for (int64_t g = threadIdx.x; g < nvec; g += 256) {
        float x[8];
        load8<T>(A, rows, cols, r, g * 8, true, x);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lo |= enc(x[j]) << (8 * j);
            hi |= enc(x[4 + j]) << (8 * j);
        }
        *reinterpret_cast<u32x2 *>(orow + g * 8) = u32x2{lo, hi};
    }
    for (int64_t c = nvec * 8 + threadIdx.x; c < cols; c += 256) orow[c] = (uint8_t)enc(to_f32(row[c]));
}

template <typename T>
__global__ __launch_bounds__(256) void k_dequantize_fp8(const uint8_t *__restrict__ q, const float *__restrict__ scales,
                                                       int64_t rows, int64_t cols, T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    out[i] = from_f32<T>(fp8_e4m3_to_float(q[i]) * scales[i / cols]);
}

template <typename T>
__global__ __launch_bounds__(256) void k_dequantize_rowwise(const int8_t *__restrict__ q,
                                                           const float *__restrict__ scales, int64_t rows,
                                                           int64_t cols, T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const float s = scales[i / cols] / 127.0f;  // functional.py:635
    out[i] = from_f32<T>((float)q[i] * s);
}

// 16 consecutive elements of one row per thread (cols % 16 == 0, 16-byte aligned q / out): one 16-byte load, the same
// per-element arithmetic as the scalar kernels above (bit-identical), 16-byte stores.  FP8: the byte is an E4M3 code.
template <typename T, bool FP8>
__global__ __launch_bounds__(256) void k_dequantize_rows16(const uint8_t *__restrict__ q, const float *__restrict__ scales, int64_t rows,
                                                          int64_t cols, T *__restrict__ out) {
    // blockIdx.y = row, blockIdx.x = 256 groups of 16 columns: no 64-bit division per thread
    const int64_t c0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (c0 >= cols) return;
    const int64_t i = (int64_t)blockIdx.y * cols + c0;
    const float sc = scales[blockIdx.y];
    const float s = FP8 ? sc : sc / 127.0f;  // functional.py:635
    const u32x4 w = *reinterpret_cast<const u32x4 *>(q + i);
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        // FP8: v_cvt_f32_fp8 with its byte select (common.h w8_decode_sel: the same function as fp8_e4m3_to_float on all 256 bytes; the pass is
        // HBM-bound either way -- tools/exp/ab_w8_sweep.py: no difference inside the LinearFP8 step)
        v[e] = (FP8 ? w8_decode_sel<W8_FP8>(w[e >> 2], e & 3) : w8_decode_sel<W8_INT8>(w[e >> 2], e & 3)) * s;
    }
    if constexpr (sizeof(T) == 2) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o0[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
            o1[e] = pack2<T>(v[8 + 2 * e], v[8 + 2 * e + 1]);
        }
        *reinterpret_cast<u32x4 *>(out + i) = o0;
        *reinterpret_cast<u32x4 *>(out + i + 8) = o1;
    } else {
#pragma unroll
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4 *>(out + i + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
    }
}

// =====================================================================================
// double_quant (LLM.int8 row + column statistics)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_row_absmax(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                   float *__restrict__ row_stats) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    float am = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) am = fmaxf(am, fabsf(to_f32(A[r * cols + c])));
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) row_stats[r] = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-8f);
}

// column absmax: thread per column, 64-row slabs per block in y; non-negative floats order like uints
template <typename T>
__global__ __launch_bounds__(256) void k_col_absmax(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                   uint32_t *__restrict__ col_bits) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int64_t r1 = r0 + 64 < rows ? r0 + 64 : rows;
    float am = 0.0f;
    for (int64_t r = r0; r < r1; r++) am = fmaxf(am, fabsf(to_f32(A[r * cols + c])));
    atomicMax(col_bits + c, __float_as_uint(am));
}

__global__ __launch_bounds__(256) void k_clamp_stats(float *__restrict__ s, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) s[i] = fmaxf(s[i], 1e-8f);
}

template <typename T>
__global__ __launch_bounds__(256) void k_double_quant(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                     const float *__restrict__ col_stats,
                                                     const float *__restrict__ row_stats,
                                                     int8_t *__restrict__ out_col, int8_t *__restrict__ out_row) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const float x = to_f32(A[i]);
    out_row[i] = quant_i8(x, rscale127(row_stats[i / cols]));  // functional.py:851-854
    out_col[i] = quant_i8(x, rscale127(col_stats[i % cols]));  // functional.py:858-861
}

// ---- double_quant, vector form (cols % 8 == 0, 16-byte aligned A, 8-byte aligned outputs).  The scalar kernels above cost
// 63 us at 4096^2: two 64-bit divisions and two IEEE divisions (127 / stat) per element, 2-byte loads, A read three times
// by four launches.  Here: ONE statistics pass (k_rowcol_absmax8: atomic maxima, non-negative floats order like their bit
// patterns), then one quantise pass (a thread owns 8 columns x 8 rows: the eight column scales 127 / stat are divided once
// and reused for the rows; 16-byte loads, all eight in flight, 8-byte stores; it also applies and writes back the 1e-8 clamp).  Same values as the scalar path:
// max is exact in any order, and each output is quant_i8(x, rscale127(stat)) of the same two operands.
__global__ __launch_bounds__(256) void k_zero_stats2(float *__restrict__ a, int64_t na, float *__restrict__ b, int64_t nb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a && i < na) a[i] = 0.0f;
    if (b && i < nb) b[i] = 0.0f;
}
// A workgroup owns 128 rows x 128 columns: a wave's load instruction covers 4 rows x 256 bytes (16 lanes x 16 bytes per
// row), the workgroup 16 rows, eight such steps in flight per thread.  Column maxima: registers -> across the wave's four
// row groups by shuffle -> across the four waves through LDS -> ONE atomicMax per column and workgroup (rows / 128 per
// column in total: with one per 16 rows the 4096 hot addresses serialised the kernel, 53 us).  Row maxima: across the 16
// lanes of a row -> one atomicMax per row and workgroup.
template <typename T>
__global__ __launch_bounds__(256) void k_rowcol_absmax8(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                       uint32_t *__restrict__ row_bits, uint32_t *__restrict__ col_bits) {
    __shared__ float s_cm[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 15, rg = lane >> 4;                      // column group of 8, row within the wave's 4
    const int64_t c0 = (int64_t)blockIdx.x * 128 + cg * 8;
    const bool active = c0 < cols;
    const int64_t r0 = (int64_t)blockIdx.y * 128 + wave * 4 + rg;  // + 16 * step
    float x[8][8];
#pragma unroll
    for (int st = 0; st < 8; st++) {
        const int64_t r = r0 + 16 * st;
        if (active && r < rows) load8<T>(A, rows, cols, r, c0, true, x[st]);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) x[st][j] = 0.0f;
        }
    }
    float cm[8];
#pragma unroll
    for (int j = 0; j < 8; j++) cm[j] = 0.0f;
#pragma unroll
    for (int st = 0; st < 8; st++) {
        float rm = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float a = fabsf(x[st][j]);
            rm = fmaxf(rm, a);
            cm[j] = fmaxf(cm[j], a);
        }
        if (row_bits) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rm = fmaxf(rm, __shfl_xor(rm, off, 64));
            const int64_t r = r0 + 16 * st;
            if (cg == 0 && r < rows) atomicMax(row_bits + r, __float_as_uint(rm));
        }
    }
    if (col_bits) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            cm[j] = fmaxf(cm[j], __shfl_xor(cm[j], 16, 64));
            cm[j] = fmaxf(cm[j], __shfl_xor(cm[j], 32, 64));
        }
        if (rg == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) s_cm[wave][cg * 8 + j] = cm[j];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int64_t c = (int64_t)blockIdx.x * 128 + threadIdx.x;
            const float m = fmaxf(fmaxf(s_cm[0][threadIdx.x], s_cm[1][threadIdx.x]), fmaxf(s_cm[2][threadIdx.x], s_cm[3][threadIdx.x]));
            if (c < cols) atomicMax(col_bits + c, __float_as_uint(m));
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_double_quant8(const T *__restrict__ A, int64_t rows, int64_t cols,
                                                      float *col_stats, float *row_stats,
                                                      int8_t *__restrict__ out_col, int8_t *__restrict__ out_row,
                                                      bool clamp_cols, bool clamp_rows) {
    // clamp_rows / clamp_cols: the statistics were just computed as raw maxima; the reference's clamp(min=1e-8) is applied
    // on read here, and written back by the first workgroup of each row / column group (readers clamp too, so either value
    // they see gives the same scale) -- no separate clamp launch
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * 8;
    float x[8][8], rst[8];
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {   // the eight rows' loads go out together
        const int64_t r = r0 + rr;
        if (r < rows) {
            load8<T>(A, rows, cols, r, c0, true, x[rr]);
            rst[rr] = clamp_rows ? fmaxf(row_stats[r], 1e-8f) : row_stats[r];   // caller-supplied statistics are used as given
            if (clamp_rows && blockIdx.x == 0 && threadIdx.x == 0) row_stats[r] = rst[rr];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) x[rr][j] = 0.0f;
            rst[rr] = 1.0f;
        }
    }
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float c = clamp_cols ? fmaxf(col_stats[c0 + j], 1e-8f) : col_stats[c0 + j];
        if (clamp_cols && blockIdx.y == 0) col_stats[c0 + j] = c;
        cs[j] = rscale127(c);   // functional.py:858-861
    }
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
        const int64_t r = r0 + rr;
        if (r >= rows) break;
        const float rs = rscale127(rst[rr]);                             // functional.py:851-854
        uint32_t rl = 0, rh = 0, cl = 0, ch = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            rl |= (uint32_t)(uint8_t)quant_i8(x[rr][j], rs) << (8 * j);
            rh |= (uint32_t)(uint8_t)quant_i8(x[rr][4 + j], rs) << (8 * j);
            cl |= (uint32_t)(uint8_t)quant_i8(x[rr][j], cs[j]) << (8 * j);
            ch |= (uint32_t)(uint8_t)quant_i8(x[rr][4 + j], cs[4 + j]) << (8 * j);
        }
        *reinterpret_cast<u32x2 *>(out_row + r * cols + c0) = u32x2{rl, rh};
        *reinterpret_cast<u32x2 *>(out_col + r * cols + c0) = u32x2{cl, ch};
    }
}

// =====================================================================================
// host launchers
// =====================================================================================
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
static int launch_quantize_4bit(const void *A, int64_t rows, int64_t cols, int64_t cols_padded, int blocksize,
                                int qt, const float *absmax_in, uint8_t *packed, float *absmax_out,
                                hipStream_t st) {
    const T *a = static_cast<const T *>(A);
    if (blocksize < 8) {
        const int64_t nbytes = rows * cols_padded / 2;
        const unsigned grid = (unsigned)((nbytes + 255) / 256);
        if (qt == MBNB_NF4)
            hipLaunchKernelGGL((k_quantize_4bit_tiny<T, MBNB_NF4>), dim3(grid), dim3(256), 0, st, a, rows, cols,
                               cols_padded, blocksize, absmax_in, packed, absmax_out);
        else
            hipLaunchKernelGGL((k_quantize_4bit_tiny<T, MBNB_FP4>), dim3(grid), dim3(256), 0, st, a, rows, cols,
                               cols_padded, blocksize, absmax_in, packed, absmax_out);
        return check_launch("quantize_4bit(tiny)");
    }
    const bool vec_ok = aligned16(A) && (cols % 8 == 0);
    const int span = blocksize > 512 ? blocksize : 512;
    const int64_t spans_per_row = (cols_padded + span - 1) / span;
    const int64_t waves = rows * spans_per_row;
    // wide rows: one grid row per matrix row (no per-thread division); narrow ones keep the flat wave index.  Two spans
    // per wave (both loads in flight before the first is consumed) once a row has 8 of them and blocks fit a wave step.
    const bool row_grid = spans_per_row >= 4 && (spans_per_row + 3) / 4 <= 65535 && rows <= 0x7FFFFFFF;
    const bool two = row_grid && blocksize <= 512 && spans_per_row >= 8;
    const int64_t per_wg = two ? 8 : 4;
    const dim3 grid = row_grid ? dim3((unsigned)rows, (unsigned)((spans_per_row + per_wg - 1) / per_wg)) : dim3((unsigned)((waves + 3) / 4));
#define MBNB_Q4(QT, SPW)                                                                                           \
    hipLaunchKernelGGL((k_quantize_4bit<T, QT, SPW>), grid, dim3(256), 0, st, a, rows, cols, cols_padded, blocksize, \
                       absmax_in, packed, absmax_out, vec_ok, row_grid)
    if (qt == MBNB_NF4) {
        if (two) MBNB_Q4(MBNB_NF4, 2);
        else MBNB_Q4(MBNB_NF4, 1);
    } else {
        if (two) MBNB_Q4(MBNB_FP4, 2);
        else MBNB_Q4(MBNB_FP4, 1);
    }
#undef MBNB_Q4
    return check_launch("quantize_4bit");
}

template <typename T>
static int launch_quantize_4bit_dq(const void *A, int64_t rows, int64_t cols, int64_t cols_padded, int blocksize, int qt,
                                   uint8_t *packed, int8_t *am_codes, float *absmax2, hipStream_t st) {
    const T *a = static_cast<const T *>(A);
    const bool vec_ok = aligned16(A) && (cols % 8 == 0);
    const int64_t total = rows * (cols_padded / blocksize);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (qt == MBNB_NF4)
        hipLaunchKernelGGL((k_quantize_4bit_dq<T, MBNB_NF4>), dim3(grid), dim3(256), 0, st, a, rows, cols, cols_padded, blocksize,
                           packed, am_codes, absmax2, vec_ok);
    else
        hipLaunchKernelGGL((k_quantize_4bit_dq<T, MBNB_FP4>), dim3(grid), dim3(256), 0, st, a, rows, cols, cols_padded, blocksize,
                           packed, am_codes, absmax2, vec_ok);
    return check_launch("quantize_4bit(dq)");
}

int quantize_4bit_dq_dispatch(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded, int blocksize,
                              int qt, uint8_t *packed, int8_t *am_codes, float *absmax2, hipStream_t st) {
    switch (dtype) {
        case MBNB_F16: return launch_quantize_4bit_dq<f16_t>(A, rows, cols, cols_padded, blocksize, qt, packed, am_codes, absmax2, st);
        case MBNB_BF16: return launch_quantize_4bit_dq<bf16_t>(A, rows, cols, cols_padded, blocksize, qt, packed, am_codes, absmax2, st);
        default: return launch_quantize_4bit_dq<float>(A, rows, cols, cols_padded, blocksize, qt, packed, am_codes, absmax2, st);
    }
}

int quantize_4bit_dispatch(const void *A, int dtype, int64_t rows, int64_t cols, int64_t cols_padded,
                           int blocksize, int qt, const float *absmax_in, uint8_t *packed, float *absmax_out,
                           hipStream_t st) {
    switch (dtype) {
        case MBNB_F16: return launch_quantize_4bit<f16_t>(A, rows, cols, cols_padded, blocksize, qt, absmax_in, packed, absmax_out, st);
        case MBNB_BF16: return launch_quantize_4bit<bf16_t>(A, rows, cols, cols_padded, blocksize, qt, absmax_in, packed, absmax_out, st);
        default: return launch_quantize_4bit<float>(A, rows, cols, cols_padded, blocksize, qt, absmax_in, packed, absmax_out, st);
    }
}

// Flat form (round 3) for the matrices the decode-once path dequantises: 16-bit outputs, no row padding (cols == cols_padded), whole
// dwords and whole blocks per row -- the flat dword index IS the address of everything (packed dword g, block 8 g >> bs_shift, output
// piece g), so the row / column arithmetic of k_dequantize_4bit goes away.  Same arithmetic per value -> the same bits.  4096^2: 9.05 ->
// 8.17 us; inside the M = 1024 / 4096 steps -1.4 / -0.7 us (tools/exp/ab_dq4.py, ab_dq4_step.py, profiles/r03_dequant_flat_ab.txt; the
// same kernel with NONTEMPORAL stores runs 6.7 us -- the rate of a plain fill -- but the GEMM that reads the scratch next then pays more than
// the pass saved: 1024 x 4096^2 step 44.0 -> 49.8 us, so the stores stay cached).
// The W8A16 / FP8 scratch of the decode-once path (linear8_dense_path), shaped like k_dequantize_4bit_flat's in-step form: a thread takes 8 values
// of UN consecutive rows (one 8-byte load and ONE 16-byte store each: a wave's store instruction writes 1 KiB contiguous), write-through stores.
// The same per-element arithmetic as k_dequantize_rows16 (bit-identical).  cols % 8 == 0, q 8-byte and out 16-byte aligned, 16-bit T.
template <typename T, bool FP8, int UN>
__global__ __launch_bounds__(256) void k_dequantize_rows8_wt(const uint8_t *__restrict__ q, const float *__restrict__ scales, int64_t rows, int64_t cols,
                                                            T *__restrict__ out) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= cols) return;
    u32x2 w[UN];
    float s[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t r = (int64_t)blockIdx.y * UN + u;
        const bool ok = r < rows;
        w[u] = ok ? *reinterpret_cast<const u32x2 *>(q + r * cols + c0) : u32x2{0u, 0u};
        const float sc = ok ? scales[r] : 0.0f;
        s[u] = FP8 ? sc : sc / 127.0f;  // functional.py:635
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t r = (int64_t)blockIdx.y * UN + u;
        if (r >= rows) continue;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float v0 = (FP8 ? w8_decode_sel<W8_FP8>(w[u][e >> 1], (2 * e) & 3) : w8_decode_sel<W8_INT8>(w[u][e >> 1], (2 * e) & 3)) * s[u];
            const float v1 = (FP8 ? w8_decode_sel<W8_FP8>(w[u][e >> 1], (2 * e + 1) & 3) : w8_decode_sel<W8_INT8>(w[u][e >> 1], (2 * e + 1) & 3)) * s[u];
            o[e] = pack2<T>(v0, v1);
        }
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(out + r * cols + c0), "v"(o) : "memory");
    }
}
template <bool FP8>
static bool launch_rows8_wt(const uint8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype, void *out, hipStream_t st, int policy) {
    if (policy == 0 || (out_dtype != MBNB_F16 && out_dtype != MBNB_BF16) || cols % 8 != 0 || rows <= 0 || rows >= 65536 * 4 ||
        ((reinterpret_cast<uintptr_t>(q) & 7) | (reinterpret_cast<uintptr_t>(out) & 15)) != 0)
        return false;
    const unsigned gx = (unsigned)((cols / 8 + 255) / 256);
    if (policy == 2) {
        const dim3 grid(gx, (unsigned)((rows + 3) / 4));
        if (grid.y > 65535u) return false;
        if (out_dtype == MBNB_F16) hipLaunchKernelGGL((k_dequantize_rows8_wt<f16_t, FP8, 4>), grid, dim3(256), 0, st, q, scales, rows, cols, static_cast<f16_t *>(out));
        else hipLaunchKernelGGL((k_dequantize_rows8_wt<bf16_t, FP8, 4>), grid, dim3(256), 0, st, q, scales, rows, cols, static_cast<bf16_t *>(out));
    } else {
        const dim3 grid(gx, (unsigned)rows);
        if (grid.y > 65535u) return false;
        if (out_dtype == MBNB_F16) hipLaunchKernelGGL((k_dequantize_rows8_wt<f16_t, FP8, 1>), grid, dim3(256), 0, st, q, scales, rows, cols, static_cast<f16_t *>(out));
        else hipLaunchKernelGGL((k_dequantize_rows8_wt<bf16_t, FP8, 1>), grid, dim3(256), 0, st, q, scales, rows, cols, static_cast<bf16_t *>(out));
    }
    return true;
}

template <typename T, int QT, bool NESTED, int UN>
__global__ __launch_bounds__(256) void k_dequantize_4bit_flat(const uint8_t *__restrict__ packed, AbsmaxView am, int64_t ndw, int bs_shift,
                                                             T *__restrict__ out, int write_through) {
    __shared__ float lut[16];
    // thread t handles dwords t + 256 u (u < UN) of the workgroup's contiguous run of 256 UN: every load instruction of a wave reads 256
    // contiguous bytes, every store instruction writes 1 KiB; the packed dwords and their absmax are requested before the code table is
    // filled (its barrier sits under the HBM round trip)
    const int64_t base = (int64_t)blockIdx.x * (256 * UN) + threadIdx.x;
    uint32_t w[UN];
    float a[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t g = base + 256 * u;
        const bool ok = g < ndw;
        w[u] = ok ? reinterpret_cast<const uint32_t *>(packed)[g] : 0u;
        a[u] = ok ? load_absmax<NESTED>(am, (g * 8) >> bs_shift) : 0.0f;
    }
    fill_code_lut<QT>(lut, threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t g = base + 256 * u;
        if (g >= ndw) continue;
        u32x4 p;
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = pack2<T>(lut[(w[u] >> (8 * j)) & 15] * a[u], lut[(w[u] >> (8 * j + 4)) & 15] * a[u]);
        u32x4 *o = reinterpret_cast<u32x4 *>(out) + g;
        // write_through (the scratch of a matmul_4bit call): "sc1" stores leave no dirty lines for the end of the launch to drain before the
        // GEMM's first load (tools/exp/ab_dq4_step.py, profiles/r03_dequant_store_policy_ab.txt)
        if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(p) : "memory");
        else *o = p;
    }
}

template <typename T, int QT>
static int launch_dequantize_4bit(const uint8_t *packed, const AbsmaxView &am, int64_t rows, int64_t cols,
                                  int64_t cols_padded, int blocksize, void *out, hipStream_t st, int store_policy) {
    if constexpr (sizeof(T) == 2) {
        const int64_t ndw = rows * cols / 8;
        if (cols == cols_padded && cols % 8 == 0 && blocksize >= 8 && cols % blocksize == 0 && aligned16(out) &&
            (reinterpret_cast<uintptr_t>(packed) & 3) == 0 && ndw >= 65536 && (ndw + 255) / 256 <= 0x7FFFFFFF) {
            const int sh = __builtin_ctz((unsigned)blocksize);
            // store_policy (internal; the public dequantize_4bit passes 0): 0 = one dword per thread, cached stores (the fastest pass on its
            // own: 8.2 us at 4096^2); 1 = the same with write-through stores; 2 = FOUR dwords per thread + write-through stores -- a quarter
            // of the workgroups; slower alone, but the GEMM behind it starts earlier: the step gains 1.5-4.4 us on weights of up to 32 Mi
            // elements (600 x 4096^2 40.4 -> 36.0 us, 1024 rows 42.1 -> 39.7, 2048 rows 66.9 -> 63.2, 4096 rows 103.3 -> 101.5) and loses
            // ~1 us on larger ones (8192^2, 11008 x 4096 at 1024 rows), which keep policy 0 / 1
            if (store_policy == 2) {
                const dim3 grid((unsigned)((ndw + 1023) / 1024));
                if (am.i8) hipLaunchKernelGGL((k_dequantize_4bit_flat<T, QT, true, 4>), grid, dim3(256), 0, st, packed, am, ndw, sh, static_cast<T *>(out), 1);
                else hipLaunchKernelGGL((k_dequantize_4bit_flat<T, QT, false, 4>), grid, dim3(256), 0, st, packed, am, ndw, sh, static_cast<T *>(out), 1);
            } else {
                const dim3 grid((unsigned)((ndw + 255) / 256));
                if (am.i8) hipLaunchKernelGGL((k_dequantize_4bit_flat<T, QT, true, 1>), grid, dim3(256), 0, st, packed, am, ndw, sh, static_cast<T *>(out), store_policy);
                else hipLaunchKernelGGL((k_dequantize_4bit_flat<T, QT, false, 1>), grid, dim3(256), 0, st, packed, am, ndw, sh, static_cast<T *>(out), store_policy);
            }
            return check_launch("dequantize_4bit");
        }
    }
    const int64_t gpr = (cols + 7) / 8;
    const int64_t groups = rows * gpr;
    const bool row_grid = gpr >= 256 && (gpr + 255) / 256 <= 65535 && rows <= 0x7FFFFFFF;
    const dim3 grid = row_grid ? dim3((unsigned)rows, (unsigned)((gpr + 255) / 256)) : dim3((unsigned)((groups + 255) / 256));
    const bool vec_ok = aligned16(out) && (cols % 8 == 0);
    if (am.i8)
        hipLaunchKernelGGL((k_dequantize_4bit<T, QT, true>), grid, dim3(256), 0, st, packed, am, rows, cols,
                           cols_padded, blocksize, static_cast<T *>(out), vec_ok, row_grid);
    else
        hipLaunchKernelGGL((k_dequantize_4bit<T, QT, false>), grid, dim3(256), 0, st, packed, am, rows, cols,
                           cols_padded, blocksize, static_cast<T *>(out), vec_ok, row_grid);
    return check_launch("dequantize_4bit");
}

int dequantize_4bit_dispatch(const uint8_t *packed, const AbsmaxView &am, int64_t rows, int64_t cols,
                             int64_t cols_padded, int blocksize, int qt, int out_dtype, void *out,
                             hipStream_t st, int store_policy) {
#define MBNB_DQ(T)                                                                                              \
    (qt == MBNB_NF4 ? launch_dequantize_4bit<T, MBNB_NF4>(packed, am, rows, cols, cols_padded, blocksize, out, st, store_policy) \
                    : launch_dequantize_4bit<T, MBNB_FP4>(packed, am, rows, cols, cols_padded, blocksize, out, st, store_policy))
    switch (out_dtype) {
        case MBNB_F16: return MBNB_DQ(f16_t);
        case MBNB_BF16: return MBNB_DQ(bf16_t);
        default: return MBNB_DQ(float);
    }
#undef MBNB_DQ
}

int quantize_rowwise_dispatch(const void *, int, int64_t, int64_t, int8_t *, float *, hipStream_t);
int dequantize_rowwise_dispatch(const int8_t *, const float *, int64_t, int64_t, int, void *, hipStream_t, int store_policy = 0);

int quantize_blockwise_dispatch(const void *A, int dtype, int64_t numel, int blocksize, const float *absmax_in,
                                int8_t *out, float *absmax_out, hipStream_t st) {
    // whole blocks with their own absmax ARE quantize_rowwise over [numel / blocksize, blocksize] (functional.py:513-521 and
    // :617-622: the same clamp, the same 127 / absmax, the same round and clamp; both return the absmax itself): its
    // vectorised / single-pass kernels instead of the scalar one below
    if (absmax_in == nullptr && numel > 0 && numel % blocksize == 0 && blocksize >= 64)
        return quantize_rowwise_dispatch(A, dtype, numel / blocksize, blocksize, out, absmax_out, st);
    const unsigned grid = (unsigned)((numel + blocksize - 1) / blocksize);
    switch (dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_quantize_blockwise<f16_t>, dim3(grid), dim3(256), 0, st, static_cast<const f16_t *>(A), numel, blocksize, absmax_in, out, absmax_out); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_quantize_blockwise<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t *>(A), numel, blocksize, absmax_in, out, absmax_out); break;
        default: hipLaunchKernelGGL(k_quantize_blockwise<float>, dim3(grid), dim3(256), 0, st, static_cast<const float *>(A), numel, blocksize, absmax_in, out, absmax_out); break;
    }
    return check_launch("quantize_blockwise");
}

int dequantize_blockwise_dispatch(const int8_t *q, int64_t numel, const float *absmax, int blocksize, int out_dtype,
                                  void *out, hipStream_t st) {
    // whole blocks of >= 1024 values: dequantize_rowwise over [numel / blocksize, blocksize] (functional.py:592-594 and :635:
    // q.float() * (absmax / 127)), 16 values per thread
    if (numel > 0 && numel % blocksize == 0 && blocksize >= 1024 && blocksize % 16 == 0 && numel / blocksize < 65536)
        return dequantize_rowwise_dispatch(q, absmax, numel / blocksize, blocksize, out_dtype, out, st);
    const unsigned grid = (unsigned)((numel + 255) / 256);
    switch (out_dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_dequantize_blockwise<f16_t>, dim3(grid), dim3(256), 0, st, q, numel, absmax, blocksize, static_cast<f16_t *>(out)); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_dequantize_blockwise<bf16_t>, dim3(grid), dim3(256), 0, st, q, numel, absmax, blocksize, static_cast<bf16_t *>(out)); break;
        default: hipLaunchKernelGGL(k_dequantize_blockwise<float>, dim3(grid), dim3(256), 0, st, q, numel, absmax, blocksize, static_cast<float *>(out)); break;
    }
    return check_launch("dequantize_blockwise");
}

int quantize_rowwise_dispatch(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out, float *scales,
                              hipStream_t st) {
    const bool vec_ok = aligned16(A) && aligned16(out) && (cols % 8 == 0);
    const unsigned grid = (unsigned)rows;
    if (vec_ok && cols <= 8192 && dtype != MBNB_F32) {   // the row fits the registers of one workgroup: one pass over memory
        if (dtype == MBNB_F16) hipLaunchKernelGGL(k_quantize_rowwise_regs<f16_t>, dim3(grid), dim3(256), 0, st, static_cast<const f16_t *>(A), cols, out, scales);
        else hipLaunchKernelGGL(k_quantize_rowwise_regs<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t *>(A), cols, out, scales);
        return check_launch("quantize_rowwise");
    }
    switch (dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_quantize_rowwise<f16_t>, dim3(grid), dim3(256), 0, st, static_cast<const f16_t *>(A), rows, cols, out, scales, vec_ok); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_quantize_rowwise<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t *>(A), rows, cols, out, scales, vec_ok); break;
        default: hipLaunchKernelGGL(k_quantize_rowwise<float>, dim3(grid), dim3(256), 0, st, static_cast<const float *>(A), rows, cols, out, scales, vec_ok); break;
    }
    return check_launch("quantize_rowwise");
}

int quantize_fp8_dispatch(const void *A, int dtype, int64_t rows, int64_t cols, uint8_t *out, float *scales, hipStream_t st) {
    const bool vec_ok = (cols % 8 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
    const unsigned grid = (unsigned)rows;
    switch (dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_quantize_fp8<f16_t>, dim3(grid), dim3(256), 0, st, static_cast<const f16_t *>(A), rows, cols, out, scales, vec_ok); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_quantize_fp8<bf16_t>, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t *>(A), rows, cols, out, scales, vec_ok); break;
        default: hipLaunchKernelGGL(k_quantize_fp8<float>, dim3(grid), dim3(256), 0, st, static_cast<const float *>(A), rows, cols, out, scales, vec_ok); break;
    }
    return check_launch("quantize_fp8_e4m3");
}

int dequantize_fp8_dispatch(const uint8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype, void *out,
                            hipStream_t st, int store_policy) {
    if (launch_rows8_wt<true>(q, scales, rows, cols, out_dtype, out, st, store_policy)) return check_launch("dequantize_fp8_e4m3");
    if (cols % 16 == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && rows * cols > 0 && rows < 65536) {
        const dim3 g16((unsigned)((cols / 16 + 255) / 256), (unsigned)rows);
        switch (out_dtype) {
            case MBNB_F16: hipLaunchKernelGGL((k_dequantize_rows16<f16_t, true>), dim3(g16), dim3(256), 0, st, q, scales, rows, cols, static_cast<f16_t *>(out)); break;
            case MBNB_BF16: hipLaunchKernelGGL((k_dequantize_rows16<bf16_t, true>), dim3(g16), dim3(256), 0, st, q, scales, rows, cols, static_cast<bf16_t *>(out)); break;
            default: hipLaunchKernelGGL((k_dequantize_rows16<float, true>), dim3(g16), dim3(256), 0, st, q, scales, rows, cols, static_cast<float *>(out)); break;
        }
        return check_launch("dequantize_fp8_e4m3");
    }
    const unsigned grid = (unsigned)((rows * cols + 255) / 256);
    switch (out_dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_dequantize_fp8<f16_t>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<f16_t *>(out)); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_dequantize_fp8<bf16_t>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<bf16_t *>(out)); break;
        default: hipLaunchKernelGGL(k_dequantize_fp8<float>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<float *>(out)); break;
    }
    return check_launch("dequantize_fp8_e4m3");
}

int dequantize_rowwise_dispatch(const int8_t *q, const float *scales, int64_t rows, int64_t cols, int out_dtype,
                                void *out, hipStream_t st, int store_policy) {
    if (launch_rows8_wt<false>(reinterpret_cast<const uint8_t *>(q), scales, rows, cols, out_dtype, out, st, store_policy)) return check_launch("dequantize_rowwise");
    if (cols % 16 == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && rows * cols > 0 && rows < 65536) {
        const dim3 g16((unsigned)((cols / 16 + 255) / 256), (unsigned)rows);
        const uint8_t *qb = reinterpret_cast<const uint8_t *>(q);
        switch (out_dtype) {
            case MBNB_F16: hipLaunchKernelGGL((k_dequantize_rows16<f16_t, false>), dim3(g16), dim3(256), 0, st, qb, scales, rows, cols, static_cast<f16_t *>(out)); break;
            case MBNB_BF16: hipLaunchKernelGGL((k_dequantize_rows16<bf16_t, false>), dim3(g16), dim3(256), 0, st, qb, scales, rows, cols, static_cast<bf16_t *>(out)); break;
            default: hipLaunchKernelGGL((k_dequantize_rows16<float, false>), dim3(g16), dim3(256), 0, st, qb, scales, rows, cols, static_cast<float *>(out)); break;
        }
        return check_launch("dequantize_rowwise");
    }
    const unsigned grid = (unsigned)((rows * cols + 255) / 256);
    switch (out_dtype) {
        case MBNB_F16: hipLaunchKernelGGL(k_dequantize_rowwise<f16_t>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<f16_t *>(out)); break;
        case MBNB_BF16: hipLaunchKernelGGL(k_dequantize_rowwise<bf16_t>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<bf16_t *>(out)); break;
        default: hipLaunchKernelGGL(k_dequantize_rowwise<float>, dim3(grid), dim3(256), 0, st, q, scales, rows, cols, static_cast<float *>(out)); break;
    }
    return check_launch("dequantize_rowwise");
}

template <typename T>
static int launch_double_quant(const void *A, int64_t rows, int64_t cols, int8_t *out_col, int8_t *out_row,
                               float *col_stats, float *row_stats, int col_given, int row_given, hipStream_t st) {
    const T *a = static_cast<const T *>(A);
    if (cols % 8 == 0 && aligned16(A) && ((reinterpret_cast<uintptr_t>(out_col) | reinterpret_cast<uintptr_t>(out_row)) & 7) == 0 &&
        (rows + 7) / 8 <= 65535) {
        const unsigned gx = (unsigned)((cols / 8 + 255) / 256);
        if (!row_given || !col_given) {
            float *zr = row_given ? nullptr : row_stats, *zc = col_given ? nullptr : col_stats;
            const int64_t nmax = rows > cols ? rows : cols;
            hipLaunchKernelGGL(k_zero_stats2, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, st, zr, rows, zc, cols);
            hipLaunchKernelGGL(k_rowcol_absmax8<T>, dim3((unsigned)((cols + 127) / 128), (unsigned)((rows + 127) / 128)), dim3(256), 0, st, a,
                               rows, cols, reinterpret_cast<uint32_t *>(zr), reinterpret_cast<uint32_t *>(zc));
        }
        hipLaunchKernelGGL(k_double_quant8<T>, dim3(gx, (unsigned)((rows + 7) / 8)), dim3(256), 0, st, a, rows, cols, col_stats,
                           row_stats, out_col, out_row, !col_given, !row_given);
        return check_launch("double_quant");
    }
    if (!row_given) hipLaunchKernelGGL(k_row_absmax<T>, dim3((unsigned)rows), dim3(256), 0, st, a, rows, cols, row_stats);
    if (!col_given) {
        hipError_t e = hipMemsetAsync(col_stats, 0, sizeof(float) * cols, st);
        if (e != hipSuccess) {
            set_error("double_quant: hipMemsetAsync failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        dim3 grid((unsigned)((cols + 255) / 256), (unsigned)((rows + 63) / 64));
        hipLaunchKernelGGL(k_col_absmax<T>, grid, dim3(256), 0, st, a, rows, cols, reinterpret_cast<uint32_t *>(col_stats));
        hipLaunchKernelGGL(k_clamp_stats, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, col_stats, cols);
    }
    hipLaunchKernelGGL(k_double_quant<T>, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, st, a, rows, cols,
                       col_stats, row_stats, out_col, out_row);
    return check_launch("double_quant");
}

int double_quant_dispatch(const void *A, int dtype, int64_t rows, int64_t cols, int8_t *out_col, int8_t *out_row,
                          float *col_stats, float *row_stats, int col_given, int row_given, hipStream_t st) {
    switch (dtype) {
        case MBNB_F16: return launch_double_quant<f16_t>(A, rows, cols, out_col, out_row, col_stats, row_stats, col_given, row_given, st);
        case MBNB_BF16: return launch_double_quant<bf16_t>(A, rows, cols, out_col, out_row, col_stats, row_stats, col_given, row_given, st);
        default: return launch_double_quant<float>(A, rows, cols, out_col, out_row, col_stats, row_stats, col_given, row_given, st);
    }
}

}  // namespace mbnb
