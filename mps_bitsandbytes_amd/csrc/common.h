// common.h — shared device/host helpers for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mbnb_hip.h"

namespace mbnb {

// ---------------------------------------------------------------- element types
using f16_t = _Float16;
using bf16_t = __bf16;

template <int DT> struct ElemT;
template <> struct ElemT<MBNB_F16> { using type = f16_t; };
template <> struct ElemT<MBNB_BF16> { using type = bf16_t; };
template <> struct ElemT<MBNB_F32> { using type = float; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
// RNE conversions: plain casts lower to v_cvt_f16_f32 / v_cvt_pk_bf16_f32 on gfx950.
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
// f32 -> f16 must round a value that already exists in f32: without the (empty) asm the compiler folds a preceding
// multiply into v_fma_mixlo_f16, which rounds the exact product ONCE, where the reference rounds to f32 and then
// to f16 (functional.py: `(code * absmax).to(dtype)`) -- 1 ulp apart on f16 ties (seen: 64 of 393 216 embedding values).
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) {
    asm("" : "+v"(v));
    return (f16_t)v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t f16x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

// pack two f32 into one dword of two 16-bit values (RNE), element 0 in the low half
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) {
    asm("" : "+v"(lo), "+v"(hi));  // no v_fma_mix*_f16 folding of the producing multiplies (see from_f32<f16_t>)
    f16x2 v = {(f16_t)lo, (f16_t)hi};
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
    bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> __device__ __forceinline__ float unpack_lo(uint32_t w);
template <typename T> __device__ __forceinline__ float unpack_hi(uint32_t w);
template <> __device__ __forceinline__ float unpack_lo<f16_t>(uint32_t w) {
    return (float)__builtin_bit_cast(f16x2, w)[0];
}
template <> __device__ __forceinline__ float unpack_hi<f16_t>(uint32_t w) {
    return (float)__builtin_bit_cast(f16x2, w)[1];
}
template <> __device__ __forceinline__ float unpack_lo<bf16_t>(uint32_t w) {
    return __builtin_bit_cast(float, w << 16);
}
template <> __device__ __forceinline__ float unpack_hi<bf16_t>(uint32_t w) {
    return __builtin_bit_cast(float, w & 0xFFFF0000u);
}

// ---------------------------------------------------------------- code tables (functional.py:21-32)
__device__ __forceinline__ float nf4_code(int i) {
    constexpr float t[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                             -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                             0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                             0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                             0.7229568362236023f, 1.0f};
    return t[i];
}
__device__ __forceinline__ float fp4_code(int i) {
    constexpr float t[16] = {0.0f, 0.0625f, 0.125f, 0.25f, 0.375f, 0.5f, 0.75f, 1.0f,
                             -0.0f, -0.0625f, -0.125f, -0.25f, -0.375f, -0.5f, -0.75f, -1.0f};
    return t[i];
}
template <int QT> __device__ __forceinline__ float code_value(int i) {
    return QT == MBNB_NF4 ? nf4_code(i) : fp4_code(i);
}

// Fill a 16-entry LDS table with the code values (one lane per entry).
template <int QT> __device__ __forceinline__ void fill_code_lut(float *lut, int tid) {
    if (tid < 16) {
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (tid == i) v = code_value<QT>(i);
        lut[tid] = v;
    }
}

// ---------------------------------------------------------------- 8-bit weight formats (W8A16 kernels)
// WF = 0: row-wise INT8 (functional.py:607-636: value = q * (scale / 127));  WF = 1: the reference's FP8 E4M3
// (functional.py:1086-1215: value = decode(byte) * scale).  Both are decoded byte by byte inside the GEMM producers.
constexpr int W8_INT8 = 0, W8_FP8 = 1;

// decoder of functional.py:1178-1215: (1 + m/8) * 2^(e-7); e = 0: (m/8) * 2^-6; 0x7F / 0xFF: NaN
__device__ __forceinline__ float fp8_e4m3_to_float(uint32_t b) {
    // (sign | e << 23 | m << 20) read as f32 is (1 + m/8) * 2^(e-127) -- or the f32 subnormal m * 2^-129 for e = 0 --
    // so one multiply by 2^120 yields (1 + m/8) * 2^(e-7) and (m/8) * 2^-6 alike (f32 denormals are not flushed).
    // Upper bits of `b` are ignored.
    const uint32_t t = b << 24;
    const uint32_t u = (t & 0x80000000u) | ((t >> 4) & 0x07F00000u);
    float f = __builtin_bit_cast(float, u) * 0x1p120f;
    if ((t & 0x7F000000u) == 0x7F000000u) f = __builtin_bit_cast(float, 0x7FC00000u);
    return f;
}

// floor(torch.log2(a)) for finite a > 0 as the reference's encoder sees it: log2 is correctly rounded, so the
// exponent already reads k a few f32 ulps below 2^k (oracle/oracle.c fp8_exponent; pinned by tests/golden/g7_fp8.npz)
__device__ __forceinline__ int fp8_exponent(float a) {
    const uint32_t u = __builtin_bit_cast(uint32_t, a);
    const uint32_t ef = (u >> 23) & 0xFFu;
    if (ef == 0) return -127;
    int e = (int)ef - 127;
    const int k = e + 1;
    const uint32_t n = k >= 9 ? 5u : k >= 5 ? 2u : k >= 3 ? 1u : k >= -1 ? 0u : k >= -3 ? 1u : k >= -7 ? 2u : k >= -15 ? 5u : 11u;
    if ((0x7FFFFFu - (u & 0x7FFFFFu)) < n) e += 1;
    return e;
}

// encoder of functional.py:1106-1163 (NOT the OCP conversion): no mantissa carry, subnormals flushed to signed zero,
// biased exponent >= 15 -> 0x77 (so every |v| >= 256 becomes 240), NaN -> 0x7F.  The reference's float steps
//   e = floor(log2(a)); biased = e + 7; mb = clamp((a / 2^e - 1) * 8 + 0.5, 0, 7)
// on the bits of a = min(|v|, 448) (exponent field ef, mantissa m, k = ef - 126):
//   * log2 is correctly rounded, so a mantissa within n(k) ulps below the next power of two already reads e + 1
//     (fp8_exponent above); only k in [-6, 8] can change the byte, where n = 0 / 1 / 2 for |2k - 1| < 5 / < 9 / >= 9;
//   * without that bump (a / 2^e - 1) * 8 + 0.5 = m / 2^20 + 0.5 exactly, so mb = min((m + 2^19) >> 20, 7);
//     with it the expression is 0.5 - (2^23 - m) / 2^21 < 0.5, so mb = 0.
// Same bytes as the float form on every input (tests: g7 goldens from the reference, all |v| patterns of
// test_fp8_quantize_dynamic_range_bit_exact and the full-size oracle comparison); ~20 integer VALU instead of ~60.
__device__ __forceinline__ uint32_t float_to_fp8_e4m3(float v) {
    if (v != v) return 0x7Fu;
    const uint32_t sign = v < 0.0f ? 0x80u : 0u;
    const uint32_t ab = __builtin_bit_cast(uint32_t, fminf(fabsf(v), 448.0f));
    if (ab == 0u) return sign;
    const int k = (int)(ab >> 23) - 126;
    const uint32_t m = ab & 0x7FFFFFu;
    const int t = 2 * k - 1, at = t < 0 ? -t : t;
    const uint32_t n = at >= 9 ? 2u : (at >= 5 ? 1u : 0u);
    const bool bump = (0x7FFFFFu - m) < n;
    const int biased = k + 6 + (bump ? 1 : 0);
    if (biased >= 15) return sign | 0x77u;
    if (biased <= 0) return sign;
    uint32_t mb = (m + 0x80000u) >> 20;
    mb = bump ? 0u : (mb > 7u ? 7u : mb);
    return sign | ((uint32_t)biased << 3) | mb;
}

// Optional second term of an int8 GEMM epilogue (OutlierAwareLinear.forward, nn/outlier_aware.py:141-143, :110-111):
//   out = RNE(RNE(RNE(acc * sA/127 * sB/127) + RNE(X[:, oidx] . ow^T)) + bias)
// x == nullptr: no outlier term; bias == nullptr: no bias; both null: the plain matmul_int8 epilogue.
struct OutlierEpilogue {
    const void *x;        // [M, ldx] outlier activations (compact, zero padded) in the output dtype (16-bit)
    int64_t ldx;          // = 16 * ceil(n_out / 16)
    const int64_t *oidx;  // [n_out]
    int64_t n_out;
    const void *ow;       // [N, n_out] outlier weights in the output dtype
    const void *bias;     // [N] or nullptr
};

template <int WF> __device__ __forceinline__ float w8_row_scale(float s) { return WF == W8_INT8 ? s / 127.0f : s; }
template <int WF> __device__ __forceinline__ float w8_decode(uint32_t byte) {   // byte in bits 0..7
    if constexpr (WF == W8_INT8) return (float)(int)(int8_t)byte;
    else return fp8_e4m3_to_float(byte);
}
// byte `sel` (0..3, a constant after unrolling) of `word`.  FP8: gfx950's v_cvt_f32_fp8 is the OCP E4M3 conversion, which
// is exactly the reference's decoder (bias 7, subnormals, NaN at 0x7F / 0xFF) -- one instruction including the byte
// select; fp8_e4m3_to_float above is the same function in portable arithmetic (dequantize kernel, all-bytes test).
template <int WF> __device__ __forceinline__ float w8_decode_sel(uint32_t word, int sel) {
    if constexpr (WF == W8_INT8) {
        return (float)(int)(int8_t)(word >> (8 * sel));
    } else {
        switch (sel) {
            case 0: return __builtin_amdgcn_cvt_f32_fp8((int)word, 0);
            case 1: return __builtin_amdgcn_cvt_f32_fp8((int)word, 1);
            case 2: return __builtin_amdgcn_cvt_f32_fp8((int)word, 2);
            default: return __builtin_amdgcn_cvt_f32_fp8((int)word, 3);
        }
    }
}

// ---------------------------------------------------------------- absmax decode
struct AbsmaxView {
    const float *f32;
    const int8_t *i8;
    const float *am2;
    int bs2;
};

// absmax[i]; for the double-quantised form this is dequantize_blockwise's arithmetic
// (functional.py:592-594): q.float() * (absmax2 / 127.0)  -- true division, then one multiply.
template <bool NESTED> __device__ __forceinline__ float load_absmax(const AbsmaxView &a, int64_t i) {
    if constexpr (NESTED) {
        float s = a.am2[i / a.bs2] / 127.0f;
        return (float)a.i8[i] * s;
    } else {
        return a.f32[i];
    }
}

// `127.0 / tensor` in the reference is evaluated by torch as reciprocal(tensor) * 127.0
// (Python scalar / Tensor -> Tensor.__rtruediv__): two roundings.  See oracle/oracle.c rscale127.
__device__ __forceinline__ float rscale127(float absmax) {
    float r = 1.0f / absmax;  // correctly rounded (no fast-math)
    return r * 127.0f;
}

__device__ __forceinline__ int8_t quant_i8(float x, float scale) {
    float q = rintf(x * scale);  // torch.round = half-to-even
    q = fminf(fmaxf(q, -127.0f), 127.0f);
    return (int8_t)(int)q;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------- host side
// Internal status of the per-path launchers ("this path does not serve the call, try the next one").  Never crosses the C ABI,
// and cannot collide with a hipError_t (> 0; hipErrorInvalidValue == 1) or an MBNB_ERR_* code (-1 .. -3).
constexpr int MBNB_NOT_APPLICABLE = -1000;

// LDS-DMA issue in the pipelined GEMMs (gemm_dense.h, gemm_dense128.h, gemm_i8_inplace.h, gemm_small.h, gemm_small8.h).  1 (default): four
// pieces share ONE M0 write -- the instruction's 12-bit offset is added to the LDS address and to the global address alike, so piece
// 4 g + m goes out with offset 1024 m from a per-lane offset that is 1024 m smaller (tools/exp/ab_m0.py: same bits; k_gemm_dense 93.4 ->
// 92.7 us at 4096^3).  0 (diagnostic builds): one s_mov m0 + s_nop per piece.
#ifndef GD_M0_GROUP
#define GD_M0_GROUP 1
#endif
// k_gemm_dense's 16-bit epilogue stages the wave's tile through its private LDS in parts of 16 * GD_EPI_GROUPS rows: with one 16-row group
// per part the first stores leave after 1/8 of the conversions instead of 1/2 (tools/exp/ab_dense_epilogue.py: same bits; 91.9 -> 91.1 us at
// 4096^3, 29.8 -> 28.6 us at 4000 x 4096 x 1024).  4 (diagnostic builds): two halves of 64 rows.
#ifndef GD_EPI_GROUPS
#define GD_EPI_GROUPS 1
#endif
// Cache policy of the 16-bit epilogue stores of k_gemm_dense and k_gemm_dense128 (store_out16).  1 (default): write-through "sc1" -- the tile's
// bytes are not read again by this launch and leave nothing for the end of the launch to drain (tools/exp/ab_dense_store.py: k_gemm_dense alone
// 93.9 -> 93.8 us, inside the step 104.5 -> 103.8 us at 4096^3, 284.0 -> 283.1 at 11008 x 4096; ab_epilogue_store.py: the 1024-row step on
// k_gemm_dense128 43.0 -> 42.1 us).  Diagnostic builds: 0 nontemporal (round 2: 98.5 -> 95.6 us against plain stores), 2 "sc1 nt",
// 3 "sc0 sc1 nt", 4 plain.  k_gemm_i8_inplace keeps nontemporal stores (store_out16_nt; "sc1" there: 53.0 -> 53.6 us), and so do the
// kernels that are off by default (not measured).
#ifndef GD_EPI_STORE
#define GD_EPI_STORE 1
#endif
// One aligned 16-byte f32 store, write-through ("sc1"): split-K partials, read next by the reduction launch on other XCDs.
__device__ __forceinline__ void store_f32x4_wt(float *dst, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory"); }
__device__ __forceinline__ void store_out16_nt(u32x4 *dst, u32x4 v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void store_out16(u32x4 *dst, u32x4 v) {
#if GD_EPI_STORE == 0
    __builtin_nontemporal_store(v, dst);
#elif GD_EPI_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#elif GD_EPI_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v) : "memory");
#elif GD_EPI_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
#else
    *dst = v;
#endif
}
void set_error(const char *fmt, ...);
void set_kernel_name(const char *name);
int check_launch(const char *what);
// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) ONCE per (device, kernel): the
// attribute belongs to the current device's copy of the function, so the record is kept per device; later calls are
// one mutex-protected table lookup and never reach the HIP runtime (api.hip).  Returns 0 or a hipError_t.
int ensure_dyn_lds(const void *func, int bytes, const char *what);

}  // namespace mbnb
