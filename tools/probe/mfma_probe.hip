// mfma_probe.hip — measurement aid for bench.py's `roofline.empirical` block.  NOT part of the product: built as
// tools/libmbnb_probe.so (tools/Makefile; __graft_entry__.build()), loaded by bench.py / tools/exp only when present:
// a bare MFMA loop, one wave per SIMD on every CU, so that the bench can print what the matrix pipe SUSTAINS on the box
// it runs on (the clock the chip holds under MFMA load is well under the 2.4 GHz the spec peak assumes) next to the
// fraction of the spec peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

namespace {

template <bool I8>
__global__ __launch_bounds__(256) void k_probe_mfma(int iters, float *sink, uint32_t seed) {
    const uint32_t h = (threadIdx.x * 2654435761u) ^ seed ^ (blockIdx.x * 40503u);
    float s = 0.0f;
    if constexpr (!I8) {
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            a[e] = (bf16_t)((float)((int)((h >> e) & 255) - 128) * 0.01f);
            b[e] = (bf16_t)((float)((int)((h >> (e + 8)) & 255) - 128) * 0.01f);
        }
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        s = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        i32x4 a = {(int)h, (int)(h * 3), (int)(h * 5), (int)(h * 7)}, b = {(int)(h * 11), (int)(h * 13), (int)(h * 17), (int)(h * 19)};
        i32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
        for (int i = 0; i < iters; i++) {
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d3, 0, 0, 0);
            a[0] ^= i;   // keeps the chain from being folded: one VALU per four MFMAs
        }
        int si = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) si += d0[e] ^ d1[e] ^ d2[e] ^ d3[e];
        // an integer test: `(float)si == 123.456f` can never hold (an integer has no fraction) and the whole loop was folded away
        if (si == 0x12345679) sink[0] = 1.0f;
    }
    if (s == 123.456f) sink[0] = s;
}

}  // namespace

// Launches the loop on `stream` (256 workgroups x 4 waves, `iters` iterations of four MFMAs: kind 0 v_mfma_f32_32x32x16_bf16,
// kind 1 v_mfma_i32_32x32x32_i8) and returns the number of MFMA wave-instructions issued (> 0), or <= 0 on failure.  The
// caller times it: sustained rate = return value * 32768 (bf16) or 65536 (i8) operations / time.  `sink`: 4 device bytes.
extern "C" int64_t mbnb_probe_mfma(int kind, int iters, float *sink, void *stream) {
    if ((kind != 0 && kind != 1) || iters <= 0 || !sink) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind == 0) hipLaunchKernelGGL(k_probe_mfma<false>, dim3(256), dim3(256), 0, st, iters, sink, 7u);
    else hipLaunchKernelGGL(k_probe_mfma<true>, dim3(256), dim3(256), 0, st, iters, sink, 7u);
    if (hipGetLastError() != hipSuccess) return -2;
    return (int64_t)256 * 4 * 4 * iters;
}
